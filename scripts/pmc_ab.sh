#!/bin/bash
# Run ON THE GPU BOX (via gpurun):  bash scripts/pmc_ab.sh <workload> <options A> <options B> [kernel-name regex]
# SQ counters of one kernel under two SURFEL_OPTIONS settings (short bench runs; PMC only, no trace domains besides the implied kernel trace).
WL=${1:-trained}; A=${2:-fwd_pipe=0}; B=${3:-fwd_pipe=1}; KREG=${4:-blend_fwd}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STATE=""
case $WL in trained|garden) STATE="--state /tmp/state_$WL.ply"; timeout 120 python $ROOT/bench.py --workload $WL $STATE --quick --steps 2 --warmup 1 > $OUT/make_state.log 2>&1;; esac
for tag in A B; do
  if [ $tag = A ]; then OPT=$A; else OPT=$B; fi
  SURFEL_OPTIONS="$OPT" timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
      --output-format csv -d $OUT/$tag -o p -- python $ROOT/bench.py --workload $WL $STATE --steps 8 --warmup 2 --quick > $OUT/bench_$tag.log 2>&1
  SURFEL_OPTIONS="$OPT" timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
      --output-format csv -d $OUT/${tag}2 -o p -- python $ROOT/bench.py --workload $WL $STATE --steps 8 --warmup 2 --quick > $OUT/bench_${tag}2.log 2>&1
done
cd $ROOT
python - "$OUT" "$KREG" "$A" "$B" <<'PY'
import csv, glob, re, sys, collections, json
out, kreg, A, B = sys.argv[1:5]
for tag, opt in (("A", A), ("B", B)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in (tag, tag + "2"):
        for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True):
            for r in csv.DictReader(open(f)):
                if re.search(kreg, r["Kernel_Name"]):
                    acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(json.dumps({"setting": opt, "kernel": k, **{c: round(sum(x) / len(x), 1) for c, x in sorted(v.items())}, "launches": len(next(iter(v.values())))}))
PY
find $OUT -name '*.db' -delete; find $OUT -name '*.csv' -size +5M -delete
