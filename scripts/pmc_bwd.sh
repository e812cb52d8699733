#!/bin/bash
# SQ counters of the two blend_bwd variants on one workload: bash scripts/pmc_bwd.sh <tag> <workload ...>
TAG=${1:-q}; shift; ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcb_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
    --output-format csv -d $OUT/sq -o p -- python $ROOT/scripts/bwd_ab.py "$@" > $OUT/ab_sq.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY \
    --output-format csv -d $OUT/sq2 -o p -- python $ROOT/scripts/bwd_ab.py "$@" > $OUT/ab_sq2.log 2>&1
cd $ROOT
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -delete
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "blend_bwd" not in k: continue
        name = ("rows" if "rows" in k else "quad") + ("_stats" if "Lb1" in k or "<true>" in k else "")
        acc[name + "|grid" + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
for k in sorted(out):
    print(k, {c: round(v) for c, v in sorted(out[k].items())})
PY
