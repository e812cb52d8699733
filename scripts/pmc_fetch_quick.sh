#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every kernel of a short bench run: bash scripts/pmc_fetch_quick.sh <tag> [workload]
TAG=${1:-q}; WL=${2:-C2}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcf_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="--workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-1080p --no-raster-only"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python $ROOT/bench.py $SHORT > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python $ROOT/bench.py $SHORT > $OUT/bench_write.log 2>&1
cd $ROOT
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -delete
python - <<PY
import csv, glob, collections
def means(d):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("surfel::","").replace("void ","")[:40]].append(float(r["Counter_Value"]))
    return {k: sum(v)/len(v) for k, v in acc.items()}
f, w = means("fetch"), means("write")
for k in sorted(f, key=lambda k: -f[k]):
    if f[k] + w.get(k, 0) > 500:
        print("%-42s fetch %8.1f MB (x2 corrected)  write %8.1f MB" % (k, 2*f[k]*1024/1e6, w.get(k,0)*1024/1e6))
PY
