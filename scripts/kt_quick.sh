#!/bin/bash
# quick kernel-trace stats on the GPU box: bash scripts/kt_quick.sh <tag> [bench args...]
TAG=${1:-q}; shift; ROOT=$(pwd); OUT=$ROOT/gpurun_out/kt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python $ROOT/bench.py --no-cpu-baseline --no-1080p --no-train-iter "$@" > $OUT/bench.log 2>&1
cd $ROOT
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -delete
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::","").split("(")[0].replace("surfel::","").replace("void ","")[:50]
    print("%-52s calls %5s  avg %9.1f us  tot %5.1f%%" % (n, r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
