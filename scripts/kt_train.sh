#!/bin/bash
# kernel-trace stats of full training iterations on the GPU box: bash scripts/kt_train.sh <tag> [workload] [iters]
TAG=${1:-q}; WL=${2:-C2}; IT=${3:-60}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/ktt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python $ROOT/scripts/train_bench.py $WL $IT > $OUT/bench.log 2>&1
cd $ROOT
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -delete
tail -n 1 $OUT/bench.log
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::","").split("(")[0].replace("surfel::","").replace("void ","")[:60]
    print("%-62s calls %5s  avg %9.1f us  tot %5.1f%%" % (n, r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
