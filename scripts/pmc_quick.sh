#!/bin/bash
# quick SQ counter pass on the GPU box: bash scripts/pmc_quick.sh <tag> [workload] ; prints blend kernel counters
TAG=${1:-q}; WL=${2:-C2}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcq_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="--workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-1080p --no-train-iter"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT \
    --output-format csv -d $OUT/sq -o p -- python $ROOT/bench.py $SHORT > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES \
    --output-format csv -d $OUT/sq2 -o p -- python $ROOT/bench.py $SHORT > $OUT/bench_sq2.log 2>&1
cd $ROOT
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -delete
python scripts/pmc_summary.py $OUT/sq/*/p_counter_collection.csv $OUT/sq/p_counter_collection.csv 2>/dev/null | grep -A9 "^blend"
python scripts/pmc_summary.py $OUT/sq2/*/p_counter_collection.csv $OUT/sq2/p_counter_collection.csv 2>/dev/null | grep -A9 "^blend"
