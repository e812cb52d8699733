#!/bin/bash
# Run ON THE GPU BOX from the repo root (via gpurun):  bash scripts/profile_gpu.sh <tag> [workload] [full|short]
# Produces under gpurun_out/prof_<tag>_<workload>/ :
#   kt/      rocprofv3 --kernel-trace --stats of `python bench.py --workload <workload> --quick` (timed training step only: side legs off,
#            so the per-kernel averages are those of the timed region)
#   fetch/   --pmc FETCH_SIZE   (own pass: FETCH_SIZE takes 3 of the 4 TCC slots)
#   write/   --pmc WRITE_SIZE   (own pass)
#   sq/      --pmc SQ_* issue/wait counters
#   grbm/    --pmc GRBM_GUI_ACTIVE (+ that pass's own kernel trace) -> effective clock per kernel
#   sq2/     (full only) lanes enabled per VALU instruction, wait breakdown
# PMC passes never combine with sys/hip/hsa trace domains (only --kernel-trace is implied by rocprofv3 itself).
# The blend-backward walk is FORCED (SURFEL_OPTIONS=bwd_variant=<walk>; default rows) in every pass: under auto the kernel the device
# rule does not pick is launched as well and returns at once, which dilutes its per-launch means.  Trained workloads (trained / garden) are trained once and cached in /tmp.
# scripts/profile_summary.py then condenses these into profiles/.
TAG=${1:-r03}
WL=${2:-C2}
MODE=${3:-full}
WALK=${WALK:-0}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SURFEL_OPTIONS="bwd_variant=$WALK"
echo "$WALK" > $OUT/walk.txt
STATE=""
case $WL in trained|garden) STATE="--state /tmp/state_$WL.ply"; timeout 200 python $ROOT/bench.py --workload $WL $STATE --quick --steps 2 --warmup 1 > $OUT/bench_make_state.log 2>&1;; esac
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $ROOT/bench.py --workload $WL $STATE --quick > $OUT/bench_kt.log 2>&1
# (the PMC passes time the same views as the bench leg of this workload: same steps / warm-up, the window reseeds its view order)
case $WL in trained) SHORT="--workload $WL $STATE --steps 30 --warmup 5 --quick";; garden|C4|C2H) SHORT="--workload $WL $STATE --steps 20 --warmup 5 --quick";; *) SHORT="--workload $WL $STATE --steps 20 --warmup 5 --quick";; esac
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python $ROOT/bench.py $SHORT > $OUT/bench_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python $ROOT/bench.py $SHORT > $OUT/bench_write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT \
    --output-format csv -d $OUT/sq -o p -- python $ROOT/bench.py $SHORT > $OUT/bench_sq.log 2>&1
# effective shader clock per kernel: GRBM_GUI_ACTIVE / kernel wall time (MI355X_MICROARCH.md "DVFS give-back"); its kernel trace stays
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o p -- python $ROOT/bench.py $SHORT > $OUT/bench_grbm.log 2>&1
if [ "$MODE" = "full" ]; then
# lanes enabled per VALU instruction (the hardware's VALUUtilization; EXEC-enabled lanes, not lanes doing useful work: see
# profiles/r02_blend_bwd_variants.md) + wait breakdown
timeout 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE \
    --output-format csv -d $OUT/sq2 -o p -- python $ROOT/bench.py $SHORT > $OUT/bench_sq2.log 2>&1
fi
cd $ROOT
# keep what comes back small: kernel_trace of the PMC passes is not needed
find $OUT -name '*.db' -delete
find $OUT/fetch $OUT/write $OUT/sq $OUT/sq2 -name '*kernel_trace.csv' -delete 2>/dev/null
find $OUT/kt -name '*kernel_trace.csv' -size +20M -delete 2>/dev/null
tail -n 1 $OUT/bench_kt.log | cut -c1-300
