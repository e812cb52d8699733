#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the garden preset's knobs -> surfels after training, step time.  Usage: bash scripts/garden_tune.sh "GT INIT ITERS PX" ...
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  set -- $cfg
  i=$((i+1))
  GARDEN_GT=$1 GARDEN_INIT=$2 GARDEN_ITERS=$3 GARDEN_PX=${4:-0.010} timeout 900 python bench.py --workload garden --quick --steps 10 --warmup 3 2> gpurun_out/garden_tune_$i.err | tail -n 1 > gpurun_out/garden_tune_$i.json
  python - <<PY
import json
try:
    j = json.load(open("gpurun_out/garden_tune_$i.json"))
    print("$cfg ->", j["config"]["workload"], "| ms/step", j["ms_per_step"], "| R", j["config"]["instances_R_mean"], "Rs", j["config"]["instances_staged_mean"], "|", j["config"]["trained_state"])
except Exception as e:
    print("$cfg -> failed", e)
PY
done
