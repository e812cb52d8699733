#!/bin/bash
# Run ON THE GPU BOX: profile passes of several workloads, condensed on the box (raw rocprofv3 output is too large to travel back).
#   bash scripts/profile_round.sh <tag> "<workload>:<walk>:<mode> ..."      e.g.  r06 "garden:3:full C2:0:short"
# Leaves gpurun_out/profiles_<tag>/ = what scripts/profile_summary.py wrote (copy into profiles/ and commit).
TAG=$1; shift
ROOT=$(pwd)
mkdir -p gpurun_out/profiles_$TAG
for spec in $1; do
  IFS=: read WL WK MODE <<< "$spec"
  WALK=$WK bash scripts/profile_gpu.sh $TAG $WL ${MODE:-short} > gpurun_out/profiles_$TAG/${WL}_profile_gpu.log 2>&1
  python scripts/profile_summary.py $TAG $WL > gpurun_out/profiles_$TAG/${WL}_summary.log 2>&1
  python scripts/roofline_table.py $TAG $WL >> gpurun_out/profiles_$TAG/${WL}_summary.log 2>&1
  cp profiles/${TAG}_${WL}_* profiles/pmc_traffic.json gpurun_out/profiles_$TAG/ 2>/dev/null
  for l in gpurun_out/prof_${TAG}_$WL/bench_*.log; do tail -n 3 $l > gpurun_out/profiles_$TAG/${WL}_$(basename $l).tail; done
  rm -rf gpurun_out/prof_${TAG}_$WL
done
du -sh gpurun_out/profiles_$TAG
