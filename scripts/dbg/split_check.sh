#!/bin/bash
# round 4: validates blend_bwd list splitting on a GPU box inside a small budget: the tests, then rows / scan / split on the trained leg
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py -q -m gpu -k "list_splitting or trained_state" -s 2>&1 | grep -v "^$" | cut -c1-300 | tail -150 > gpurun_out/split_tests.log
tail -3 gpurun_out/split_tests.log
timeout 150 python scripts/ab_stage.py trained bwd_variant=0,bwd_tune=0,bwd_split=0 bwd_variant=2,bwd_tune=1,bwd_split=0 bwd_variant=2,bwd_tune=1,bwd_split=1 --steps 30 --reps 2 > gpurun_out/split_ab.log 2>&1
tail -c 2500 gpurun_out/split_ab.log
