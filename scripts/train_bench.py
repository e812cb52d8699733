"""Full-training-iteration throughput alone (the `train_iter` leg of bench.py): python scripts/train_bench.py [workload] [iters]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
import torch
from helpers_bench import train_iter
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
print(json.dumps(train_iter(torch.device("cuda:0"), wl, iters=iters)))
