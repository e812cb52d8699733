"""Per-step wall time of the bench's training job (synchronised): python scripts/dbg/step_times.py C5 40"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
import torch
import helpers_bench as HB
import diff_surfel_rasterization as dsr
wl = sys.argv[1] if len(sys.argv) > 1 else "C5"; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
tr = HB.make_trainer(dev, wl)
if isinstance(tr, tuple): tr = tr[0]
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("step %d: host %.1f ms, total %.1f ms, R %d, reserved %.1f GB alloc %.1f GB" % (i, (t1 - t0) * 1e3, (t2 - t0) * 1e3, dsr.last_num_rendered,
          torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30), flush=True)
