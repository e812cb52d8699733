"""Forward-only Msplats/s @1080p over P in {0.3, 1, 2, 5, 10} M (SURVEY.md 8d sweep) and fwd+bwd ms: python scripts/sweep_1080p.py
Writes one JSON line per P (same generator / timing as bench.py's fwd_1080p leg)."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
import torch, synthetic
from helpers_bench import fwd_1080p, raster_fwd_bwd
dev = torch.device("cuda:0")
for P in (300_000, 1_000_000, 2_000_000, 5_000_000, 10_000_000):
    name = "1080p_%d" % P
    synthetic.CONFIGS[name] = (P, 1920, 1080, 12.0)
    r = fwd_1080p(dev, name, iters=12 if P > 2_000_000 else 20, warmup=4)
    fb = raster_fwd_bwd(dev, name, iters=8 if P > 2_000_000 else 20, warmup=3)
    r.update(P=P, fwd_bwd_ms=fb["ms_per_view"], fwd_bwd_Msplats_per_s=fb["Msplats_per_s"])
    print(json.dumps(r), flush=True)
    torch.cuda.empty_cache()
