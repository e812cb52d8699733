"""Large-sort A/B on the GPU box: the library's three-launches-per-pass radix sort vs rocprim::radix_sort_pairs inside the
rasterizer forward (stage timings from HIP events, debug mode 2).   python scripts/sort_ab.py [workload ...]"""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "2d-gaussian-splatting_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
import surfel_native as n
import synthetic
from helpers import HipRun, scene_args

lib = n.load()
for name in (sys.argv[1:] or ["C4", "C5"]):
    P, W, H, zf = synthetic.CONFIGS[name]
    a = scene_args(synthetic.make_scene(P, W, H, seed=0, z_far=zf, px_radius=synthetic.PX_RADIUS.get(name)))
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    row = {"workload": name, "P": P, "tiles": tiles, "tile_id_bits": max(1, (tiles - 1).bit_length())}
    ref = None
    for impl, nm in ((0, "own"), (3, "rocprim")):
        lib.surfel_set_option(b"large_sort", impl)
        for mode in (0, 2):                      # depth-presorted emission (P-sized 32-bit sort + R-sized tile sort) | per-tile depth sort (R-sized tile sort only)
            run = HipRun(a, debug=2 | n.opt_tile_sort(mode))
            for _ in range(2):
                run.forward()
            n.collect_stage_times()
            for _ in range(5):
                run.forward()
            t = n.collect_stage_times()
            row["R"] = run.R
            key = "%s_%s" % (nm, "presorted" if mode == 0 else "pertile")
            row[key] = {k: round(1e3 * v[0] / v[1], 1) for k, v in t.items() if k in ("depth_sort_scan", "tile_sort", "tile_depth_sort", "emit_instances")}
            img = run.color.cpu().numpy()
            if ref is None:
                ref = img
            else:
                row["identical_images"] = bool(row.get("identical_images", True) and np.array_equal(ref, img))
            del run
            torch.cuda.empty_cache()
    lib.surfel_set_option(b"large_sort", 2)
    R = row["R"]
    for nm in ("own", "rocprim"):
        ts = row[nm + "_presorted"]["tile_sort"]
        passes = -(-row["tile_id_bits"] // 8)
        row[nm + "_tile_sort_TBps_per_pass"] = round(R * 16 / (ts * 1e-6 / passes) / 1e12, 2)
    print(json.dumps(row), flush=True)
