"""A/B of the two blend_bwd variants on the GPU box: kernel time (HIP events on the launch stream, debug mode 2) and the
instrumented useful-lane fraction (surfel_debug_set_blend_stats).   python scripts/bwd_ab.py [workload ...]
Workloads: synthetic.CONFIGS names, or  name:px_radius[:P]  to override the median 1-sigma radius / the surfel count
(e.g. C2:7 = heavy footprints, C2::900000 = three times the surfels)."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "2d-gaussian-splatting_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import surfel_native as n  # noqa: E402
import synthetic  # noqa: E402
from helpers import HipRun, scene_args  # noqa: E402


def main():
    lib = n.load()
    out = []
    for spec in (sys.argv[1:] or ["C2", "C4"]):
        parts = spec.split(":")          # name[:px_radius[:P]]
        name, rad = parts[0], (parts[1] if len(parts) > 1 else "")
        P, W, H, zf = synthetic.CONFIGS[name]
        if len(parts) > 2:
            P = int(parts[2])
        sc = synthetic.make_scene(P, W, H, seed=0, z_far=zf, px_radius=float(rad) if rad else None)
        a = scene_args(sc)
        rng = np.random.default_rng(0)
        gC = rng.normal(size=(3, H, W)).astype(np.float32); gO = rng.normal(size=(7, H, W)).astype(np.float32)
        run = HipRun(a).forward()
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        row = {"workload": spec, "P": P, "R": run.R, "inst_per_tile": round(run.R / tiles, 1)}
        ref = None
        for vname, flag in (("rows", n.OPT_BWD_ROWS), ("quad", n.OPT_BWD_QUAD), ("scan", n.OPT_BWD_SCAN)):
            run.debug = flag | 2
            for _ in range(3):
                g = run.backward(gC, gO)
            n.collect_stage_times()
            for _ in range(10):
                g = run.backward(gC, gO)
            t = n.collect_stage_times()
            row[vname + "_us"] = round(1e3 * t["blend_bwd"][0] / t["blend_bwd"][1], 1)
            row["preprocess_bwd_us"] = round(1e3 * t["preprocess_bwd"][0] / t["preprocess_bwd"][1], 1)
            st = torch.zeros(8, dtype=torch.int64, device="cuda:0")
            lib.surfel_debug_set_blend_stats(n.ptr(st))
            run.debug = flag
            run.backward(gC, gO)
            lib.surfel_debug_set_blend_stats(None)
            s = st.cpu().numpy()
            row[vname + "_useful_lane_frac"] = round(float(s[1]) / max(1, float(s[0])), 4)
            row[vname + "_wave_visits"] = int(s[2]); row[vname + "_unit_visits"] = int(s[3]); row["useful_pairs"] = int(s[1])
            row[vname + "_units_with_hits"] = int(s[4]); row[vname + "_s5"] = int(s[5]); row[vname + "_raw"] = [int(x) for x in s]
            if ref is None:
                ref = g
            elif vname == "quad":
                row["bit_identical"] = all(np.array_equal(ref[k], g[k]) for k in ref)
            else:      # the scan walk sums in another order: agreement to summation noise
                row["scan_min_cosine_vs_rows"] = round(min(float(np.dot(ref[k].ravel().astype(np.float64), g[k].ravel().astype(np.float64)) /
                                                              max(1e-300, np.linalg.norm(ref[k].astype(np.float64)) * np.linalg.norm(g[k].astype(np.float64)))) for k in ref), 9)
        run.debug = 2          # library default: variant chosen on the device
        for _ in range(3):
            run.backward(gC, gO)
        n.collect_stage_times()
        for _ in range(10):
            run.backward(gC, gO)
        t = n.collect_stage_times()
        row["auto_us"] = round(1e3 * t["blend_bwd"][0] / t["blend_bwd"][1], 1)
        for nm, flag in (("pbwd_thread_us", n.OPT_PBWD_THREAD), ("pbwd_coop_us", n.OPT_PBWD_COOP)):
            run.debug = 2 | flag
            for _ in range(3):
                run.backward(gC, gO)
            n.collect_stage_times()
            for _ in range(8):
                run.backward(gC, gO)
            t = n.collect_stage_times()
            row[nm] = round(1e3 * t["preprocess_bwd"][0] / t["preprocess_bwd"][1], 1)
        row["speedup"] = round(row["quad_us"] / row["rows_us"], 3)
        row["scan_vs_best_other"] = round(row["scan_us"] / min(row["rows_us"], row["quad_us"]), 3)
        print(json.dumps(row), flush=True)
        out.append(row)
        del run
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
