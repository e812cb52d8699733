#!/usr/bin/env python
"""Where and when every blend workgroup ran (diagnostic build: python build.py --variant trace -DBLEND_TRACE).

    SURFEL_LIB=2d-gaussian-splatting_amd/lib/libsurfel_hip_trace.so SURFEL_OPTIONS=bwd_variant=0,bwd_tune=0 python scripts/wg_trace.py trained garden

Per workload: workgroups per CU, per-CU busy time, start / end distribution, the slowest tiles -> stdout (JSON) and
gpurun_out/wg_trace_<workload>.npz."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd"))
sys.path.insert(0, REPO)


def summarise(tr, name):
    t0, t1, hw, tn = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64), tr[:, 2], tr[:, 3]
    ok = t1 > 0
    t0, t1, hw, tn = t0[ok], t1[ok], hw[ok], tn[ok]
    if t0.size == 0:
        return {"kernel": name, "workgroups": 0}
    base = t0.min()
    s, e = (t0 - base) * 0.01, (t1 - base) * 0.01          # us
    hwid, xcc = (hw & 0xffffffff).astype(np.int64), (hw >> 32).astype(np.int64) & 0xf
    cu, sh, se = (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    n = (tn >> 32).astype(np.int64)
    heavy = n > 0
    dur = e - s
    per_cu = {}
    for c, d, h in zip(cuid, dur, heavy):
        a = per_cu.setdefault(int(c), [0, 0, 0.0])
        a[0] += 1; a[1] += int(h); a[2] += float(d) if h else 0.0
    wg = np.array([v[0] for v in per_cu.values()]); hv = np.array([v[1] for v in per_cu.values()]); bus = np.array([v[2] for v in per_cu.values()])
    order = np.argsort(-dur)[:8]
    return {"kernel": name, "workgroups": int(t0.size), "non_empty": int(heavy.sum()), "distinct_cus": len(per_cu),
            "span_us": round(float(e.max()), 1), "last_start_us": round(float(s.max()), 1), "last_start_non_empty_us": round(float(s[heavy].max()), 1),
            "wg_per_cu_min_med_max": [int(wg.min()), int(np.median(wg)), int(wg.max())],
            "non_empty_per_cu_min_med_max": [int(hv.min()), int(np.median(hv)), int(hv.max())],
            "sum_wg_us_per_cu_min_med_max": [round(float(x), 1) for x in (bus.min(), np.median(bus), bus.max())],
            "dur_us_non_empty_p50_p90_max": [round(float(np.percentile(dur[heavy], q)), 1) for q in (50, 90, 100)],
            "end_us_p50_p90_p99": [round(float(np.percentile(e[heavy], q)), 1) for q in (50, 90, 99)],
            "slowest": [{"n": int(n[i]), "start": round(float(s[i]), 1), "dur": round(float(dur[i]), 1), "cu": int(cuid[i])} for i in order],
            "dur_per_list_position_ns_p50": round(float(np.median(1e3 * dur[heavy] / np.maximum(1, n[heavy]))), 1),
            # wave 0 of every non-empty workgroup: shader-clock cycles by phase, loop iterations (visits)
            "wave0_cycles_stage_walk_barrier_sum": [int(tr[ok][heavy][:, k].astype(np.int64).sum()) for k in (4, 5, 6)],
            "wave0_iterations_sum": int(tr[ok][heavy][:, 7].astype(np.int64).sum()),
            "wave0_iterations_per_list_position": round(float(tr[ok][heavy][:, 7].astype(np.int64).sum()) / max(1, int(n[heavy].sum())), 3),
            "wave0_cycles_per_iteration_walk": round(float(tr[ok][heavy][:, 5].astype(np.int64).sum()) / max(1, float(tr[ok][heavy][:, 7].astype(np.int64).sum())), 1),
            "slowest_phase_cycles": [[int(tr[ok][i, k]) for k in (4, 5, 6, 7)] for i in order[:4]]}


def main():
    import torch
    import synthetic
    import surfel_native as sn
    from helpers_bench import TRAINED_PRESETS, make_trainer, trained_trainer
    dev = torch.device("cuda:0")
    lib = sn.load()
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    for wl in [x for x in sys.argv[1:] if not x.startswith("--")] or ["trained"]:
        if wl in TRAINED_PRESETS:
            tr, _ = trained_trainer(dev, wl)
        else:
            tr = make_trainer(dev, wl, n_views=8)
        for _ in range(12):
            tr.step()
        torch.cuda.synchronize()
        buf = torch.zeros(16 + 8 * 2 * 65536, dtype=torch.int64, device=dev)
        lib.surfel_debug_set_blend_stats(sn.ptr(buf))
        try:
            tr.step()
            torch.cuda.synchronize()
        finally:
            lib.surfel_debug_set_blend_stats(None)
        raw = buf[16:].view(2, 65536, 8).cpu().numpy().astype(np.uint64)
        np.savez_compressed(os.path.join(REPO, "gpurun_out", "wg_trace_%s.npz" % wl), fwd=raw[0], bwd=raw[1])
        if "--sub" in sys.argv:      # old forward kernel (fwd_pipe=0), trace build: staging sub-phases packed into words 4 and 6
            d = raw[0].astype(np.int64)
            okk = (d[:, 1] > 0) & ((d[:, 3] >> 32) > 0)
            d = d[okk]
            sub = {"ids_wait": d[:, 4] >> 40, "records_wait": (d[:, 4] >> 20) & 0xfffff, "lds_write_and_footprint": d[:, 4] & 0xfffff,
                   "ballots": d[:, 6] >> 24, "barriers": d[:, 6] & 0xffffff, "walk": d[:, 5]}
            print(json.dumps({"workload": wl, "kernel": "blend_fwd (batch-synchronous) wave-0 cycles, sum over non-empty workgroups",
                              **{k: int(v.sum()) for k, v in sub.items()}, "workgroups": int(d.shape[0]),
                              "batches": int(((d[:, 3] >> 32) + 255).sum() // 256)}), flush=True)
            raw[0][:, 4:7] = 0
        for k, name in ((0, "blend_fwd"), (1, "blend_bwd_rows")):
            out = summarise(raw[k], name)
            out["workload"] = wl
            print(json.dumps(out), flush=True)
        del tr
        import diff_surfel_rasterization as dsr
        dsr.set_grad_arena(None)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
