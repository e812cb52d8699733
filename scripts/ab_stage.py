#!/usr/bin/env python
"""A/B of library options inside ONE process on the same trainer state and the same views: per setting, the full-iteration time
(HIP events around a window of steps, no per-stage events) and the per-stage kernel times (a second window with debug = 2).

    python scripts/ab_stage.py trained,garden,C2 fwd_pipe=0 fwd_pipe=1 [--steps 30] [--reps 2]

Each setting is a comma-separated list of name=value pairs for surfel_set_option ("bwd_variant=3,tile_stream=0").  Settings alternate
(A B A B ...) so that clock drift hits both alike.  GPU only; prints one JSON line per workload.
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd"))
sys.path.insert(0, REPO)


def main():
    import torch
    import surfel_native as sn
    from helpers_bench import TRAINED_PRESETS, make_trainer, trained_trainer
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 30
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
    if "--steps" in sys.argv:
        args.remove(str(steps))
    if "--reps" in sys.argv:
        args.remove(str(reps))
    workloads, settings = args[0].split(","), args[1:]
    dev = torch.device("cuda:0")
    lib = sn.load()

    def apply(setting):
        for kv in filter(None, setting.split(",")):
            k, v = kv.split("=")
            if lib.surfel_set_option(k.encode(), int(v)) != 0:
                raise SystemExit("unknown option %r" % k)

    for wl in workloads:
        tr = trained_trainer(dev, wl)[0] if wl in TRAINED_PRESETS else make_trainer(dev, wl, n_views=8)
        for _ in range(12):
            tr.step()
        res = {s: {"ms_per_step": [], "stages": []} for s in settings}
        for rep in range(reps):
            for s in settings:
                apply(s)
                import random
                tr._rng = random.Random(1234 + rep); tr._stack = []      # same views in every setting's window
                for _ in range(4):
                    tr.step()
                tr._rng = random.Random(99 + rep); tr._stack = []
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(steps):
                    tr.step()
                e1.record()
                torch.cuda.synchronize()
                res[s]["ms_per_step"].append(round(e0.elapsed_time(e1) / steps, 4))
                tr._rng = random.Random(99 + rep); tr._stack = []
                tr.pipe.debug = 2
                sn.collect_stage_times()
                for _ in range(max(8, steps // 2)):
                    tr.step()
                torch.cuda.synchronize()
                st = sn.collect_stage_times()
                tr.pipe.debug = 0
                res[s]["stages"].append({k: round(v[0] / v[1], 4) for k, v in st.items()})
        out = {"workload": wl, "P": int(tr.model.P), "steps": steps}
        for s in settings:
            r = res[s]
            keys = r["stages"][0].keys()
            out[s] = {"ms_per_step": r["ms_per_step"], "stages_ms": {k: [x.get(k) for x in r["stages"]] for k in keys}}
        print(json.dumps(out), flush=True)
        del tr
        import diff_surfel_rasterization as dsr
        dsr.set_grad_arena(None)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
