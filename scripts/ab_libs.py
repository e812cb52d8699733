#!/usr/bin/env python
"""A/B of whole libraries (and option settings) on one GPU box: every (library, SURFEL_OPTIONS) arm runs `bench.py --workload W --quick` in
its own process, arms interleaved and repeated, trained states shared through /tmp/state_<W>.ply.
    python scripts/ab_libs.py "C2,trained,garden" 2 name=lib_tag[:opt=val,...] ...      (lib_tag "" = the product library)
Prints one JSON line per (workload, arm): ms_per_step and the blend kernels' event times, min / median over the repeats."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    wls, reps, arms = sys.argv[1].split(","), int(sys.argv[2]), sys.argv[3:]
    res = {}
    for wl in wls:
        state = ["--state", "/tmp/state_%s.ply" % wl] if wl in ("trained", "garden") else []
        if state and not os.path.exists(state[1]):
            subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", wl, "--quick", "--steps", "2", "--warmup", "1"] + state,
                           capture_output=True, text=True, timeout=600)
        for rep in range(reps):
            for arm in arms:
                name, _, spec = arm.partition("=")
                tag, _, opts = spec.partition(":")
                env = dict(os.environ)
                if tag and tag != "-":
                    env["SURFEL_LIB"] = os.path.join(REPO, "2d-gaussian-splatting_amd", "lib", "libsurfel_hip_%s.so" % tag)
                if opts:
                    env["SURFEL_OPTIONS"] = opts
                p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", wl, "--quick", "--steps", "40", "--warmup", "10"] + state,
                                   capture_output=True, text=True, timeout=900, env=env)
                lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
                if not lines:
                    print(json.dumps({"workload": wl, "arm": name, "error": p.stderr[-400:]}), flush=True)
                    continue
                d = json.loads(lines[-1])
                k = d["roofline"]["all_kernels_ms"]
                res.setdefault((wl, name), []).append((d["ms_per_step"], k.get("blend_bwd"), k.get("blend_fwd"), d["roofline"].get("kernel_ms"), k))
    for (wl, name), v in res.items():
        col = lambda i: sorted(x[i] for x in v if x[i] is not None)
        med = lambda c: c[len(c) // 2] if c else None
        print(json.dumps({"workload": wl, "arm": name, "runs": len(v), "ms_per_step_min_med": [col(0)[0], med(col(0))],
                          "blend_bwd_ms_min_med": [col(1)[0] if col(1) else None, med(col(1))], "blend_fwd_ms_min_med": [col(2)[0] if col(2) else None, med(col(2))],
                          "dominant_kernel_in_window_ms_min_med": [col(3)[0] if col(3) else None, med(col(3))],
                          "all_kernels_ms_min": {kk: min(x[4][kk] for x in v if kk in x[4]) for kk in v[0][4]}}), flush=True)


if __name__ == "__main__":
    main()
