#!/usr/bin/env python
"""Do two builds of the library give the same BITS?  Every (scene, walk) runs forward + backward under each library in its own
process (SURFEL_LIB); the gradient tensors' SHA-1 digests are compared.
    python scripts/lib_identity.py <lib_tag_a> <lib_tag_b> [walk ...]      (tag "-" = the product library; walks: rows quad scan auto)"""
import hashlib
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "2d-gaussian-splatting_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
SCENES = {"plain": (30000, 400, 304, dict(seed=3, px_radius=4.0)), "wide": (20000, 320, 240, dict(seed=5, px_radius=9.0)),
          "long_lists": (60000, 160, 128, dict(seed=12, px_radius=6.0, z_near=1.0, z_far=9.0)), "C1": "C1"}


def child(walks):
    import numpy as np
    import surfel_native as n
    import synthetic
    from helpers import HipRun, scene_args
    flags = {"rows": n.OPT_BWD_ROWS, "quad": n.OPT_BWD_QUAD, "scan": n.OPT_BWD_SCAN, "auto": 0}
    out = {}
    for name, spec in SCENES.items():
        sc = synthetic.make_config(spec, seed=2) if isinstance(spec, str) else synthetic.make_scene(spec[0], spec[1], spec[2], **spec[3])
        if name == "long_lists":
            sc["opacities"] = np.full_like(sc["opacities"], 0.015)
        a = scene_args(sc)
        rng = np.random.default_rng(4)
        gC = rng.normal(size=(3, a["H"], a["W"])).astype(np.float32); gO = rng.normal(size=(7, a["H"], a["W"])).astype(np.float32)
        run = HipRun(a).forward()
        out[name + "/images"] = hashlib.sha1(run.color.cpu().numpy().tobytes() + run.others.cpu().numpy().tobytes()).hexdigest()
        for w in walks:
            run.debug = flags[w]
            g = run.backward(gC, gO)
            for k in sorted(g):
                out["%s/%s/%s" % (name, w, k)] = hashlib.sha1(g[k].tobytes()).hexdigest()
    print("DIGESTS " + json.dumps(out))


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2:])
    tags, walks = sys.argv[1:3], (sys.argv[3:] or ["rows", "scan", "auto"])
    res = []
    for tag in tags:
        env = dict(os.environ)
        if tag != "-":
            env["SURFEL_LIB"] = os.path.join(REPO, "2d-gaussian-splatting_amd", "lib", "libsurfel_hip_%s.so" % tag)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + walks, capture_output=True, text=True, env=env, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("DIGESTS ")]
        if not line:
            print("lib %r failed:\n%s" % (tag, p.stderr[-1500:]))
            return 1
        res.append(json.loads(line[0][8:]))
    diff = [k for k in res[0] if res[0][k] != res[1].get(k)]
    print(json.dumps({"libs": tags, "tensors_compared": len(res[0]), "identical": not diff, "different": diff}))
    return 0 if not diff else 2


if __name__ == "__main__":
    sys.exit(main())
