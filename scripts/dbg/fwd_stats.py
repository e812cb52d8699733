import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "2d-gaussian-splatting_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
import surfel_native as n
from helpers import HipRun, scene_args
import test_gpu_parity as T
lib = n.load()
for kind in sys.argv[1:] or ["needles"]:
    a = scene_args(T._walk_scene(kind))
    st = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    out = {}
    for pipe in (0, 1):
        lib.surfel_set_option(b"fwd_pipe", pipe)
        r0 = HipRun(a).forward()
        lib.surfel_debug_set_blend_stats(n.ptr(st)); st.zero_()
        r1 = HipRun(a).forward()
        cnt = int(st.cpu().numpy()[6])
        lib.surfel_debug_set_blend_stats(None)
        out[pipe] = (r0.color.cpu().numpy(), r1.color.cpu().numpy(), cnt)
        print(kind, "pipe", pipe, "pairs", cnt, "stats image == plain image:", np.array_equal(out[pipe][0], out[pipe][1]))
    print("  plain images equal across kernels:", np.array_equal(out[0][0], out[1][0]), " stats images equal:", np.array_equal(out[0][1], out[1][1]))
lib.surfel_set_option(b"fwd_pipe", 1)
