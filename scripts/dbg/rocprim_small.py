"""Diagnostic: rocprim::radix_sort_pairs behind surfel_debug_sort_pairs at small sizes / narrow key fields (large_sort = 3)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd"))
import torch
import surfel_native as nat
lib = nat.load()
dev = torch.device("cuda:0")
lib.surfel_set_option(b"large_sort", 3)
for n in (1000, 4096, 4097, 5000, 70000):
    for lo, hi in ((0, 32), (0, 1), (7, 8), (31, 32), (24, 32), (30, 32)):
        rng = np.random.default_rng(n + lo)
        keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        k = torch.from_numpy(keys.view(np.int32)).to(dev); v = torch.from_numpy(vals.view(np.int32)).to(dev)
        alloc = nat.TorchAllocator(dev)
        rc = lib.surfel_debug_sort_pairs(alloc.cb, None, nat.ptr(k), nat.ptr(v), n, lo, hi, nat.current_stream_ptr(dev))
        torch.cuda.synchronize()
        field = (keys >> np.uint32(lo)) & np.uint32((1 << (hi - lo)) - 1 if hi - lo < 32 else 0xffffffff)
        order = np.argsort(field, kind="stable")
        ok = np.array_equal(v.cpu().numpy().view(np.uint32), vals[order])
        print(n, (lo, hi), "rc", rc, "ok" if ok else "WRONG", flush=True)
