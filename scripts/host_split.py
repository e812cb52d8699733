"""Where the host's time per training iteration goes: inside the library's C calls (HIP launches, allocator callbacks), waiting for
the device (surfel_forward_count), and in Python around them.  Wraps the ctypes entry points with timers; C2 workload.
Usage: python scripts/host_split.py [steps]"""
import os
import sys
import time

import torch

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd")); sys.path.insert(0, REPO)
import surfel_native as n  # noqa: E402
from helpers_bench import make_trainer  # noqa: E402

acc = {}


class Timed:
    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *a):
        t0 = time.perf_counter()
        r = self.fn(*a)
        d = acc.setdefault(self.name, [0.0, 0])
        d[0] += time.perf_counter() - t0; d[1] += 1
        return r


class Proxy:
    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)
        object.__setattr__(self, "_cache", {})

    def __getattr__(self, k):
        c = self._cache
        if k not in c:
            c[k] = Timed(k, getattr(self._lib, k))
        return c[k]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    tr = make_trainer(dev, "C2", 8)
    for _ in range(40):
        tr.step()
    torch.cuda.synchronize()
    n._lib = Proxy(n.load())
    acc.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    c_total = sum(v[0] for v in acc.values())
    wait = acc.get("surfel_forward_count", [0.0, 0])[0]
    print("per iteration: wall %.1f us (host loop %.1f us); inside C calls %.1f us of which waiting for the device %.1f us; Python around them %.1f us"
          % (t_all / steps * 1e6, t_host / steps * 1e6, c_total / steps * 1e6, wait / steps * 1e6, (t_host - c_total) / steps * 1e6))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print("  %-32s %6.1f us / iteration  (%d calls)" % (k, v[0] / steps * 1e6, v[1]))


if __name__ == "__main__":
    main()
