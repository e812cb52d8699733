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


def measure(tr, steps, label):
    for _ in range(40):
        tr.step()
    torch.cuda.synchronize()
    n._lib = Proxy(n.load() if not isinstance(n._lib, Proxy) else n._lib._lib)
    acc.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    t_host = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    c_total = sum(v[0] for v in acc.values())
    wait = acc.get("surfel_forward_count", [0.0, 0])[0]
    print("%s: per iteration wall %.1f us (host loop %.1f us, device span %.1f us); inside C calls %.1f us of which waiting for the device "
          "(surfel_forward_count) %.1f us; Python + torch.distributed around them %.1f us; host busy (loop - that wait) %.1f us = %.2f x the step"
          % (label, t_all / steps * 1e6, t_host / steps * 1e6, e0.elapsed_time(e1) / steps * 1e3, c_total / steps * 1e6, wait / steps * 1e6,
             (t_host - c_total) / steps * 1e6, (t_host - wait) / steps * 1e6, (t_host - wait) / t_all))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:8]:
        print("    %-34s %7.1f us / iteration  (%d calls)" % (k, v[0] / steps * 1e6, v[1]))


def main():
    """python scripts/host_split.py [steps] [--exchange]: --exchange adds the N > 1 step's host path (view-parallel schedule, early colour
    gather hook, asynchronous all-gather + all-reduce with stream-level waits, Adam in two parts) against RCCL with world size 1."""
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    steps = int(args[0]) if args else 200
    dev = torch.device("cuda:0")
    tr = make_trainer(dev, "C2", 8)
    measure(tr, steps, "single-GPU step")
    if "--exchange" in sys.argv:
        import torch.distributed as dist
        import surfel_trainer as TR
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        for early in (True, False):
            tr2 = TR.Trainer(tr.model, tr.cams, tr.opt, tr.pipe, rehearse_exchange=True)
            tr2.early_gather = early
            measure(tr2, steps, "N > 1 step rehearsed on RCCL (world 1), early gather %s" % ("on" if early else "off"))
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
