"""Overrun hunt with guard pages (tests/guard_run.py): every device buffer the library is handed — inputs, outputs, the buffers of the
three allocator callbacks, the backward's scratch — ends exactly at the end of its mapped pages with reserved, UNMAPPED address space
behind it (HIP virtual-memory API, 4-KiB granularity).  A read or write past any buffer is a GPU memory fault on every box, not only
on those whose allocator happens to map small pages.  One configuration per process (a fault kills it)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(*args):
    p = subprocess.run([sys.executable, os.path.join(HERE, "guard_run.py")] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, "guard_run %s: rc %d\n%s\n%s" % (args, p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    return p.stdout


@pytest.mark.parametrize("pipe,variant", [(1, 0), (0, 1), (1, 3)])
def test_no_access_past_any_buffer(pipe, variant):
    """forward (pipelined | batch-synchronous blend) + backward (rows | quad | scan walk) on the exact and the capacity binning path"""
    out = _run(pipe, variant)
    assert out.count(" ok: R=") == 3 and "binning=1" in out, out


@pytest.mark.parametrize("variant", [0, 3])
def test_backward_of_an_overflowed_lazy_frame_touches_nothing(variant):
    """the backward of a lazily counted frame that REALLY overflowed its capacity, before the count is collected,
    with the gradient records at the end of their mapping; then the reported overflow and the redo."""
    assert "overflow ok" in _run("overflow", variant)


@pytest.mark.parametrize("pipe,variant", [(1, 0), (0, 3), (1, 3)])
def test_long_lists_stay_inside_their_buffers(pipe, variant):
    """lists of ~700 faint instances per tile: several staged batches per tile in both blend directions, the prefetch of the batch
    behind a tile's last one, with unmapped pages right behind every buffer"""
    out = _run("long", pipe, variant)
    assert out.count(" ok: R=") == 3, out
