"""Compile-time resource guard of the blend kernels (hipcc cross-compiles gfx950 here; no GPU): registers, LDS and scratch decide how
many waves a SIMD holds, and the blend kernels are bound by what their resident waves issue — a change that costs five VGPRs costs a
wave per SIMD without failing any numerical test (round 4: the list-splitting checkpoint store took blend_fwd_pipe_kernel from 96 to
113 VGPRs until its address computation was pinned to the store).  The limits are the occupancy steps the kernels were tuned to
(profiles/r04_isa_walk_loops.md), read from the assembly's own kernel descriptors."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "scripts"))

# kernel (demangled prefix) -> (max VGPRs, max LDS bytes): waves / SIMD by registers = 512 // VGPRs, workgroups / CU by LDS = 160 KB // LDS
LIMITS = {
    "blend_fwd_pipe_kernel<false>": (96, 32 * 1024),           # 5 waves / SIMD, 5 workgroups / CU
    "blend_fwd_kernel<false>": (96, 32 * 1024),
    "blend_bwd_rows_kernel<false, true>": (128, 40 * 1024),    # 4 waves / SIMD, 4 workgroups / CU (5 measured no faster); staging from the tile stream
    "blend_bwd_rows_kernel<false, false>": (128, 40 * 1024),   # ... staging by gather
    "blend_bwd_quad_kernel<false>": (96, 32 * 1024),
    "blend_bwd_scan_kernel<false, true>": (168, 160 * 1024 // 3),    # 3 waves / SIMD, 3 workgroups / CU
    "blend_bwd_scan_kernel<false, false>": (168, 160 * 1024 // 3),
}


@pytest.mark.parametrize("src", ["surfel_forward.hip", "surfel_backward.hip", "surfel_backward_scan.hip"])
def test_blend_kernels_keep_their_occupancy(src):
    import isa_count
    ks = isa_count.kernels(isa_count.assemble(src))
    seen = 0
    for name, (lines, md) in ks.items():
        dn = isa_count.demangle(name)
        if dn not in LIMITS:
            continue
        seen += 1
        vg, lds = LIMITS[dn]
        assert md["private_segment_fixed_size"] == 0, "%s spills %d B to scratch" % (dn, md["private_segment_fixed_size"])
        assert md["next_free_vgpr"] <= vg, "%s: %d VGPRs (limit %d)" % (dn, md["next_free_vgpr"], vg)
        assert md["group_segment_fixed_size"] <= lds, "%s: %d B of LDS (limit %d)" % (dn, md["group_segment_fixed_size"], lds)
        # the walk must still be there: an innermost loop with the per-visit arithmetic
        assert any(c.get("valu", 0) >= 60 for (_h, _d, _p, c, _dpp, _tr) in isa_count.loops(lines)), dn
    tags = {"surfel_forward.hip": ("_fwd_",), "surfel_backward.hip": ("_rows_", "_quad_"), "surfel_backward_scan.hip": ("_scan_",)}[src]
    assert seen == sum(1 for k in LIMITS if any(t in k for t in tags)), "a kernel of %s was renamed or is gone" % src
