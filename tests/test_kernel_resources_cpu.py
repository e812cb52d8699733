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


def test_loss_launches_keep_five_workgroups_per_cu():
    """Round 5 (scripts/loss_trace.hip): the fused loss launches were the sum of their workgroups' latencies over the resident slots; at
    <= 96 VGPRs and <= 32 KB of LDS five workgroups fit a CU (three before: 43 / 45 KB).  The horizontal pass keeps its results in
    registers across a barrier to get there, so a careless edit of the bodies shows up here, not in a numerical test."""
    import re
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd"))
    import build as B
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.FLAGS + ["-S", "--cuda-device-only", "-c", os.path.join(B.CSRC, "train_fused.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\w+)(.*?)\.end_amdhsa_kernel", text, re.S):
        if "train_loss_fwd_kernel" not in m.group(1) and "train_loss_bwd_kernel" not in m.group(1):
            continue
        seen += 1
        md = {k: int(re.search(r"\.amdhsa_%s (\d+)" % k, m.group(2)).group(1)) for k in ("next_free_vgpr", "group_segment_fixed_size", "private_segment_fixed_size")}
        assert md["private_segment_fixed_size"] == 0, (m.group(1), md)
        assert md["next_free_vgpr"] <= 96 and md["group_segment_fixed_size"] <= 32 * 1024, (m.group(1), md)
    assert seen == 2


@pytest.mark.parametrize("prog, flags", [("issue_probe.hip", []), ("loss_trace.hip", ["-std=c++17", "-ffp-contract=fast", "-I", os.path.join(REPO, "2d-gaussian-splatting_amd", "csrc")])])
def test_diagnostic_programs_build(prog, flags, tmp_path):
    """scripts/issue_probe.hip (what a SIMD issues, by instruction kind) and scripts/loss_trace.hip (per-workgroup timeline of the loss
    launches, built around the product's own kernel bodies) are compiled on the GPU box when they are used; here they only have to build."""
    import subprocess
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3"] + flags + [os.path.join(REPO, "scripts", prog), "-o", str(tmp_path / "prog")],
                          stderr=subprocess.DEVNULL)
