#!/usr/bin/env python
"""Static instruction accounting of the blend kernels, from hipcc's own assembly (no GPU needed): per kernel the registers / LDS /
occupancy the compiler reports, and per LOOP the instruction mix of one trip — VALU, DPP among them, transcendental, SALU, branches,
LDS, vector memory, waits.  The blend kernels are bound by the instructions their waves issue (profiles/r04_wg_trace.md), so the
innermost walk loop's count per trip IS the cost of a visit; this is the table a change to a walk is judged by before it is timed.

    python scripts/isa_count.py [kernel-name-substring ...]      -> markdown on stdout (profiles/r04_isa_walk_loops.md)
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd"))
import build as B      # the flags the product is built with

FILES = ["surfel_forward.hip", "surfel_backward.hip", "surfel_backward_scan.hip"]
TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op in ("s_waitcnt", "s_nop", "s_sleep"):
        return "wait"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def assemble(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.FLAGS + B.EXTRA.get(src, []) + ["-S", "--cuda-device-only", "-c", os.path.join(B.CSRC, src), "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def kernels(text):
    """name -> (body lines, metadata dict)"""
    res = {}
    for m in re.finditer(r"^(_ZN6surfel\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        res[m.group(1)] = [m.group(2).split("\n"), {}]
    for m in re.finditer(r"\.amdhsa_kernel (\w+)(.*?)\.end_amdhsa_kernel", text, re.S):
        if m.group(1) in res:
            md = res[m.group(1)][1]
            for key in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size", "accum_offset"):
                mm = re.search(r"\.amdhsa_%s (\d+)" % key, m.group(2))
                if mm:
                    md[key] = int(mm.group(1))
    return res


def demangle(name):
    try:
        return subprocess.check_output(["c++filt", name], text=True).strip().replace("surfel::", "").replace("void ", "").replace("(BlendFwdArgs)", "").replace("(BlendBwdArgs)", "")
    except Exception:
        return name


def loops(lines):
    """[(header label, depth, parent header or None, {class: count}, dpp, transcendental)] — a block belongs to the loop named in its
    label's comment (`in Loop: Header=BBx_y Depth=d`, or `=>This ... Loop Header: Depth=d` for the header block itself)."""
    cur = None
    table = {}
    order = []
    parent = {}
    for i, ln in enumerate(lines):
        s = ln.strip()
        lab = re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)\s*(;.*)?$", s)
        if lab:
            # the label's comment may run over the following comment-only lines (parents first, then "=> This [Inner] Loop Header")
            c = lab.group(2) or ""
            j = i + 1
            while j < len(lines) and lines[j].strip().startswith(";") and not re.match(r"^; %bb\.\d+:", lines[j].strip()):
                c += " " + lines[j].strip()
                j += 1
            hdr = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", c)
            inl = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", c)
            if hdr:
                cur = lab.group(1).strip(".:").replace("LBB", "BB")
                if cur not in table:
                    table[cur] = [int(hdr.group(1)), {}, 0, 0]
                    order.append(cur)
                for pm in re.finditer(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", c):
                    if int(pm.group(2)) == int(hdr.group(1)) - 1:
                        parent[cur] = pm.group(1)
            elif inl:
                cur = inl.group(1)
                if cur not in table:
                    table[cur] = [int(inl.group(2)), {}, 0, 0]
                    order.append(cur)
            else:
                cur = None
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        if cur is None:
            continue
        k = classify(op)
        t = table[cur]
        t[1][k] = t[1].get(k, 0) + 1
        if "_dpp" in op or " row_" in s or "quad_perm" in s:
            t[2] += 1
        if op.startswith(TRANS):
            t[3] += 1
    return [(h, table[h][0], parent.get(h), table[h][1], table[h][2], table[h][3]) for h in order]


def occupancy(vgpr, lds):
    """waves per SIMD by registers (512 VGPRs per lane and SIMD, allocation granule 8) and workgroups per CU by LDS (160 KB)"""
    g = (vgpr + 7) // 8 * 8
    return min(8, 512 // max(g, 8)), (160 * 1024) // lds if lds else 99


def main():
    want = sys.argv[1:]
    print("# Static instruction accounting of the blend kernels (`python scripts/isa_count.py`; hipcc's assembly of the product's flags)\n")
    print("Per loop, the instructions of ONE trip through the blocks that belong to it (nested loops excluded: they have their own row).")
    print("valu includes the DPP and transcendental ones listed beside it; `wait` = s_waitcnt + s_nop.\n")
    for src in FILES:
        ks = kernels(assemble(src))
        for name, (lines, md) in ks.items():
            dn = demangle(name)
            if want and not any(w in dn for w in want):
                continue
            if name.split("kernel")[1].startswith("ILb1"):
                continue      # the STATS instances
            vg, lds = md.get("next_free_vgpr", 0), md.get("group_segment_fixed_size", 0)
            wv, wg = occupancy(vg, lds)
            print("## `%s` (%s)\n" % (dn, src))
            print("VGPRs %d, LDS %d B, scratch %d B -> %d waves / SIMD by registers, %d workgroups / CU by LDS\n" % (vg, lds, md.get("private_segment_fixed_size", 0), wv, wg))
            print("| loop | depth | inside | valu | of which dpp | transcendental | salu | branch | lds | vmem | wait | barrier | all |")
            print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
            for h, d, par, c, dpp, tr in loops(lines):
                tot = sum(c.values())
                if tot < 8:
                    continue
                print("| %s | %d | %s | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d |" % (h, d, par or "", c.get("valu", 0), dpp, tr, c.get("salu", 0), c.get("branch", 0), c.get("lds", 0), c.get("vmem", 0), c.get("wait", 0), c.get("barrier", 0), tot))
            print()


if __name__ == "__main__":
    main()
