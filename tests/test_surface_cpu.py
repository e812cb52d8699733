"""CPU tests of the library's public surface and of the conventions round 5 introduced: the documented options are exactly the accepted
ones, removed ones are refused, the tile stream's bit permutation matches the thread -> pixel map it serves, the bench's box-probe
model is the identity on its reference box, the product header stays the product surface, and the design document a design document."""
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPTIONS = {"cull": 1, "tile_depth_sort": 1, "capacity_binning": 1, "large_sort": 2, "tile_order": 0, "fwd_pipe": 1, "tile_stream": 1, "bwd_variant": 2}
REMOVED = ["bwd_split", "bwd_tune", "scan_large", "host_total", "fat_sort", "pbwd_coop"]


def _header():
    return open(os.path.join(REPO, "include", "surfel_hip.h")).read()


@pytest.mark.parametrize("name", sorted(OPTIONS))
def test_documented_option_is_accepted_with_its_default(name):
    """every option include/surfel_hip.h documents is accepted by surfel_set_option, with the default the header states"""
    import surfel_native
    lib = surfel_native.load()
    m = re.search(r'\*\s+"%s"\s+(-?\d+)\s' % name, _header())
    assert m, "option %r is not documented in include/surfel_hip.h" % name
    assert int(m.group(1)) == OPTIONS[name]
    assert lib.surfel_set_option(name.encode(), OPTIONS[name]) == 0


def test_the_header_documents_exactly_the_accepted_options():
    """<= 8 documented options, none undocumented: the names in the header's option table are the library's"""
    import surfel_native
    lib = surfel_native.load()
    hdr = _header()
    blk = hdr[hdr.index("Process-wide defaults of the eight switches"):hdr.index("int surfel_set_option")]
    names = re.findall(r'^ \*   "([a-z_]+)"', blk, flags=re.M)
    assert sorted(names) == sorted(OPTIONS) and len(names) <= 8
    src = open(os.path.join(REPO, "2d-gaussian-splatting_amd", "csrc", "surfel_api.hip")).read()
    accepted = re.findall(r'std::strcmp\(name, "([a-z_]+)"\) == 0', src)
    assert sorted(accepted) == sorted(OPTIONS)
    assert lib.surfel_set_option(b"no_such_option", 1) < 0


@pytest.mark.parametrize("name", REMOVED)
def test_removed_options_are_refused(name):
    """options of earlier rounds that no longer exist fail loudly instead of being silently ignored"""
    import surfel_native
    lib = surfel_native.load()
    assert lib.surfel_set_option(name.encode(), 1) < 0
    assert name not in _header()


def test_per_call_bits_do_not_collide():
    """the SURFEL_OPT_* bits of the `debug` word are disjoint, lie above the debug-mode byte, and surfel_native mirrors them"""
    import surfel_native as n
    hdr = _header()
    single = {k: int(v) for k, v in re.findall(r"#define (SURFEL_OPT_[A-Z_]+)\s+\(1 << (\d+)\)", hdr)}
    fields = {"SURFEL_OPT_TILE_SORT": (9, 2), "SURFEL_OPT_TILE_ORDER": (19, 2)}
    used = {}
    for k, b in single.items():
        assert b >= 8 and b not in used, (k, used.get(b))
        used[b] = k
    for k, (lo, w) in fields.items():
        assert re.search(r"#define %s\(m\)\s+\(\(\(\(m\) \+ 1\) & 3\) << %d\)" % (k, lo), hdr), k
        for b in range(lo, lo + w):
            assert b not in used, (k, used.get(b))
            used[b] = k
    for py, c in (("OPT_NO_CULL", "SURFEL_OPT_NO_CULL"), ("OPT_BWD_ROWS", "SURFEL_OPT_BWD_ROWS"), ("OPT_BWD_QUAD", "SURFEL_OPT_BWD_QUAD"),
                  ("OPT_BWD_SCAN", "SURFEL_OPT_BWD_SCAN"), ("OPT_BWD_GATHER", "SURFEL_OPT_BWD_GATHER"), ("OPT_EXACT_BINNING", "SURFEL_OPT_EXACT_BINNING"),
                  ("OPT_LAZY_COUNT", "SURFEL_OPT_LAZY_COUNT"), ("OPT_TILE_CUTS", "SURFEL_OPT_TILE_CUTS"), ("OPT_ZERO_RECORDS", "SURFEL_OPT_ZERO_RECORDS")):
        assert getattr(n, py) == 1 << single[c], py
    assert n.opt_tile_sort(2) == 3 << 9 and n.opt_tile_order(1) == 2 << 19


def test_tile_stream_bit_permutation_matches_the_thread_pixel_map():
    """csrc/surfel_common.h: subtile_bits_to_rows turns blend_fwd's sub-tile bits (bit 4 by + bx) into blend_bwd's row order (bit 4 w + r:
    DPP row r of wave w) with three masks; restated here from thread_pixel (wave w -> quad (w & 1, w >> 1), row r -> sub-tile
    (r & 1, r >> 1) of the quad) and checked on all 65 536 masks — and against the masks in the source."""
    src = open(os.path.join(REPO, "2d-gaussian-splatting_amd", "csrc", "surfel_common.h")).read()
    m = re.search(r"return \(m & (0x[0-9A-Fa-f]+)u\) \| \(\(m & (0x[0-9A-Fa-f]+)u\) << 2\) \| \(\(m & (0x[0-9A-Fa-f]+)u\) >> 2\);", src)
    assert m, "subtile_bits_to_rows changed its form"
    keep, up, down = (int(x, 16) for x in m.groups())
    row_of_sub = {}
    for tid in range(0, 256, 16):      # one thread per DPP row
        w, r = tid >> 6, (tid >> 4) & 3
        bx, by = ((w & 1) << 1) | (r & 1), (w & 2) | (r >> 1)
        row_of_sub[4 * by + bx] = tid >> 4
    assert sorted(row_of_sub) == list(range(16)) and sorted(row_of_sub.values()) == list(range(16))
    x = np.arange(1 << 16, dtype=np.uint32)
    got = (x & keep) | ((x & up) << 2) | ((x & down) >> 2)
    want = np.zeros_like(x)
    for sub, row in row_of_sub.items():
        want |= ((x >> sub) & 1) << row
    assert np.array_equal(got, want)


def test_box_probe_normalisation_is_the_identity_on_its_reference_box():
    """helpers_bench: the step-split weights add up to one and a box that probes like the reference box gets slowdown 1"""
    import helpers_bench as hb
    assert abs(sum(hb.STEP_SPLIT.values()) - 1.0) < 1e-9 and set(hb.STEP_SPLIT) == {"blend", "latency", "hbm"}
    r = hb.BOX_REF
    s = (hb.STEP_SPLIT["blend"] * r["blend_mix_Mvisits_per_s"] / r["blend_mix_Mvisits_per_s"] + hb.STEP_SPLIT["latency"] * r["sort_512k_us"] / r["sort_512k_us"]
         + hb.STEP_SPLIT["hbm"] * r["hbm_copy_GBps"] / r["hbm_copy_GBps"])
    assert abs(s - 1.0) < 1e-12
    src = open(os.path.join(REPO, "helpers_bench.py")).read()
    assert 'STEP_SPLIT["blend"] * r["blend_mix_Mvisits_per_s"]' in src and "slowdown_vs_reference_box_by_probes" in src
    # (round 6: the headline carries the raw step only — no normalisation by one of the product's own kernels)
    assert "ms_per_step_normalised" not in open(os.path.join(REPO, "bench.py")).read()


def test_the_product_header_is_the_product_surface():
    """include/surfel_hip.h = entry points + options (<= 120 lines, no probes, no profile file names, no round history);
    diagnostics live in include/surfel_debug.h"""
    hdr = _header()
    assert len(hdr.splitlines()) <= 120, len(hdr.splitlines())
    for word in ("surfel_debug_", "profiles/", "round ", "surfel_last_stage_ms", "box_probe"):
        assert word not in hdr, word
    dbg = open(os.path.join(REPO, "include", "surfel_debug.h")).read()
    for fn in ("surfel_debug_box_probe", "surfel_debug_latency_probe", "surfel_debug_set_blend_stats", "surfel_debug_sort_pairs", "surfel_collect_stage_ms"):
        assert fn in dbg, fn


def test_design_document_stays_a_design_document():
    """DESIGN.md <= 20 KB with the sections the task names; the history lives in CHANGELOG.md"""
    d = open(os.path.join(REPO, "DESIGN.md")).read()
    assert len(d.encode()) <= 20 * 1024, len(d.encode())
    for head in ("## 1. The path and its boundary", "## 2. Oracle and parity", "## 3. Data layout in HBM", "## 4. Kernels", "## 5. Measurement", "## 6. Multi-GPU",
                 "## 8. Out of scope"):
        assert head in d, head
    assert "parity unpinned" in d
    assert os.path.exists(os.path.join(REPO, "CHANGELOG.md"))


def test_no_timing_instrumentation_in_the_product_kernels():
    """no #ifdef timing / trace blocks inside the product kernels: the diagnostics of rounds 2 - 4 are gone from csrc/"""
    csrc = os.path.join(REPO, "2d-gaussian-splatting_amd", "csrc")
    for f in os.listdir(csrc):
        s = open(os.path.join(csrc, f)).read()
        for macro in ("BLEND_TRACE", "ROWS_TIMING", "SCAN_TIMING", "SCAN_PAD_LDS", "TRACE_TM(", "s_memtime"):
            if f == "box_probe.hip" and macro == "s_memtime":
                continue
            assert macro not in s, (f, macro)
