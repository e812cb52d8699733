"""Per-kernel roofline table from the committed profile artefacts: python scripts/roofline_table.py <tag> [workload]
Reads profiles/<tag>_<wl>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python bench.py`) and profiles/<tag>_<wl>_pmc.json
(separate FETCH_SIZE / WRITE_SIZE / SQ passes) and writes profiles/<tag>_<wl>_roofline.md:
  HBM GB/s = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch / average duration  vs 8 000 GB/s (MI355X_MICROARCH.md)
  VALU     = SQ_INSTS_VALU per launch / average duration                      vs 1 228.9 G wave-instructions/s (157.3 TFLOP/s / 128)"""
import csv, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]; wl = sys.argv[2] if len(sys.argv) > 2 else "C2"
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(REPO, "profiles", "%s_%s_kernel_stats.csv" % (tag, wl))))}
pmc = json.load(open(os.path.join(REPO, "profiles", "%s_%s_pmc.json" % (tag, wl))))
rows = []
for k, c in pmc.items():
    base = k.split("<")[0]
    st = stats.get(k) or stats.get(base)
    if not st or "HBM_bytes_per_launch" not in c:
        continue
    us = float(st["AverageNs"]) / 1e3
    gbs = c["HBM_bytes_per_launch"] / (us * 1e-6) / 1e9
    valu = c.get("SQ_INSTS_VALU", 0.0) / (us * 1e-6) / 1e9
    # (GRBM_GUI_ACTIVE also counts the dispatch's ramp before / after the kernel's own timestamps: a few thousand cycles, i.e. the ratio is
    # only meaningful for kernels that run for tens of microseconds)
    clk = c.get("eff_clock_GHz") if us >= 30.0 else None
    # VALU issue against the clock the kernel actually sustained: 1024 SIMDs x clk / 2 cycles per wave64 fp32 instruction
    fv_sus = (valu / (1024 * clk / 2.0)) if clk else None
    rows.append((float(st["TotalDurationNs"]), base, int(st["Calls"]), us, c["HBM_bytes_per_launch"] / 1e6, gbs, gbs / 8000.0, valu, valu / 1228.9, clk, fv_sus))
rows.sort(reverse=True)
out = ["# Per-kernel roofline, %s / %s (MI355X: HBM 8 000 GB/s, fp32 VALU issue 1 228.9 G wave-inst/s)" % (tag, wl), "",
       "eff_clock_GHz = GRBM_GUI_ACTIVE / kernel wall time (own --pmc pass; kernels of >= 30 us only); 'of sustained' = VALU rate / (1024 SIMDs x eff clock / 2),",
       "i.e. against the NOMINAL 2 cycles per wave64 fp32 instruction.  Measured per instruction kind (scripts/issue_probe.hip, profiles/r05_issue_probe.md):",
       "fma / add / mul / mov / and 2.5 cycles, every DPP form, min / max, compares, selects, shifts and SGPR-operand forms 4.3, transcendentals 8.2 — a blend",
       "kernel at 0.45 of the nominal figure runs at ~0.75 of what its own instruction mix can issue.", "",
       "| kernel | launches | avg µs | HBM MB / launch | GB/s | frac of HBM peak | VALU G wave-inst/s | frac of VALU peak (2.4 GHz) | eff_clock_GHz | frac of VALU issue at the sustained clock | bound |",
       "|---|---|---|---|---|---|---|---|---|---|---|"]
for _, k, n, us, mb, gbs, fh, valu, fv, clk, fvs in rows:
    bound = "HBM" if fh >= 0.4 and fh >= fv else ("VALU issue" if fv >= 0.3 else "latency / launch")
    out.append("| `%s` | %d | %.1f | %.1f | %.0f | %.2f | %.0f | %.2f | %s | %s | %s |" % (k, n, us, mb, gbs, fh, valu, fv, "%.2f" % clk if clk else "-", "%.2f" % fvs if fvs else "-", bound))
path = os.path.join(REPO, "profiles", "%s_%s_roofline.md" % (tag, wl))
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(out))
