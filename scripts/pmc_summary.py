"""Summarise rocprofv3 --pmc counter_collection CSVs: per-kernel mean of each counter (surfel kernels + sort)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "surfel" not in k and "rocprim" not in k:
            continue
        k = k.split("(")[0].replace("surfel::", "")[:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
for k in sorted(acc):
    print("%-60s vgpr/sgpr/lds/scratch=%s" % (k, "/".join(meta[k])))
    for c, v in sorted(acc[k].items()):
        print("    %-24s mean %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
