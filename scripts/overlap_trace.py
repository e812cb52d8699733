#!/usr/bin/env python
"""Reads a rocprofv3 kernel_trace.csv and prints, for the last dispatches of one kernel, what ran concurrently with it (other queues).
    python scripts/overlap_trace.py <kernel_trace.csv> [kernel substring = adam_sh_kernel]"""
import csv
import sys

path, needle = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "adam_sh_kernel")
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("surfel::", "").replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0].split("<")[0][-48:], r.get("Queue_Id", "?")))
rows.sort()
hits = [k for k, r in enumerate(rows) if needle in r[2]]
print("dispatches of", needle, ":", len(hits), "| queues seen:", sorted(set(r[3] for r in rows)))
for k in hits[len(hits) // 3:len(hits) // 3 + 2]:
    s, e, name, q = rows[k]
    print("\n%s on queue %s: %.1f us" % (name, q, (e - s) / 1e3))
    lo = max(0, k - 6)
    for s2, e2, n2, q2 in rows[lo:k + 14]:
        ov = min(e, e2) - max(s, s2)
        print("   %-48s q%-3s start %+9.1f us  dur %8.1f us  overlap %7.1f us" % (n2, q2, (s2 - s) / 1e3, (e2 - s2) / 1e3, max(0, ov) / 1e3))
