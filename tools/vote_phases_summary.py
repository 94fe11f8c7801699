#!/usr/bin/env python
"""tools/vote_phases_summary.py <dir with rocprofv3 csv outputs of tools/vote_phases.py runs> k0 k1 ... : per variant the k_vote duration and
counters (second dispatch of each pair), and the differences between consecutive variants = what each phase adds."""
import collections
import csv
import glob
import sys

d, ks = sys.argv[1], sys.argv[2:]
rows = collections.defaultdict(dict)          # variant -> metric -> value
for fn in sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)):
    v = [r for r in csv.DictReader(open(fn)) if r["Kernel_Name"].startswith("k_vote(")]
    v.sort(key=lambda r: int(r["Start_Timestamp"]))
    for i, k in enumerate(ks):
        r = v[2 * i + 1]
        rows[k]["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    break
for fn in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))    # dispatch id -> counter -> value
    for r in csv.DictReader(open(fn)):
        if r["Kernel_Name"].startswith("k_vote("):
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(per)
    for i, k in enumerate(ks):
        for c, val in per[ids[2 * i + 1]].items():
            rows[k][c] = val
cols = sorted({c for k in ks for c in rows[k]})
print("variant," + ",".join(cols))
prev = None
for k in ks:
    print(k + "," + ",".join("%.4g" % rows[k].get(c, float("nan")) for c in cols))
print("\n# added by the phase (difference to the variant in front)")
for k in ks:
    if prev is not None:
        print(k + "," + ",".join("%.4g" % (rows[k].get(c, 0) - rows[prev].get(c, 0)) for c in cols))
    prev = k
