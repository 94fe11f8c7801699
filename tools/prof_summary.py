#!/usr/bin/env python
"""(rocpd top_kernels view reports microseconds; the CSV reports nanoseconds.)
Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db or *_kernel_stats.csv) for profiles/:
only the engine's own kernels (k_*), with calls / total / average duration in microseconds."""
import csv
import glob
import sqlite3
import sys


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    return [(r[0], int(r[1]), float(r[2]), float(r[3]), float(r[4])) for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels")]


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    rows = rows_from_db(src) if src.endswith(".db") else rows_from_csv(src)
    rows = [((r[0][5:] if r[0].startswith("void ") else r[0]),) + tuple(r[1:]) for r in rows]      # templated kernels are reported as "void k_x<..>(...)"
    ours = [r for r in rows if r[0].startswith("k_")]
    tot = sum(r[2] for r in ours)
    with open(dst, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (engine kernels only); durations in microseconds\n")
        if note:
            f.write("# %s\n" % note)
        f.write("kernel,calls,total_us,average_us,share_of_engine_pct\n")
        for r in sorted(ours, key=lambda r: -r[2]):
            f.write("%s,%d,%.1f,%.1f,%.2f\n" % (r[0].split("(")[0], r[1], r[2], r[3], 100.0 * r[2] / tot))
    print(open(dst).read())


if __name__ == "__main__":
    main()
