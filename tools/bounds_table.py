#!/usr/bin/env python
"""Which roofline bounds which kernel: joins a kernel-stats summary (average duration), the SQ counter summary (waves and VALU instructions per wave,
all dispatches of its run summed) and the HBM traffic summary into one table -- per kernel its time, the time its VALU wave instructions take at the rate
the chip issues them (tools/mb/valu_rate.hip: 0.57 per ns and SIMD for everything but add / xor; 1024 SIMDs), its HBM traffic (2 x FETCH_SIZE + WRITE_SIZE)
and the time that traffic takes at 8 TB/s.
    python tools/bounds_table.py profiles/r04_m_kernel_stats.csv profiles/r04_o_sq_lds_cfg3.csv <dispatches in the SQ run> profiles/r04_m_hbm_traffic.csv out.csv"""
import csv
import sys

ks, sq, ndisp, hbm, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
rows = lambda p: list(csv.DictReader(l for l in open(p) if not l.startswith("#")))
t = {r["kernel"]: float(r["average_us"]) for r in rows(ks)}
s = {r["kernel"]: (float(r["waves"]) / ndisp, float(r["valu_per_wave"])) for r in rows(sq)}
h = {r["kernel"]: (float(r["fetch_MB_x2"]) + float(r["write_MB"])) for r in rows(hbm)}
RATE = 0.57e9 * 1024            # VALU wave instructions per second on the chip
with open(out, "w") as f:
    f.write("# per kernel (cfg3, one dispatch): duration (rocprofv3), VALU wave instructions and the time they take at 0.57 per ns and SIMD on 1024 SIMDs, HBM traffic\n")
    f.write("# (2 x FETCH_SIZE + WRITE_SIZE) and the time it takes at 8 TB/s; *_pct = share of the kernel's duration.  tools/bounds_table.py\n")
    f.write("kernel,duration_us,valu_wave_instructions_M,valu_issue_us,valu_issue_pct,hbm_traffic_MB,hbm_8tbs_us,hbm_pct\n")
    for k in sorted(t, key=lambda k: -t[k]):
        if k not in s or t[k] < 15:
            continue
        w, v = s[k]
        vi = w * v
        iu = vi / RATE * 1e6
        hm = h.get(k, 0.0)
        hu = hm * 1e6 / 8e12 * 1e6
        f.write("%s,%.1f,%.1f,%.1f,%.0f,%.0f,%.1f,%.0f\n" % (k, t[k], vi / 1e6, iu, 100 * iu / t[k], hm, hu, 100 * hu / t[k]))
print(open(out).read())
