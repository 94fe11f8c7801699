#!/usr/bin/env python
"""Summarise the two SQ counter passes of tools/pmc_round.sh into profiles/<tag>_sq_counters.csv.
    python tools/sq_summary.py gpurun_out/r01g profiles/r01_g "<command note>"
SQ_* cycle counters tick once per 4 clocks; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import glob
import sys


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void "):
                k = k[5:]
            if k.startswith("k_"):
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc


def main():
    src, dst, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    a, b = load(src + "_sq1"), load(src + "_sq2")
    with open(dst + "_sq_counters.csv", "w") as f:
        f.write("# SQ counters per kernel, rocprofv3 --pmc (2 passes, --kernel-trace only): %s\n" % note)
        f.write("# derived: ms = GRBM_GUI_ACTIVE/8/2.4e6 ; valu_busy_pct = SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * cycles) ; sca_busy_pct likewise ; "
                "wait_pct = SQ_WAIT_ANY / SQ_WAVE_CYCLES ; occupancy = SQ_WAVE_CYCLES*4 / (1024 * cycles) waves per SIMD\n")
        f.write("kernel,waves,ms,valu_per_wave,salu_per_wave,vmem_rd_per_wave,vmem_wr_per_wave,lds_per_wave,valu_busy_pct,sca_busy_pct,wait_pct,occupancy_waves_per_simd\n")
        for k in sorted(a, key=lambda k: -b[k].get("GRBM_GUI_ACTIVE", 0)):
            x, y = a[k], b[k]
            wv = x.get("SQ_WAVES", 0) or 1
            cyc = y.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
            f.write("%s,%d,%.3f,%.0f,%.0f,%.1f,%.1f,%.1f,%.1f,%.1f,%.1f,%.2f\n" % (
                k, wv, cyc / 2.4e6, x.get("SQ_INSTS_VALU", 0) / wv, x.get("SQ_INSTS_SALU", 0) / wv, x.get("SQ_INSTS_VMEM_RD", 0) / wv,
                x.get("SQ_INSTS_VMEM_WR", 0) / wv, x.get("SQ_INSTS_LDS", 0) / wv, 100 * y.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * cyc),
                100 * y.get("SQ_ACTIVE_INST_SCA", 0) * 4 / (1024 * cyc), 100 * y.get("SQ_WAIT_ANY", 0) / (x.get("SQ_WAVE_CYCLES", 0) or 1),
                x.get("SQ_WAVE_CYCLES", 0) * 4 / (1024 * cyc)))
    print(open(dst + "_sq_counters.csv").read())


if __name__ == "__main__":
    main()
