#!/usr/bin/env python
"""Summarise the three SQ/LDS counter passes of tools/lds_round.sh into profiles/<tag>_sq_lds_<workload>.csv:
    python tools/lds_summary.py gpurun_out/r03 profiles/r03 cfg3 "<note>"
Per kernel (all dispatches of the run summed): instruction counts per wave, busy / wait shares, and the LDS figures BASELINE.json's
north_star asks for: LDS instructions, cycles the LDS index unit was active, bank-conflict cycles, address-conflict (atomic collision) cycles.
  lds_conflict_pct = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE  (share of LDS-active cycles lost to bank conflicts: the "hit rate" is 100 - that)"""
import collections, csv, glob, sys


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("k_"):
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc


def main():
    src, dst, wl = sys.argv[1], sys.argv[2], sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    a, b, c = (load("%s_sq%d_%s" % (src, i, wl)) for i in (1, 2, 3))
    rows = sorted(a, key=lambda k: -a[k].get("SQ_BUSY_CYCLES", 0))
    out = dst + "_sq_lds_%s.csv" % wl
    with open(out, "w") as f:
        f.write("# SQ + LDS counters per kernel, rocprofv3 --pmc, three separate passes with --kernel-trace only (tools/lds_round.sh): %s\n" % note)
        f.write("# *_pct_of_wave_cycles: share of SQ_WAVE_CYCLES (quad-cycles a wave is resident); lds_conflict_pct = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
                "lds_addr_conflict_pct = SQ_LDS_ADDR_CONFLICT / SQ_LDS_IDX_ACTIVE (atomics of one wave on one address)\n")
        f.write("kernel,waves,valu_per_wave,salu_per_wave,vmem_rd_per_wave,vmem_wr_per_wave,lds_per_wave,active_valu_pct_of_wave_cycles,active_scalar_pct,active_lds_pct,wait_any_pct,wait_inst_lds_pct,"
                "lds_idx_active_cycles,lds_bank_conflict_cycles,lds_addr_conflict_cycles,lds_conflict_pct,lds_addr_conflict_pct,lds_atomic_return,lds_unaligned_stall\n")
        for k in rows:
            x, y, z = a[k], b[k], c[k]
            wv = x.get("SQ_WAVES", 0) or 1
            wc = x.get("SQ_WAVE_CYCLES", 0) or 1
            wc3 = z.get("SQ_WAVE_CYCLES", 0) or 1
            idx = z.get("SQ_LDS_IDX_ACTIVE", 0)
            f.write("%s,%d,%.0f,%.0f,%.1f,%.1f,%.1f,%.1f,%.1f,%.1f,%.1f,%.1f,%.0f,%.0f,%.0f,%.1f,%.1f,%.0f,%.0f\n" % (
                k, wv, x.get("SQ_INSTS_VALU", 0) / wv, x.get("SQ_INSTS_SALU", 0) / wv, x.get("SQ_INSTS_VMEM_RD", 0) / wv, x.get("SQ_INSTS_VMEM_WR", 0) / wv, x.get("SQ_INSTS_LDS", 0) / wv,
                100 * y.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * y.get("SQ_ACTIVE_INST_SCA", 0) / wc, 100 * y.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * y.get("SQ_WAIT_ANY", 0) / wc,
                100 * z.get("SQ_WAIT_INST_LDS", 0) / wc3, idx, y.get("SQ_LDS_BANK_CONFLICT", 0), y.get("SQ_LDS_ADDR_CONFLICT", 0),
                100 * y.get("SQ_LDS_BANK_CONFLICT", 0) / idx if idx else 0.0, 100 * y.get("SQ_LDS_ADDR_CONFLICT", 0) / idx if idx else 0.0,
                z.get("SQ_LDS_ATOMIC_RETURN", 0), z.get("SQ_LDS_UNALIGNED_STALL", 0)))
    # the per-column vote's figures for bench.py's roofline.lds (static, stamped with the sources' hash)
    import json, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from hbm_summary import csrc_sha16
    k = "k_vote" if "k_vote" in a else None
    if k:
        x, y, z = a[k], b[k], c[k]
        idx = z.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0
        wc, wc3 = x.get("SQ_WAVE_CYCLES", 0) or 1.0, z.get("SQ_WAVE_CYCLES", 0) or 1.0
        doc = {"tag": dst.rsplit("/", 1)[-1], "workload": wl, "note": note, "csrc_sha16": csrc_sha16(),
               "k_vote": {"what": "k_vote, rocprofv3 --pmc SQ_* (three separate passes, static: not a measurement of this run)", "source": os.path.basename(out),
                          "lds_bank_conflict_pct": round(100 * y.get("SQ_LDS_BANK_CONFLICT", 0) / idx, 1), "lds_addr_conflict_pct": round(100 * y.get("SQ_LDS_ADDR_CONFLICT", 0) / idx, 1),
                          "lds_conflict_free_pct": round(100 - 100 * y.get("SQ_LDS_BANK_CONFLICT", 0) / idx, 1),
                          "lds_active_pct_of_wave_cycles": round(100 * y.get("SQ_ACTIVE_INST_LDS", 0) / wc, 2), "wait_inst_lds_pct_of_wave_cycles": round(100 * z.get("SQ_WAIT_INST_LDS", 0) / wc3, 2),
                          "valu_active_pct_of_wave_cycles": round(100 * y.get("SQ_ACTIVE_INST_VALU", 0) / wc, 2), "lds_insts_per_wave": round(x.get("SQ_INSTS_LDS", 0) / (x.get("SQ_WAVES", 0) or 1), 1),
                          "valu_insts_per_wave": round(x.get("SQ_INSTS_VALU", 0) / (x.get("SQ_WAVES", 0) or 1), 1)}}
        json.dump(doc, open(os.path.join(os.path.dirname(out), "lds_%s.json" % wl), "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
