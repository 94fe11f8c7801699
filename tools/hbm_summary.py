#!/usr/bin/env python
"""Summarise the two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of tools/profile_round.sh
into profiles/<tag>_hbm_traffic.csv and profiles/hbm_traffic.json (the per-launch figures bench.py quotes in roofline.traffic).
    python tools/hbm_summary.py gpurun_out/r02a profiles/r02_a "<command note>" [workload]
Counter units: KB per dispatch as reported by rocprofv3.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of
wide coalesced reads (the x2 column); WRITE_SIZE and narrow access widths are uncalibrated, so both columns are given."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys


def csrc_sha16(root=None):
    """sha256 over the engine sources (gencore_amd/csrc/*.hip|hpp|cpp, sorted by name): bench.py quotes a traffic file only when its hash is the tree's."""
    root = root or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gencore_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(f for f in os.listdir(root) if f.endswith((".hip", ".hpp", ".cpp"))):
        h.update(fn.encode()); h.update(open(os.path.join(root, fn), "rb").read())
    return h.hexdigest()[:16]


def per_kernel(d):
    acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void "):
                k = k[5:]
            if not k.startswith("k_"):
                continue
            acc[k] += float(r["Counter_Value"]) * 1.024; cnt[k] += 1
    return {k: (acc[k] / cnt[k], cnt[k]) for k in acc}


def main():
    src, dst, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    workload = sys.argv[4] if len(sys.argv) > 4 else "cfg3"
    fe, wr = per_kernel(src + "_pmc_FETCH_SIZE"), per_kernel(src + "_pmc_WRITE_SIZE")
    rows = sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, (0, 0))[0] + wr.get(k, (0, 0))[0]))
    with open(dst + "_hbm_traffic.csv", "w") as f:
        f.write("# HBM traffic per dispatch, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only)\n")
        f.write("# %s\n" % note)
        f.write("# fetch_x2 = gfx950 correction (MI355X_MICROARCH.md), calibrated on this engine's access patterns: profiles/r03_fetch_calibration.json\n")
        f.write("kernel,dispatches,fetch_MB,fetch_MB_x2,write_MB\n")
        for k in rows:
            f.write("%s,%d,%.1f,%.1f,%.1f\n" % (k, fe.get(k, (0, 0))[1] or wr.get(k, (0, 0))[1], fe.get(k, (0, 0))[0] / 1e3, 2 * fe.get(k, (0, 0))[0] / 1e3, wr.get(k, (0, 0))[0] / 1e3))
    js = {k: {"fetch_bytes": fe.get(k, (0, 0))[0] * 1e3, "fetch_bytes_x2": 2e3 * fe.get(k, (0, 0))[0], "write_bytes": wr.get(k, (0, 0))[0] * 1e3} for k in rows}
    cons = [k for k in rows if k in ("k_vote", "k_score2", "k_consensus_fast", "k_consensus_slow", "k_deep_prepare", "k_vote_deep")]
    # (tools/mb/fetch_calib.hip, profiles/r03_fetch_calibration.json: the x2 of FETCH_SIZE holds for streaming reads of every lane width incl. the vote's
    #  unaligned 8-byte loads; WRITE_SIZE is exact; the counter unit is 1024 bytes)
    doc = {"note": note, "tag": dst.rsplit("/", 1)[-1], "workload": workload, "csrc_sha16": csrc_sha16(), "consensus_kernels": cons, "kernels": js}
    json.dump(doc, open(dst.rsplit("/", 1)[0] + "/hbm_traffic_%s.json" % workload, "w"), indent=1)
    if workload == "cfg3":
        json.dump(doc, open(dst.rsplit("/", 1)[0] + "/hbm_traffic.json", "w"), indent=1)
    print(open(dst + "_hbm_traffic.csv").read())


if __name__ == "__main__":
    main()
