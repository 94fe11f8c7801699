#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration for this engine's access patterns: tools/fetch_calib.sh  (GPU box; binary built on the CPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/calib_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/calib_$c -o p -- tools/mb/fetch_calib > gpurun_out/calib_$c.log 2>&1
done
python - <<'P'
import csv, glob, json
known = {"k_c16": 2 << 30, "k_coalesced<unsigned long>": 2 << 30, "k_coalesced<unsigned int>": 2 << 30, "k_coalesced<unsigned short>": 2 << 30, "k_coalesced<unsigned char>": 1 << 30,
         "k_vote_a": (2 << 30) // 150 * 225, "k_desc32": (2 << 30) // 256 * 32, "k_w16": 2 << 30, "k_w4": 2 << 30, "k_w16s": (2 << 30) // 256 * 16}
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob("gpurun_out/calib_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k in known:
                out.setdefault(k, {"useful_bytes": known[k]})[c + "_bytes"] = float(r["Counter_Value"]) * 1e3
for k, v in out.items():
    if v.get("FETCH_SIZE_bytes"):
        v["useful_over_FETCH_SIZE"] = round(v["useful_bytes"] / v["FETCH_SIZE_bytes"], 3)
    if v.get("WRITE_SIZE_bytes"):
        v["useful_over_WRITE_SIZE"] = round(v["useful_bytes"] / v["WRITE_SIZE_bytes"], 3)
json.dump({"what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (counter unit KB) of tools/mb/fetch_calib.hip: buffers of known size read / written once, 2 GiB each (beyond the 256 MB Infinity Cache)", "kernels": out},
          open("gpurun_out/r03_fetch_calibration.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
