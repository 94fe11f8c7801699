#!/usr/bin/env python
"""End-to-end measurement of the file path: a synthetic sorted BAM + FASTA on local disk -> gce_run_bam -> BAM.
    python tools/bam_bench.py --workload cfg3 --pairs 4000000 [--threads 0] [--level 1]
Prints one JSON line: the wall time of every stage and the PCIe-inclusive rates (DESIGN.md quotes them; they are never bench.py's
`value`, which starts with the inputs resident in HBM)."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from gencore_amd import capi, synth  # noqa: E402
from gencore_amd.bamio import run_bam, write_batch_as_bam  # noqa: E402
from gencore_amd.shard import effective_cpus  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--pairs", type=int, default=4_000_000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--level", type=int, default=1, help="deflate level of the OUTPUT file (htslib's default is 6)")
    ap.add_argument("--chunk", type=int, default=1 << 21)
    ap.add_argument("--dir", default=None)
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    d = synth.generate(args.workload, n_pairs=args.pairs, device=dev)
    batch = d.to_batch()
    tl = np.asarray(d.target_len, np.uint32)
    names = ["chr%d" % (i + 1) for i in range(len(tl))]
    tmp = args.dir or tempfile.mkdtemp(prefix="gce_bam_")
    src, out, fa = os.path.join(tmp, "in.bam"), os.path.join(tmp, "out.bam"), os.path.join(tmp, "ref.fa")
    t0 = time.time()
    write_batch_as_bam(src, batch, tl, names, threads=args.threads, level=1)
    t_make = time.time() - t0
    code = np.frombuffer(b"NATCG" + b"N" * 11, np.uint8)
    with open(fa, "wb") as f:
        for nm, (nib, ln) in zip(names, d.reference_host()):
            if nib is None:
                continue
            both = np.empty(len(nib) * 2, np.uint8)
            both[0::2] = nib & 0xF; both[1::2] = nib >> 4
            bases = code[both[:ln]]
            f.write(b">" + nm.encode() + b"\n")
            pad = (-ln) % 60
            lines = np.concatenate([bases, np.zeros(pad, np.uint8)]).reshape(-1, 60)
            body = np.concatenate([lines, np.full((len(lines), 1), 10, np.uint8)], 1).reshape(-1)
            f.write(body.tobytes().replace(b"\0", b""))
    prm = capi.default_params(umi_prefix="auto", cluster_size_req=d.info["supporting_reads"])
    runs = []
    for rep in range(2):                                        # second run: page cache warm, allocations done
        t0 = time.time()
        r = run_bam(src, out, prm, fasta=(fa if rep == 0 else None), threads=args.threads, chunk_reads=args.chunk, level=args.level)
        runs.append((time.time() - t0, r))
    wall, r = runs[-1]
    n_pairs = d.info["n_pairs"]
    in_bytes, out_bytes = os.path.getsize(src), os.path.getsize(out)
    unc = int(batch.seq.size + batch.qual.size + batch.qname.size + 4 * batch.cigar.size + 40 * batch.n)
    res = dict(workload=args.workload, pairs=int(n_pairs), reads=int(batch.n), host_threads=(args.threads or min(effective_cpus(), 64)), visible_cpus=os.cpu_count(),
               in_bam_bytes=in_bytes, out_bam_bytes=out_bytes, uncompressed_bytes=unc, records_out=int(r.n_out),
               stage_s=dict(open=round(r.open_s, 4), open_read=round(r.read_s, 4), open_inflate=round(r.inflate_s, 4), open_index=round(r.index_s, 4), soa_fill_and_submit=round(r.submit_s, 4), process=round(r.process_s, 4), drain=round(r.drain_s, 4),
                            write=round(r.write_s, 4), total=round(r.total_s, 4)),
               kernel_ms=round(r.kernel_ms, 3),
               pairs_per_s=dict(end_to_end=round(n_pairs / r.total_s), without_file_io=round(n_pairs / (r.submit_s + r.process_s + r.drain_s)),
                                host_to_result=round(n_pairs / (r.submit_s + r.process_s + r.drain_s)), kernels_only=round(n_pairs / (r.kernel_ms * 1e-3))),
               first_run_total_s=round(runs[0][1].total_s, 3), make_input_s=round(t_make, 2), output_level=args.level)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
