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

from gencore_amd import capi  # noqa: E402  (synth -- and torch with it -- only in the generating parent)
from gencore_amd.bamio import run_bam, run_bam_sharded, write_batch_as_bam  # noqa: E402
from gencore_amd.shard import effective_cpus  # noqa: E402


def child(args):
    tmp = args.child
    src, out, fa = os.path.join(tmp, "in.bam"), os.path.join(tmp, "out.bam"), os.path.join(tmp, "ref.fa")
    prm = capi.default_params(umi_prefix="auto", cluster_size_req=int(args.sreq))
    runs = []
    for rep in range(2):                                        # second run: page cache warm, allocations done
        t0 = time.time()
        r = run_bam(src, out, prm, fasta=(fa if rep == 0 and not os.environ.get("GCE_BENCH_NO_FASTA") else None), threads=args.threads, chunk_reads=args.chunk, level=args.level)
        runs.append((time.time() - t0, r))
    wall, r = runs[-1]
    n_pairs, t_make = int(args.npairs), float(args.make_s)
    in_bytes, out_bytes = os.path.getsize(src), os.path.getsize(out)
    unc = int(args.unc)

    class B:
        n = int(args.nreads)
    batch = B()
    res = dict(workload=args.workload, pairs=int(n_pairs), reads=int(batch.n), host_threads=(args.threads or min(effective_cpus(), 64)), visible_cpus=os.cpu_count(),
               in_bam_bytes=in_bytes, out_bam_bytes=out_bytes, uncompressed_bytes=unc, records_out=int(r.n_out),
               path=("host codec (GCE_BAM_HOSTCODEC)" if os.environ.get("GCE_BAM_HOSTCODEC") else "streaming, GPU-assisted (windows: read | inflate | copy to HBM overlap; records indexed, parsed and re-assembled in HBM)"),
               stage_s=dict(input_pipeline=round(r.open_s, 4), reader_thread_busy=round(r.read_s, 4), inflate_all_threads=round(r.inflate_s, 4), waits_for_copies=round(r.submit_s, 4), gpu_index_and_parse=round(r.index_s, 4),
                            process=round(r.process_s, 4), gpu_output_records=round(r.drain_s, 4), write=round(r.write_s, 4), total=round(r.total_s, 4)),
               peak_rss_mb=round(r.peak_rss_kb / 1024.0, 1), rss_at_entry_mb=round(r.rss_start_kb / 1024.0, 1), first_run_rss_at_entry_mb=round(runs[0][1].rss_start_kb / 1024.0, 1), first_run_peak_rss_mb=round(runs[0][1].peak_rss_kb / 1024.0, 1),
               kernel_ms=round(r.kernel_ms, 3),
               pairs_per_s=dict(end_to_end=round(n_pairs / r.total_s), kernels_only=round(n_pairs / (r.kernel_ms * 1e-3))),
               first_run_total_s=round(runs[0][1].total_s, 3), make_input_s=round(t_make, 2), output_level=args.level)
    if True:                                                    # the same run with the output deflated by the GPU (level -2) and by the host's fixed-Huffman encoder (level -1)
        res["output_encoders"] = {}
        for tag, lv in (("gpu_fixed_huffman_level_-2", -2), ("host_fixed_huffman_level_-1", -1)):
            o2 = os.path.join(tmp, "out_lv%d.bam" % lv)
            rg = None
            for rep in range(3):                               # best of three, the input re-read into the page cache in front of each (the FASTA and the outputs of the runs before push it out on a small box)
                with open(src, "rb") as fh_:
                    while fh_.read(1 << 26):
                        pass
                r_ = run_bam(src, o2, prm, fasta=None, threads=args.threads, chunk_reads=args.chunk, level=lv)
                if rg is None or r_.total_s < rg.total_s:
                    rg = r_
            res["output_encoders"][tag] = dict(total_s=round(rg.total_s, 4), write_s=round(rg.write_s, 4), out_bam_bytes=os.path.getsize(o2), records_out=int(rg.n_out))
    if args.shards and int(args.shards) > 1:                   # the same file through gce_run_bam_sharded: K engines on device 0 (one GPU here: the paths, not a scaling claim)
        K = int(args.shards)
        res["sharded"] = {"engines": K, "devices": [0] * K, "note": "all engines on one GPU: every engine inflates and indexes the whole stream itself (on K GPUs that happens side by side)"}
        for tag, env in (("gpu_codec", None), ("hostcodec_round2", "1")):
            if env:
                os.environ["GCE_BAM_HOSTCODEC"] = env
            try:
                rs, reps, thr0 = None, [], _throttle()
                for rep in range(3):
                    o2 = os.path.join(tmp, "out_sh_%s.bam" % tag)
                    with open(src, "rb") as fh_:
                        while fh_.read(1 << 26):
                            pass
                    r_ = run_bam_sharded(src, o2, prm, [0] * K, fasta=None, threads=args.threads, level=args.level)      # (no FASTA, like the timed single-engine run it is compared with)
                    reps.append(round(r_.total_s, 4))
                    if rs is None or r_.total_s < rs.total_s:
                        rs = r_
                res["sharded"][tag] = dict(all_reps_s=reps, cgroup_throttled=_throttle_delta(thr0), total_s=round(rs.total_s, 4), input_pipeline=round(rs.open_s, 4), index_plan_select=round(rs.index_s, 4), process=round(rs.process_s, 4), merge=round(rs.drain_s, 4), write=round(rs.write_s, 4),
                                           kernel_ms_slowest_engine=round(rs.kernel_ms, 3), records_out=int(rs.n_out), output_identical_to_single_engine=open(o2, "rb").read() == open(out, "rb").read(),
                                           stats_equal=bool(bytes(rs.pre) == bytes(r.pre) and bytes(rs.post) == bytes(r.post)))
            finally:
                os.environ.pop("GCE_BAM_HOSTCODEC", None)
    print(json.dumps(res))


def _throttle():
    """cgroup v2 CPU bandwidth control: (periods throttled, seconds throttled) of this container so far"""
    try:
        kv = dict(ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0)) / 1e6
    except Exception:
        return None


def _throttle_delta(a):
    b = _throttle()
    return None if a is None or b is None else {"periods": b[0] - a[0], "seconds": round(b[1] - a[1], 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--pairs", type=int, default=4_000_000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--level", type=int, default=1, help="deflate level of the OUTPUT file (htslib's default is 6)")
    ap.add_argument("--chunk", type=int, default=1 << 21)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--sam", action="store_true", help="with --c-caller: also time SAM text in / out")
    ap.add_argument("--c-caller", action="store_true", help="also run the files through tools/run_bam.c (a plain C process) and report its times and peak RSS")
    ap.add_argument("--shards", default=None, help="also run the file through gce_run_bam_sharded with this many engines on device 0 (GPU codec and round 2's host codec)")
    for k_ in ("--child", "--make-s", "--npairs", "--nreads", "--sreq", "--unc"):
        ap.add_argument(k_, default=None)
    args = ap.parse_args()
    import torch
    from gencore_amd import synth
    dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    if args.child is not None:
        return child(args)
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
    if args.child is None:                                      # the runs happen in a fresh process: its peak RSS is the file path's, not the generator's
        import subprocess
        outp = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tmp, "--workload", args.workload, "--pairs", str(args.pairs), "--threads", str(args.threads),
                               "--level", str(args.level), "--chunk", str(args.chunk)] + (["--shards", str(args.shards)] if args.shards else []) + ["--make-s", "%.2f" % t_make, "--npairs", str(d.info["n_pairs"]), "--nreads", str(batch.n),
                               "--sreq", str(d.info["supporting_reads"]), "--unc", str(int(batch.seq.size + batch.qual.size + batch.qname.size + 4 * batch.cigar.size + 40 * batch.n))], stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in outp.stdout.splitlines() if ln.startswith("{")]
        if lines and args.c_caller:                             # the same files through a plain C process: the path's own resident memory
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            exe = os.path.join(tmp, "run_bam")
            subprocess.run(["gcc", "-std=c11", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "tools", "run_bam.c"), "-L" + os.path.join(root, "gencore_amd", "csrc"),
                            "-lgencore_amd", "-Wl,-rpath," + os.path.join(root, "gencore_amd", "csrc"), "-o", exe], check=True)
            c = subprocess.run([exe, src, os.path.join(tmp, "out_c.bam"), "-" if os.environ.get("GCE_BENCH_NO_FASTA") else fa, str(args.threads), str(args.level), str(d.info["supporting_reads"]), "2"],
                               stdout=subprocess.PIPE, text=True)
            cl = [ln for ln in c.stdout.splitlines() if ln.startswith("{")]
            res = json.loads(lines[-1]); res["c_caller"] = json.loads(cl[-1]) if cl else {"error": c.stdout[-300:]}
            same = cl and open(out, "rb").read() == open(os.path.join(tmp, "out_c.bam"), "rb").read()
            res["c_caller"]["output_identical_to_python_run"] = bool(same)
            c2 = subprocess.run([exe, src, os.path.join(tmp, "out_c2.bam"), "-", str(args.threads), str(args.level), str(d.info["supporting_reads"]), "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                env=dict(os.environ, GCE_RAW_TIMING="1"))
            cl2 = [ln for ln in c2.stdout.splitlines() if ln.startswith("{")]
            res["c_caller_fresh_process_no_fasta"] = json.loads(cl2[-1]) if cl2 else {"error": c2.stdout[-300:]}      # one run, cold allocations, no reference on the host: the path's own footprint
            res["c_caller_fresh_process_no_fasta"]["rss_trace"] = [ln.strip() for ln in c2.stderr.splitlines() if "RSS" in ln]
            if args.sam:                                        # the same stream as SAM text: in and out (and SAM in -> BAM out)
                from gencore_amd.bamio import bam_to_sam
                sam_in = os.path.join(tmp, "in.sam")
                t1 = time.time(); bam_to_sam(src, sam_in, threads=args.threads); conv = time.time() - t1
                res["sam"] = {"in_sam_bytes": os.path.getsize(sam_in), "bam_to_sam_s": round(conv, 3)}
                for tag, outp2 in (("sam_to_sam", os.path.join(tmp, "out_c.sam")), ("sam_to_bam", os.path.join(tmp, "out_c3.bam"))):
                    c3 = subprocess.run([exe, sam_in, outp2, "-", str(args.threads), str(args.level), str(d.info["supporting_reads"]), "2"], stdout=subprocess.PIPE, text=True)
                    cl3 = [ln for ln in c3.stdout.splitlines() if ln.startswith("{")]
                    res["sam"][tag] = json.loads(cl3[-1]) if cl3 else {"error": c3.stdout[-300:]}
                res["sam"]["out_sam_bytes"] = os.path.getsize(os.path.join(tmp, "out_c.sam")) if os.path.exists(os.path.join(tmp, "out_c.sam")) else None
            lines[-1] = json.dumps(res)
        sys.stdout.write((lines[-1] + "\n") if lines else outp.stdout)
        return


if __name__ == "__main__":
    main()
