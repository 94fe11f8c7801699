#!/usr/bin/env python
"""bench.py — consensus read-pairs/s of the HIP engine on BASELINE.json's workload, with the dominant kernel's
HBM roofline and the CPU baseline (oracle port, 1 thread) in the same JSON line.

    python bench.py --gpus 1 --steps 5 --warmup 2                       # default workload: cfg3 (10 M pairs, UMI, depth 8, -s 2)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one gce_process() pass of the whole hot path (clustering scan -> pairing/UMI grouping -> scoring ->
template pick + column vote -> duplex/filter/tags -> Stats) over one synthetic coordinate-sorted stream that is
already resident in HBM (gce_submit_device).  Every step (warm-up included) gets its own pristine copy of the
mutable seq/qual blobs, so no work is skipped or cached.  Multi-GPU: one rank per GPU, each rank owns its own
coordinate shard (weak scaling, no data-path collective); the per-step Stats merge is one RCCL all-reduce.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch  # first: the engine must share torch's HIP runtime

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable copy rate


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", help="cfg2 | cfg3 | cfg4s | cfg5 (see gencore_amd/synth.py)")
    ap.add_argument("--pairs", type=int, default=None, help="override the workload's pair count (per GPU)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=2_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--align", type=int, default=1, help="byte alignment of each read's seq/qual slice in the SoA blobs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist:
        dist.barrier()
    import numpy as np
    from gencore_amd import capi, synth
    from gencore_amd.capi import GceBatch, GceResult, GceStats, GceTiming
    lib = capi.load_library()
    dev = torch.device("cuda", local_rank)

    # ------------------------------------------------------------------ workload (synthetic, generated on the GPU)
    data = synth.generate(args.workload, n_pairs=args.pairs, seed=rank, device=dev, align=args.align)
    t = data.t
    n_reads, n_pairs = data.n_reads, data.info["n_pairs"]
    tl = np.asarray(data.target_len, np.uint32)
    prm = capi.default_params(device=local_rank, n_targets=len(tl), target_len=tl.ctypes.data, umi_prefix=data.info["umi_prefix"],
                              cluster_size_req=data.info["supporting_reads"])
    eng = C.c_void_p()
    rc = lib.gce_create(C.byref(prm), C.byref(eng))
    if rc:
        raise SystemExit("gce_create failed: %s" % lib.gce_status_message(rc).decode())
    for tid, (nib, ln) in enumerate(data.reference):
        assert lib.gce_set_reference(eng, tid, nib.data_ptr(), ln) == 0

    n_copies = args.steps + args.warmup

    def padded_clone(x):                                    # device blobs must be readable 16 bytes past their end
        y = torch.zeros(x.numel() + 64, dtype=x.dtype, device=x.device)
        y[:x.numel()].copy_(x)
        return y
    seqs = [padded_clone(t["seq"]) for _ in range(n_copies)]
    quals = [padded_clone(t["qual"]) for _ in range(n_copies)]
    t["qname"] = padded_clone(t["qname"])

    def make_batch(k):
        b = GceBatch()
        b.n_reads = n_reads
        b.core, b.qname_off, b.qname = t["core"].data_ptr(), t["qname_off"].data_ptr(), t["qname"].data_ptr()
        b.cigar_off, b.cigar = t["cigar_off"].data_ptr(), t["cigar"].data_ptr()
        b.seq_off, b.seq, b.qual_off, b.qual = t["seq_off"].data_ptr(), seqs[k].data_ptr(), t["qual_off"].data_ptr(), quals[k].data_ptr()
        b.nm, b.nm_type = t["nm"].data_ptr(), t["nm_type"].data_ptr()
        b.mi_off, b.mi = None, None
        b.qname_bytes, b.cigar_words = t["qname"].numel(), t["cigar"].numel()
        b.seq_bytes, b.qual_bytes, b.mi_bytes = t["seq"].numel(), t["qual"].numel(), 0
        return b

    stats_dev = torch.zeros(2 * capi.GCE_STATS_WORDS, dtype=torch.int64, device=dev)
    timings, last_res = [], {}

    def step(k):
        b = make_batch(k)
        rc = lib.gce_submit_device(eng, C.byref(b))
        rc = rc or lib.gce_process(eng)
        if rc:
            raise SystemExit("engine failed: %s" % lib.gce_last_error(eng).decode())
        r = GceResult()
        lib.gce_result_device(eng, C.byref(r))
        if dist:                                            # the final Stats merge: one RCCL all-reduce over xGMI
            host = np.concatenate([r.pre.as_array(), r.post.as_array()])
            stats_dev.copy_(torch.from_numpy(host))
            dist.all_reduce(stats_dev)
        tm = GceTiming()
        lib.gce_get_timing(eng, C.byref(tm))
        timings.append(tm.as_dict())
        last_res["n_out"], last_res["pre"], last_res["post"] = int(r.n_out), r.pre.as_dict(), r.post.as_dict()

    def sync():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    sync()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    ms_per_step = elapsed * 1000.0 / args.steps
    total_pairs = torch.tensor([n_pairs], dtype=torch.int64, device=dev)
    if dist:
        dist.all_reduce(total_pairs)
    pairs_all = int(total_pairs.item())
    value = pairs_all / (ms_per_step / 1000.0)

    # ------------------------------------------------------------------ roofline of the dominant kernel (HIP events, timed steps only)
    timed = timings[args.warmup:]
    avg = {k: sum(x[k] for x in timed) / len(timed) for k in timed[0]}
    n_groups = max(1.0, avg["n_groups"])
    d = n_pairs / n_groups                                  # mean group depth
    L = data.info["read_len"]
    per_read = (L + 1) // 2 + L + 4                         # seq + qual + one CIGAR word (229 B at 150 bp)
    consensus_bytes = n_pairs * (2 * per_read + (2 * ((L + 1) // 2 + L) + L) / d)    # 458 + 600/d per pair at 150 bp
    cluster_bytes = n_reads * 40.0                          # 32 B key record in + 8 B (slot, rank) out
    kernels = {
        "cluster": dict(ms=avg["cluster_ms"], algorithmic_bytes=cluster_bytes),
        "consensus": dict(ms=avg["score_ms"] + avg["consensus_ms"], algorithmic_bytes=consensus_bytes),
    }
    for v in kernels.values():
        v["achieved_gbs"] = v["algorithmic_bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0
        v["frac"] = v["achieved_gbs"] / HBM_PEAK_GBS
    phase_ms = {k: avg[k] for k in ("prescan_ms", "cluster_ms", "csr_ms", "pairing_ms", "score_ms", "consensus_ms", "finish_ms", "total_ms")}
    dom = "consensus" if kernels["consensus"]["ms"] >= kernels["cluster"]["ms"] else "cluster"
    # HBM traffic of the dominant phase from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this
    # same command: profiles/hbm_traffic.json, tools/hbm_summary.py): bytes per launch, FETCH_SIZE doubled as MI355X_MICROARCH.md
    # prescribes for gfx950.  Quoted only for the workload it was measured on; PMC counters cannot be read inside this process.
    traffic = None
    tj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tj) and args.workload == "cfg3" and args.pairs is None:
        kk = json.load(open(tj))["kernels"]
        names = {"consensus": ("k_score", "k_score2", "k_consensus_lean2", "k_consensus_lean", "k_consensus_fast", "k_consensus_slow"), "cluster": ("k_cluster",)}
        def tr(ns):
            return sum(kk[n]["fetch_bytes_x2"] + kk[n]["write_bytes"] for n in ns if n in kk)
        traffic = {k: tr(v) for k, v in names.items()}
    roofline = dict(bound="hbm", kernel={"consensus": "k_score2+k_consensus_* (Pair::computeScore + Group::makeConsensus)",
                                         "cluster": "k_cluster (clustering scan)"}[dom],
                    achieved=round(kernels[dom]["achieved_gbs"], 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(kernels[dom]["frac"], 5), traffic=(round(traffic[dom]) if traffic else None), algorithmic_bytes=round(kernels[dom]["algorithmic_bytes"]),
                    clustering_scan=dict(achieved=round(kernels["cluster"]["achieved_gbs"], 2), frac=round(kernels["cluster"]["frac"], 5),
                                         ms=round(kernels["cluster"]["ms"], 4), algorithmic_bytes=round(cluster_bytes),
                                         traffic=(round(traffic["cluster"]) if traffic else None)),
                    phase_ms={k: round(v, 4) for k, v in phase_ms.items()}, mean_group_depth=round(d, 3))

    # ------------------------------------------------------------------ CPU baseline: the oracle port, 1 thread, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_py
        sample_pairs = min(args.cpu_sample_pairs, n_pairs)
        sd = synth.generate(args.workload, n_pairs=sample_pairs, seed=12345, device=dev)
        sb = sd.to_batch()
        stl = np.asarray(sd.target_len, np.uint32)
        sprm = capi.default_params(n_targets=len(stl), target_len=stl.ctypes.data, umi_prefix=sd.info["umi_prefix"],
                                   cluster_size_req=sd.info["supporting_reads"])
        ref_host = sd.reference_host()
        oracle_py.lib()
        c0 = time.perf_counter()
        res = oracle_py.run(sb, sprm, ref_host)
        cs = time.perf_counter() - c0
        assert res.status == 0
        cpu = dict(value=round(sd.info["n_pairs"] / cs, 1), unit="read-pairs/s", cores=1, kind="port",
                   sample="%s generator, %d pairs, oracle/gencore_oracle.c single thread, %.1f s" % (args.workload, sd.info["n_pairs"], cs))

    if rank == 0:
        out = {
            "metric": "consensus read-pairs/sec (whole node) + clustering HBM GB/s vs roofline",
            "value": round(value, 1), "unit": "read-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %d paired %d bp reads per GPU, %s, mean group depth %.1f, -s %d, coordinate-sharded x%d" % (
                args.workload, n_pairs, L, ("%d bp UMI" % data.info["umi_len"]) if data.info["umi_len"] else "no UMI", d,
                data.info["supporting_reads"], world), "pairs_per_gpu": n_pairs, "reads_per_gpu": n_reads,
                "records_out_per_gpu": last_res.get("n_out"), "parallelism": "coordinate shards x%d, Stats all-reduce" % world},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if cpu:
            out["speedup_vs_cpu_port"] = round(value / cpu["value"], 2)
        print(json.dumps(out))
    lib.gce_destroy(eng)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
