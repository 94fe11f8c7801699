"""N>1 path on CPU: two gloo ranks, each owns a key-range shard cut inside a contig (gencore_amd/shard.py: global ticks + the
stream's flush events), no data-path collective, one all-reduce(sum) of the additive Stats blocks — the same plumbing
bench.py uses with RCCL.  The per-shard compute stand-in is the oracle (tests may use it); the product's kernels are
covered by the -m gpu tests."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, seed, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fuzzgen
    from gencore_amd.shard import plan_shards, shard_by_plan, stream_context
    from oracle import oracle_py
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=60, umi_mode="prefix", period=13)
    tick, ev_tid, ev_pos = stream_context(batch.core, over["flush_period"])
    sub, idx = shard_by_plan(batch, plan_shards(batch.core, world, "range"), rank, tick)
    r = oracle_py.run(sub, fuzzgen.make_params(over, contig_len), reference, events=(ev_tid, ev_pos))
    # the payload of the final Stats merge (gce_stats_payload_device lays it out the same way): both Stats blocks, the per-contig depth bins
    # and the BED region counts of this rank's reads and records, ONE buffer, ONE all-reduce(sum)
    step, regions = 50, [(0, 10 * k, 10 * k + 35) for k in range(0, int(contig_len[0]) // 10, 7)]
    off, pre_d, post_d, pre_b, post_b = oracle_py.depth_stats(sub, r, contig_len, step, regions)
    stats = torch.from_numpy(np.concatenate([r.pre.as_array(), r.post.as_array(), pre_d, post_d, pre_b, post_b]))
    dist.barrier()
    dist.all_reduce(stats)
    n_out = torch.tensor([int((r.out_flag != 0).sum())])
    dist.all_reduce(n_out)
    if rank == 0:
        whole = oracle_py.run(batch, fuzzgen.make_params(over, contig_len), reference)
        _, w_pre_d, w_post_d, w_pre_b, w_post_b = oracle_py.depth_stats(batch, whole, contig_len, step, regions)
        want = np.concatenate([whole.pre.as_array(), whole.post.as_array(), w_pre_d, w_post_d, w_pre_b, w_post_b])
        assert w_pre_d.sum() > 0 and w_pre_b.sum() > 0
        q.put((bool(np.array_equal(stats.numpy(), want)), int(n_out.item()), int((whole.out_flag != 0).sum())))
    dist.destroy_process_group()


def test_two_rank_shards_and_stats_allreduce(built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 500, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, n_out, want_out = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok and n_out == want_out
