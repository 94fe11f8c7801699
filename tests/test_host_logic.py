"""Host-side logic on CPU: SoA batch round trips, generator determinism, stream sharding context (tick_offset /
trailing_flush) checked with the oracle, oracle vs frozen golden vectors."""
import json
import os

import numpy as np
import pytest

import fuzzgen
from gencore_amd.batch import ReadBatch
from parity_helpers import diff_results

HERE = os.path.dirname(os.path.abspath(__file__))


def test_batch_roundtrip():
    recs = [dict(qname="r1:UMI_ACGT", flag=99, tid=0, pos=10, cigar="3S10M2I5M", mtid=0, mpos=40, isize=60, seq="ACGTNACGTAACGTACGTAC", qual=list(range(20)), nm=3),
            dict(qname="r2", flag=147, tid=0, pos=40, cigar="*", mtid=0, mpos=10, isize=-60, seq="ACG", qual="III", nm=None, mi="x:ACGT")]
    b = ReadBatch.from_records(recs)
    assert b.n == 2 and b.core.dtype.itemsize == 32
    assert b.qname_of(0) == "r1:UMI_ACGT" and b.cigar_of(0) == "3S10M2I5M" and b.cigar_of(1) == "*"
    assert b.seq_of(0) == "ACGTNACGTAACGTACGTAC" and b.qual_of(1).tolist() == [40, 40, 40]
    assert b.core["l_qname"].tolist() == [12, 3] and b.nm_type.tolist() == [ord("C"), 0]
    assert b.mi is not None and b.mi_off[0] == 0xFFFFFFFFFFFFFFFF and b.mi_off[1] == 0


def test_synth_is_deterministic_and_sorted():
    from gencore_amd import synth
    a = synth.generate("cfg1s").to_batch()
    b = synth.generate("cfg1s").to_batch()
    for f in ReadBatch.FIELDS:
        x, y = getattr(a, f), getattr(b, f)
        assert (x is None and y is None) or np.array_equal(x, y), f
    key = a.core["tid"].astype(np.int64) << 32 | a.core["pos"].astype(np.int64)
    assert np.all(np.diff(key) >= 0)
    # mates agree on the cluster key (gencore.cpp:300-304)
    left = np.where(a.core["isize"] < 0, a.core["mpos"], a.core["pos"])
    names = [a.qname_of(i) for i in range(200)]
    seen = {}
    for i, nme in enumerate(names):
        if nme in seen:
            assert left[i] == left[seen[nme]]
        seen[nme] = i


@pytest.mark.parametrize("seed,world", [(400, 2), (401, 3), (402, 2)])
def test_sharded_stream_equals_whole_stream(oracle, seed, world):
    """The stream context a shard gets (tick_offset, trailing_flush) makes per-shard results identical to the
    whole-stream results — including which clusters get -d vs the end-of-file UMI threshold (quirk Q1)."""
    from gencore_amd.shard import shard_by_contig
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=70, umi_mode="prefix", period=[11, 29, 5][seed % 3])
    whole = oracle.run(batch, fuzzgen.make_params(over, contig_len), reference)
    assert whole.status == 0
    flags = np.zeros(batch.n, np.uint8)
    fr = np.full(batch.n, -1, np.int16)
    pre = np.zeros(114, np.int64)
    post = np.zeros(114, np.int64)
    covered = np.zeros(batch.n, bool)
    for rank in range(world):
        sub, idx, ctx = shard_by_contig(batch, world, rank, over["flush_period"])
        covered[idx] = True
        r = oracle.run(sub, fuzzgen.make_params(dict(over, **ctx), contig_len), reference)
        assert r.status == 0
        flags[idx], fr[idx] = r.out_flag, r.fr
        pre += r.pre.as_array(); post += r.post.as_array()
    assert covered.all()
    assert np.array_equal(flags, whole.out_flag) and np.array_equal(fr, whole.fr)
    assert np.array_equal(pre, whole.pre.as_array()) and np.array_equal(post, whole.post.as_array())


@pytest.mark.parametrize("seed,world,mode", [(400, 2, "range"), (401, 3, "range"), (402, 4, "lpt"), (403, 5, "range"), (407, 8, "range")])
def test_key_range_shards_equal_whole_stream(oracle, seed, world, mode):
    """Cuts by cluster key (tid, left) INSIDE a contig: a shard's reads interleave with its neighbours' in stream order, so every
    read carries its global tick and every shard replays the whole stream's flush events (gencore.cpp:319-354).  mode="lpt" is the
    depth^2-weighted deal of whole clusters used for ultra-deep hotspots."""
    from gencore_amd.shard import plan_shards, shard_by_plan, stream_context
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=70, umi_mode="prefix", period=[11, 29, 5][seed % 3])
    prm = fuzzgen.make_params(over, contig_len)
    whole = oracle.run(batch, prm, reference)
    tick, et, ep = stream_context(batch.core, over["flush_period"])
    plan = plan_shards(batch.core, world, mode)
    flags = np.zeros(batch.n, np.uint8); fr = np.full(batch.n, -1, np.int16)
    pre = np.zeros(114, np.int64); post = np.zeros(114, np.int64)
    cuts_inside_contig = False
    for r in range(world):
        sub, idx = shard_by_plan(batch, plan, r, tick)
        if r and len(idx) and batch.core["tid"][idx[0]] in batch.core["tid"][plan == r - 1]:
            cuts_inside_contig = True
        res = oracle.run(sub, prm, reference, events=(et, ep))
        assert res.status == 0
        flags[idx], fr[idx] = res.out_flag, res.fr
        pre += res.pre.as_array(); post += res.post.as_array()
    assert cuts_inside_contig
    assert np.array_equal(flags, whole.out_flag) and np.array_equal(fr, whole.fr)
    assert np.array_equal(pre, whole.pre.as_array()) and np.array_equal(post, whole.post.as_array())


@pytest.mark.parametrize("seed", range(420, 432))
def test_contiguous_cut_shards_equal_whole_stream(oracle, seed):
    """Contiguous slices cut where no cluster key is open (neither a mate nor the key's right end reaches over the cut) are exact
    shards with tick_offset / trailing_flush alone: the cheap way to spread one stream over processes or GPUs."""
    from gencore_amd.shard import contiguous_cuts, shard_contiguous
    world = 2 + seed % 7
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=70, umi_mode="prefix", period=[11, 29, 5][seed % 3])
    whole = oracle.run(batch, fuzzgen.make_params(over, contig_len), reference)
    bounds = contiguous_cuts(batch.core, world, contig_len)
    assert bounds[0] == 0 and bounds[-1] == batch.n and all(a <= b for a, b in zip(bounds, bounds[1:]))
    flags = np.zeros(batch.n, np.uint8); pre = np.zeros(114, np.int64); post = np.zeros(114, np.int64)
    for r in range(world):
        if bounds[r] == bounds[r + 1]:
            continue
        sub, idx, ctx = shard_contiguous(batch, bounds, r, over["flush_period"])
        res = oracle.run(sub, fuzzgen.make_params(dict(over, **ctx), contig_len), reference)
        assert res.status == 0
        flags[idx] = res.out_flag
        pre += res.pre.as_array(); post += res.post.as_array()
    assert np.array_equal(flags, whole.out_flag)
    assert np.array_equal(pre, whole.pre.as_array()) and np.array_equal(post, whole.post.as_array())


def test_sharded_generator_equals_whole_stream():
    """synth.generate(shard=(rank, world)): every rank plans the whole stream and materialises its own key range — the records, the
    global ticks and the flush events are those of the unsharded stream (what bench.py --gpus N runs)."""
    from gencore_amd import synth
    from gencore_amd.shard import stream_context
    for name, n_pairs, world in (("cfg4s", 20000, 3), ("cfg5", 12000, 2)):
        kw = dict(scale=0.01) if name == "cfg4s" else {}
        whole = synth.generate(name, n_pairs=n_pairs, **kw).to_batch()
        tick, et, ep = stream_context(whole.core, 10000)
        cover = np.zeros(whole.n, bool)
        for r in range(world):
            d = synth.generate(name, n_pairs=n_pairs, shard=(r, world), **kw)
            b, gi = d.to_batch(), d.global_index.numpy()
            cover[gi] = True
            assert np.array_equal(b.core, whole.core[gi]) and np.array_equal(b.nm, whole.nm[gi])
            assert np.array_equal(d.stream_context["tick"].numpy(), tick[gi].astype(np.int64))
            assert np.array_equal(d.stream_context["ev_tid"], et) and np.array_equal(d.stream_context["ev_pos"], ep)
            for i in range(0, b.n, max(1, b.n // 40)):
                g = int(gi[i])
                assert b.qname_of(i) == whole.qname_of(g) and b.seq_of(i) == whole.seq_of(g) and np.array_equal(b.qual_of(i), whole.qual_of(g))
        assert cover.all()


def test_oracle_matches_frozen_regression_vectors(oracle):
    """tests/golden/oracle_regression.json: digests of oracle outputs frozen by tests/golden/make_golden.py.
    These are REGRESSION vectors of the oracle itself (the reference ships no golden output for this path and cannot
    be built here); the reference's own known-answer vectors are in test_oracle_known_answers.py."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden
    frozen = json.load(open(os.path.join(HERE, "golden", "oracle_regression.json")))
    for case in frozen["cases"]:
        got = make_golden.digest_case(case["seed"], case["kwargs"])
        assert got == case["digest"], case


def rows_from_table(batch, t):
    """The engine's table of emitted records (gce_result) rebuilt from a per-read ResultTable: rows in bamComp order
    (gencore.h:19-47, input index as the last key), compact 16-byte aligned blobs."""
    em = np.nonzero(t.out_flag)[0]
    c = batch.core[em]
    order = np.lexsort((em, c["isize"], c["mpos"], c["mtid"], c["pos"], c["tid"]))
    src = em[order].astype(np.uint32)
    row_of = np.full(batch.n, 0xFFFFFFFF, np.uint32); row_of[src] = np.arange(len(src), dtype=np.uint32)
    lq = batch.core["l_qseq"].astype(np.int64)[src]
    su, qu = ((lq + 1) // 2 + 15) // 16 * 16, (lq + 15) // 16 * 16
    seq_off, qual_off = np.cumsum(su) - su, np.cumsum(qu) - qu
    seq, qual = np.zeros(int(su.sum()), np.uint8), np.zeros(int(qu.sum()), np.uint8)
    for k, i in enumerate(src):
        i = int(i); L = int(lq[k])
        seq[seq_off[k]:seq_off[k] + (L + 1) // 2] = t.seq[int(batch.seq_off[i]):int(batch.seq_off[i]) + (L + 1) // 2]
        qual[qual_off[k]:qual_off[k] + L] = t.qual[int(batch.qual_off[i]):int(batch.qual_off[i]) + L]
    m = t.mate[src]
    mate = np.where(m == 0xFFFFFFFF, np.uint32(0xFFFFFFFF), row_of[np.where(m == 0xFFFFFFFF, 0, m).astype(np.int64)])
    return dict(src=src, kind=t.out_flag[src], qname_src=t.qname_src[src], nm_new=t.nm_new[src], fr=t.fr[src], rr=t.rr[src],
                mate=mate.astype(np.uint32), seq_off=seq_off.astype(np.uint64), qual_off=qual_off.astype(np.uint64), seq=seq, qual=qual)


@pytest.mark.parametrize("seed", [3, 17, 604])
def test_result_table_round_trip(oracle, seed):
    """gce_result (rows of emitted records) <-> the per-read form diff_results compares: host-side plumbing of every GPU test."""
    from gencore_amd.batch import table_from_rows
    from parity_helpers import check_output_order
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=50, exotic=seed >= 600)
    want = oracle.run(batch, fuzzgen.make_params(over, contig_len), reference)
    rows = rows_from_table(batch, want)
    assert not check_output_order(batch, rows)
    back = table_from_rows(batch, rows, want.pre, want.post)
    assert not diff_results(batch, back, want)
    swapped = dict(rows)
    if len(rows["src"]) > 1:
        swapped["src"] = rows["src"][::-1].copy()
        assert check_output_order(batch, swapped)


def test_key_range_shards_planned_lean_equal_the_legacy_plan():
    """synth.generate(shard=...) plans a key-range shard without holding the whole stream at pair level (round 6): the rank's records, its global ticks, the stream's
    flush events and the reads' places in the stream are those of the legacy plan, and a world of one is the unsharded stream"""
    import torch
    from gencore_amd import synth
    for name, kw in (("cfg3", dict(n_pairs=12000, scale=0.005)), ("cfg4s", dict(n_pairs=8000, scale=0.003))):
        base = synth.generate(name, seed=5, **kw)
        w1 = synth.generate(name, seed=5, shard=(0, 1), **kw)
        assert all(torch.equal(base.t[k], w1.t[k]) for k in base.t)
        for rank in range(3):
            a = synth.generate(name, seed=5, shard=(rank, 3), legacy_shard=True, **kw)
            b = synth.generate(name, seed=5, shard=(rank, 3), **kw)
            assert all(torch.equal(a.t[k], b.t[k]) for k in a.t), (name, rank)
            assert torch.equal(a.stream_context["tick"], b.stream_context["tick"]) and torch.equal(a.global_index, b.global_index)
            assert (a.stream_context["ev_tid"] == b.stream_context["ev_tid"]).all() and (a.stream_context["ev_pos"] == b.stream_context["ev_pos"]).all()
