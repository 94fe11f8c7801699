"""Host-side logic on CPU: SoA batch round trips, generator determinism, stream sharding context (tick_offset /
trailing_flush) checked with the oracle, oracle vs frozen golden vectors."""
import json
import os

import numpy as np
import pytest

import fuzzgen
from gencore_amd.batch import ReadBatch, diff_results

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
