"""Parity tests proper: the HIP engine (through the C-ABI) against the CPU oracle on the same inputs — bit-exact
on every emitted record (seq nibbles, quals, NM, qname source, FR/RR, mate links) and on both Stats blocks."""
import numpy as np
import pytest

import fuzzgen
from parity_helpers import check_output_order, diff_results

pytestmark = pytest.mark.gpu


def run_both(batch, params, reference):
    from gencore_amd.engine import run_stream
    from oracle import oracle_py
    want = oracle_py.run(batch, params, reference)
    if want.status != 0:
        from gencore_amd.capi import GceError
        with pytest.raises(GceError) as ei:
            run_stream(batch, params, reference)
        assert ei.value.status == want.status
        return None, want
    got = run_stream(batch, params, reference)
    diffs = diff_results(batch, got, want) + check_output_order(batch, got.rows)
    assert not diffs, "\n".join(diffs)
    assert np.array_equal(got.out_index, np.nonzero(got.out_flag)[0])
    return got, want


@pytest.mark.parametrize("seed", range(32))        # (48 until round 5; the sweeps of tests/fuzz_sweep.py run thousands)
def test_fuzz_stream(built, seed):
    batch, over, reference, contig_len = fuzzgen.make_case(seed)
    run_both(batch, fuzzgen.make_params(over, contig_len), reference)


def test_blob_offsets_beyond_4gb(built):
    """The 32-byte read descriptor keeps 40-bit blob offsets (8 high bits each in one word): a stream whose bases and qualities
    lie behind 4.3 / 4.6 GB of padding must give what the same reads give at offset 0."""
    batch, over, reference, contig_len = fuzzgen.make_case(31, n_mol=50)
    far = batch.copy()
    pad_s, pad_q = (1 << 32) + (5 << 24) + 7, (1 << 32) + (19 << 24) + 1
    far.seq = np.concatenate([np.full(pad_s, 0x11, np.uint8), batch.seq]); far.seq_off = batch.seq_off + np.uint64(pad_s)
    far.qual = np.concatenate([np.full(pad_q, 30, np.uint8), batch.qual]); far.qual_off = batch.qual_off + np.uint64(pad_q)
    params = fuzzgen.make_params(over, contig_len)
    got_far, _ = run_both(far, params, reference)
    got, _ = run_both(batch, params, reference)
    assert np.array_equal(got_far.out_flag, got.out_flag)


@pytest.mark.parametrize("seed,kw", [(33, {}), (107, dict(n_mol=60, umi_mode="duplex", period=7))])
def test_record_layout_one_blob_for_bases_and_qualities(built, seed, kw):
    """gce_batch offsets are free-form: ONE blob with a read's qualities right behind its packed bases (a BAM record's order; bench.py --layout record) and in REVERSED
    read order must give what the two-blob layout gives -- both base pointers of the batch are the same buffer, only the offsets tell bases from qualities."""
    batch, over, reference, contig_len = fuzzgen.make_case(seed, **kw)
    params = fuzzgen.make_params(over, contig_len)
    lq = batch.core["l_qseq"].astype(np.int64)
    sb = (lq + 1) // 2
    rec_len = sb + lq + 3                                    # three pad bytes between records: nothing may depend on neighbours
    order = np.arange(batch.n)[::-1]                        # records laid out back to front
    start = np.zeros(batch.n, np.int64); start[order] = np.cumsum(rec_len[order]) - rec_len[order]
    blob = np.full(int(rec_len.sum()) + 64, 0x5A, np.uint8)
    for i in range(batch.n):
        so, qo = int(batch.seq_off[i]), int(batch.qual_off[i])
        blob[start[i]:start[i] + sb[i]] = batch.seq[so:so + sb[i]]
        blob[start[i] + sb[i]:start[i] + sb[i] + lq[i]] = batch.qual[qo:qo + lq[i]]
    rec = batch.copy()
    rec.seq = blob; rec.qual = blob.copy()                  # (the host path uploads either blob: two copies of the same bytes; the oracle gets its own as well)
    rec.seq_off = start.astype(np.uint64); rec.qual_off = (start + sb).astype(np.uint64)
    got_rec, _ = run_both(rec, params, reference)
    got, _ = run_both(batch, params, reference)
    assert np.array_equal(got_rec.out_flag, got.out_flag)


@pytest.mark.parametrize("seed,umi_mode,period", [(100, "duplex", 10000), (101, "duplex", 11), (102, "prefix", 5), (103, "colon", 3),
                                                    (104, "none", 2), (105, "duplex", 1), (106, "prefix", 10000)])
def test_fuzz_umi_modes(built, seed, umi_mode, period):
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=60, umi_mode=umi_mode, period=period)
    run_both(batch, fuzzgen.make_params(over, contig_len), reference)


@pytest.mark.parametrize("seed", range(600, 612))
def test_fuzz_exotic_nibbles_and_quals(built, seed):
    """IUPAC nibbles and quals >= 128: the register-tally fast kernel must hand these group sides to the generic kernel."""
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=50, exotic=True)
    run_both(batch, fuzzgen.make_params(over, contig_len), reference)


@pytest.mark.parametrize("seed,deep", [(200, 70), (201, 150), (202, 300)])
def test_fuzz_deep_cluster(built, seed, deep):
    """> 64 pairs in one cluster: multi-chunk wave loops; depth 300 also wraps the FR byte (quirk Q8)."""
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=6, umi_mode="duplex" if seed % 2 else "none", deep=deep)
    over["skip_low_complexity_cluster_threshold"] = 100 if seed == 202 else 1000
    got, want = run_both(batch, fuzzgen.make_params(over, contig_len), reference)
    if seed == 200:
        assert got.fr.max() >= 0


@pytest.mark.parametrize("seed,deep,umi_lens", [(220, None, (10, 12)), (221, 40, (9,)), (222, 120, (4,)), (223, 90, (12,))])
def test_duplex_matching_beyond_the_lane_path(built, seed, deep, umi_lens):
    """Cluster::isDuplex (cluster.cpp:246-258) on the paths beside finish_cluster_lanes: UMI tokens beyond eight bytes (the groups' tokens do not fit the
    64-bit words of the lanes: the walk over memory), and a deep cluster whose 4-base UMIs with errors fall into many groups (beyond 64: the same walk)."""
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=12, umi_mode="duplex", deep=deep, umi_lens=umi_lens)
    over["duplex_only"] = 0; over["disable_duplex"] = 0; over["duplex_mismatch_threshold"] = 80          # (so that partners found are merged, not dropped)
    over["skip_low_complexity_cluster_threshold"] = 1000
    run_both(batch, fuzzgen.make_params(over, contig_len), reference)


def odd_duplex_umis(rng, n, tok_lens):
    """n distinct UMIs in families that Cluster::isDuplex (cluster.cpp:246-258) tells apart only through util.h's split (util.h:59-88): leading separators are
    skipped, a doubled separator gives an empty token, a trailing one an empty last token."""
    tok = lambda: "".join(rng.choice("ACGT") for _ in range(rng.choice(tok_lens)))
    umis = []
    while len(umis) < n:
        a, b = tok(), tok()
        fam = rng.choice([[a + "_" + b, b + "_" + a], ["_" + a + "_" + b, b + "_" + a], [a + "_" + b, b + "_" + a, "_" + b + "_" + a], [a + "__" + b, b + "_" + a],
                          [a + "_" + b + "_", b + "_" + a], [a + "_", "_" + a, a], ["_"], ["__"], [""], [a + "_" + a], [a + "_" + b], ["__" + a + "_" + b, "_" + b + "_" + a]])
        for u in fam:
            if u not in umis and len(umis) < n:
                umis.append(u)
    rng.shuffle(umis)
    return umis


@pytest.mark.parametrize("seed,n_groups,tok_lens,via_mi", [(700, 12, (4,), False), (701, 64, (8,), False), (702, 65, (8,), False), (703, 20, (9,), False),
                                                             (704, 64, (8, 9), True), (705, 70, (3,), True), (706, 30, (7, 8), True), (707, 5, (8,), False)])
def test_odd_shaped_duplex_umis_on_both_duplex_paths(built, seed, n_groups, tok_lens, via_mi):
    """Odd-shaped duplex UMIs ('_A_B', 'A__B', 'A_B_', 'A_', '_', tokens of exactly 8 and 9 bytes) reach the GPU duplex stage on BOTH of its paths: the groups in the
    lanes (finish_cluster_lanes: <= 64 groups per cluster, tokens <= 8 bytes as zero-padded words) and the walk over memory (65 groups, or a longer token) -- from read
    names in prefix mode (bamutil.cpp:45-63 takes the whole [ATCG_] run) and from MI:Z tags (bamutil.cpp:23-38: the same parser).  Engine vs oracle; the oracle's
    tokenizer is pinned by the reference's own split (tests/test_oracle_known_answers.py)."""
    import random
    from gencore_amd.batch import ReadBatch
    from gencore_amd.capi import default_params
    from oracle import oracle_py
    rng = random.Random(seed)
    contig = "".join(rng.choice("ACGT") for _ in range(4000))
    recs, serial = [], 0
    for c in range(3):
        pos, L = 100 + 900 * c, 30
        mpos = pos + 200
        n_g = n_groups if c < 2 else max(2, n_groups // 3)
        for gi, u in enumerate(odd_duplex_umis(rng, n_g, tok_lens)):
            for d in range(rng.choice([1, 1, 2])):
                serial += 1
                mut = lambda s: "".join(rng.choice("ACGT") if rng.random() < 0.03 else ch for ch in s)
                name = "r%d" % serial
                kw = {}
                if via_mi:
                    kw["mi"] = "UMI_" + u
                else:
                    name += ":UMI_" + u
                ql = [rng.choice([37, 37, 37, 20, 10]) for _ in range(L)]
                qr = [rng.choice([37, 37, 37, 20, 10]) for _ in range(L)]
                recs.append(dict(qname=name, flag=99, tid=0, pos=pos, cigar="%dM" % L, mtid=0, mpos=mpos, isize=mpos + L - pos, seq=mut(contig[pos:pos + L]), qual=ql, nm=0, **kw))
                recs.append(dict(qname=name, flag=147, tid=0, pos=mpos, cigar="%dM" % L, mtid=0, mpos=pos, isize=-(mpos + L - pos), seq=mut(contig[mpos:mpos + L]), qual=qr, nm=0, **kw))
    recs.sort(key=lambda r: r["pos"])
    batch = ReadBatch.from_records(recs)
    tl = np.asarray([len(contig)], np.uint32)
    for thr, dthr in ((0, 80), (1, 2)):
        prm = default_params(n_targets=1, target_len=tl.ctypes.data, umi_prefix="UMI", proper_umi_diff_threshold=thr, duplex_mismatch_threshold=dthr,
                             flush_period=[10000, 7][thr])
        got, want = run_both(batch, prm, [(oracle_py.pack_reference(contig), len(contig))])
        if thr == 0 and n_groups >= 12:
            assert want.post.as_dict()["dcs"] > 0 and want.post.as_dict()["sscs"] > 0


@pytest.mark.parametrize("depth,with_mates", [(40, 0.0), (70, 0.1), (33, 0.0)])
def test_singleton_pair_slots_of_handed_on_groups(built, depth, with_mates):
    """A stream of pairs that hold ONE read each (the mate is mapped nearby but absent from the stream: a read1-only or region-extracted deep amplicon BAM) in groups
    beyond 32 pairs: k_vote hands every such group on and lists all of its pair slots for k_score2 -- more than N / 2 slots (one per read), which the list must hold
    (it was sized for N / 2 + 16 entries)."""
    import random
    from gencore_amd.batch import ReadBatch
    from gencore_amd.capi import default_params
    rng = random.Random(depth)
    contig = "".join(rng.choice("ACGT") for _ in range(30000))
    recs = []
    for c in range(40):
        pos, L = 200 + 600 * c, 40
        mpos = pos + 150
        for d in range(depth):
            seq = "".join(rng.choice("ACGT") if rng.random() < 0.02 else ch for ch in contig[pos:pos + L])
            q = [rng.choice([37, 37, 30, 12]) for _ in range(L)]
            r1 = dict(qname="s%d_%d" % (c, d), flag=99, tid=0, pos=pos, cigar="%dM" % L, mtid=0, mpos=mpos, isize=mpos + L - pos, seq=seq, qual=q, nm=0)
            recs.append(r1)
            if rng.random() < with_mates:
                recs.append(dict(r1, flag=147, pos=mpos, mpos=pos, isize=-(mpos + L - pos), seq=contig[mpos:mpos + L]))
    recs.sort(key=lambda r: r["pos"])
    batch = ReadBatch.from_records(recs)
    tl = np.asarray([len(contig)], np.uint32)
    from oracle import oracle_py
    got, want = run_both(batch, default_params(n_targets=1, target_len=tl.ctypes.data), [(oracle_py.pack_reference(contig), len(contig))])
    assert len(got.emitted()) >= 40


@pytest.mark.parametrize("seed,deep,umi_mode", [(210, 2300, "duplex"), (211, 2700, "none"), (212, 5200, "prefix")])
def test_cluster_beyond_the_lds_pairing_kernel(built, seed, deep, umi_mode):
    """> 4096 reads in one cluster: k_pairing_deep's LDS instantiation leaves it for its size, the device-memory instantiation (same
    algorithm, arrays in a slab per block, up to 65 534 reads) takes it -- not the generic O(n^2) kernels."""
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=4, umi_mode=umi_mode, deep=deep)
    over["skip_low_complexity_cluster_threshold"] = 1000
    run_both(batch, fuzzgen.make_params(over, contig_len), reference)


def synth_case(name, n_pairs, **over):
    from gencore_amd import synth
    from gencore_amd.capi import default_params
    d = synth.generate(name, n_pairs=n_pairs)
    batch = d.to_batch()
    tl = np.asarray(d.target_len, np.uint32)
    prm = default_params(n_targets=len(tl), target_len=tl.ctypes.data, umi_prefix=d.info["umi_prefix"],
                         cluster_size_req=d.info["supporting_reads"], **over)
    prm._keep = tl
    return batch, prm, d.reference_host()


@pytest.mark.parametrize("name,n_pairs", [("cfg1s", None), ("cfg2", 60000), ("cfg3", 60000), ("cfg4s", 40000), ("cfg5", 3000)])
def test_synthetic_configs(built, name, n_pairs):
    """Down-scaled BASELINE.json configs (same generator, same flags) at sizes the oracle finishes in seconds."""
    batch, prm, ref = synth_case(name, n_pairs)
    got, want = run_both(batch, prm, ref)
    assert len(got.emitted()) > 0


def test_cfg5_many_deep_clusters(built):
    """BASELINE configs[4] at 72 molecules: 72 clusters of >= 500 pairs each go through the deep-cluster kernels (gce_deep.hpp:
    block-per-cluster pairing, block-per-side votes), and one group has 1503 pairs at the DEFAULT
    skipLowComplexityClusterThreshold of 1000 -- the low-complexity check and the early break of group.cpp:142-175,231-232
    run on natural data."""
    import collections
    batch, prm, ref = synth_case("cfg5", 90000)
    core = batch.core
    left = np.where(core["isize"] < 0, core["mpos"], core["pos"])
    groups, clusters = collections.Counter(), collections.Counter()
    for i in np.nonzero(core["flag"] & 64)[0]:
        q = batch.qname_of(int(i))
        key = (int(core["tid"][i]), int(left[i]), abs(int(core["isize"][i])))
        groups[key + (q[q.rfind("UMI_") + 4:],)] += 1
        clusters[key] += 1
    assert sum(1 for v in clusters.values() if v >= 500) >= 50 and max(groups.values()) > 1000
    got, want = run_both(batch, prm, ref)
    assert len(got.emitted()) > 1000


def test_sharded_stream_context(built):
    """tick_offset / trailing_flush (coordinate-sharded multi-GPU runs): per-contig slices reproduce the whole stream."""
    from gencore_amd.shard import shard_by_contig
    batch, over, reference, contig_len = fuzzgen.make_case(300, n_mol=80, umi_mode="prefix", period=17)
    whole_p = fuzzgen.make_params(over, contig_len)
    got_whole, _ = run_both(batch, whole_p, reference)
    flags = np.zeros(batch.n, np.uint8)
    for rank in range(2):
        sub, idx, ctx = shard_by_contig(batch, 2, rank, over["flush_period"])
        p = fuzzgen.make_params(dict(over, **ctx), contig_len)
        got, _ = run_both(sub, p, reference)
        flags[idx] = got.out_flag
    assert np.array_equal(flags, got_whole.out_flag)


def test_multiple_submits_concatenate(built):
    from gencore_amd.engine import Engine
    from oracle import oracle_py
    batch, over, reference, contig_len = fuzzgen.make_case(301, n_mol=50, umi_mode="prefix", period=13)
    prm = fuzzgen.make_params(over, contig_len)
    want = oracle_py.run(batch, prm, reference)
    from gencore_amd.shard import slice_batch
    e = Engine(prm)
    for tid, (nib, ln) in enumerate(reference):
        if nib is not None:
            e.set_reference(tid, nib, ln)
    cut = batch.n // 3
    e.add_reads(slice_batch(batch, np.arange(0, cut)))
    e.add_reads(slice_batch(batch, np.arange(cut, batch.n)))
    e.finish()
    got = e.output(batch)
    with pytest.raises(Exception):          # a second gce_process without a new submit: the stream was mutated in place
        e.finish()
    e.add_reads(batch)                      # a new submit after a process starts a new stream (no stale sizes)
    e.finish()
    again = e.output(batch)
    e.close()
    assert not diff_results(batch, got, want) and not diff_results(batch, again, want)


def test_error_codes_match_reference_fatal_paths(built):
    from gencore_amd.batch import ReadBatch
    from gencore_amd.capi import default_params
    tl = np.asarray([100000], np.uint32)
    base = dict(flag=99, tid=0, cigar="20M", mtid=0, isize=50, seq="ACGTACGTACGTACGTACGT", qual=[37] * 20, nm=0)

    def status_of(recs, **over):
        got, want = run_both(ReadBatch.from_records(recs), default_params(n_targets=1, target_len=tl.ctypes.data, **over), [])
        return want.status          # run_both already asserted that the engine raised the same status

    # unsorted input: src/gencore.cpp:233-241
    assert status_of([dict(base, qname="a", pos=500, mpos=530), dict(base, qname="b", pos=100, mpos=130)]) == -10
    # mates whose MI tags give different UMIs: src/pair.cpp:201-212 (the MI string goes through the same getUMI parser)
    assert status_of([dict(base, qname="a", pos=100, mpos=130, mi="x:AAAA"),
                      dict(base, qname="a", flag=147, pos=130, mpos=100, isize=-50, mi="x:CCCC")]) == -11
    assert status_of([dict(base, qname="a", pos=100, mpos=130, mi="AAAA"),
                      dict(base, qname="a", flag=147, pos=130, mpos=100, isize=-50, mi="CCCC")]) == 0      # no ':' -> both UMIs ""
    # UMI substr throw: src/bamutil.cpp:62
    assert status_of([dict(base, qname="readUI", pos=100, mpos=130)], umi_prefix="UMI") == -13
    # NM absent on a template whose mismatch count changes: src/group.cpp:532-535 dereferences NULL (quirk Q9)
    ref_seq = "ACGTACGTACGTACGTACGT"
    lowq = [37] * 20
    lowq[3] = 2
    bad = ref_seq[:3] + "A" + ref_seq[4:]
    from oracle import oracle_py
    refnib = oracle_py.pack_reference("G" * 100 + ref_seq + "G" * 200)
    recs = [dict(base, qname="a", pos=100, mpos=130, seq=bad, qual=lowq, nm=None),
            dict(base, qname="a", flag=147, pos=130, mpos=100, isize=-50, nm=None)]
    got, want = run_both(ReadBatch.from_records(recs), default_params(n_targets=1, target_len=tl.ctypes.data), [(refnib, 320)])
    assert want.status == -12


@pytest.mark.parametrize("prefix", ["UMI", ""])
def test_umi_parse_window_edges(built, prefix):
    """BamUtil::getUMI edge cases around the engine's 32-byte register window (anchor in front of the window, at its first
    byte, missing, run cut by a foreign character, duplex UMIs, underscore rules of the no-prefix form): every name is one
    pair of one cluster, so the parsed UMIs decide the grouping and the FR tags, which must match the oracle."""
    from gencore_amd.batch import ReadBatch
    from gencore_amd.capi import default_params
    tl = np.asarray([100000], np.uint32)
    seq = "ACGTACGTACGTACGTACGT"
    if prefix:
        names = ["r1:UMI_ACGTAC", "r2:UMI_ACGTAC", "r3:UMI_ACGTAA", "UMI_ACGT" + "x" * 40, "UMI_ACGT" + "y" * 24, "UMI_ACGT" + "z" * 23,
                 "k" * 50, "kk", "abcUA", "q:UMI_ACGTxACGT", "d1:UMI_ACGT_TTGA", "d2:UMI_TTGA_ACGT", "w" * 22 + ":UMI_ACGTAC",
                 "w" * 21 + ":UMI_ACGTAC", "w" * 23 + ":UMI_ACGTAC", "e:UMI_", "f:UMI_ACGTAC:tail", "g" * 40 + ":UMI_GGGGGGGGGGGGGGGGGGGG"]
    else:
        names = ["a:b:ACGT", "a:c:ACGT", "a:d:ACGA", "a:b:ACGT_TTGA", "a:e:TTGA_ACGT", "a:b:_ACGT", "a:b:AC_GT_TT", "a:b:ACGX", "abc", "abc:",
                 ":" + "A" * 40, "n" * 45, "n" * 30 + ":ACGT", "m" * 27 + ":ACGT", "m" * 28 + ":ACGT", "m" * 26 + ":ACGT", "p:q:__A", "p:q:_"]
    recs = []
    for nm in names:
        recs.append(dict(qname=nm, flag=99, tid=0, pos=100, cigar="20M", mtid=0, mpos=130, isize=50, seq=seq, qual=[37] * 20, nm=0))
    for nm in names:
        recs.append(dict(qname=nm, flag=147, tid=0, pos=130, cigar="20M", mtid=0, mpos=100, isize=-50, seq=seq, qual=[37] * 20, nm=0))
    for thr in (0, 1, 2):
        prm = default_params(n_targets=1, target_len=tl.ctypes.data, umi_prefix=prefix, proper_umi_diff_threshold=thr, flush_period=7)
        got, want = run_both(ReadBatch.from_records(recs), prm, [])
        assert want.status == 0 and got.out_flag.sum() > 0


def test_empty_and_tiny_inputs(built):
    from gencore_amd.batch import ReadBatch
    from gencore_amd.capi import default_params
    tl = np.asarray([100000], np.uint32)
    prm = default_params(n_targets=1, target_len=tl.ctypes.data)
    base = dict(flag=99, tid=0, cigar="20M", mtid=0, isize=50, seq="ACGTACGTACGTACGTACGT", qual=[37] * 20, nm=0)
    for recs in ([dict(base, qname="a", pos=100, mpos=130)],
                 [dict(base, qname="a", pos=100, mpos=130), dict(base, qname="a", flag=147, pos=130, mpos=100, isize=-50)],
                 [dict(base, qname="u", flag=77, tid=-1, pos=-1, mtid=-1, mpos=-1, isize=0, cigar="*", nm=None)]):
        run_both(ReadBatch.from_records(recs), prm, [])


def test_full_size_shard_property(built):
    """Size-independent property at scale (cfg3 generator, 400 k pairs / 800 k reads on 24 contigs, ~80 flush events):
    processing the stream as three coordinate shards (with their tick_offset / trailing_flush context) must
    reproduce the whole-stream result table and the additive Stats exactly; and two runs of the same stream are identical."""
    from gencore_amd import synth
    from gencore_amd.capi import default_params
    from gencore_amd.engine import run_stream
    d = synth.generate("cfg3", n_pairs=400000)
    batch = d.to_batch()
    tl = np.asarray(d.target_len, np.uint32)

    def prm(**ctx):
        p = default_params(n_targets=len(tl), target_len=tl.ctypes.data, umi_prefix=d.info["umi_prefix"], cluster_size_req=2, **ctx)
        p._keep = tl
        return p
    ref = d.reference_host()
    whole = run_stream(batch, prm(), ref)
    again = run_stream(batch, prm(), ref)
    assert not diff_results(batch, whole, again)
    # conservation laws of the path
    n_clustered = int((whole.pre.as_dict()["reads"]))
    assert n_clustered == batch.n
    post = whole.post.as_dict()
    assert post["molecules"] == post["sscs"] + post["dcs"] and post["reads"] == int((whole.out_flag != 0).sum())
    assert whole.pre.as_dict()["molecules"] >= post["molecules"]
    from gencore_amd.shard import clustered_mask
    tid = batch.core["tid"].astype(np.int64)
    cm = clustered_mask(batch.core)
    bounds = [0, 6, 14, 24]
    flags = np.zeros(batch.n, np.uint8); fr = np.full(batch.n, -1, np.int16); nm = np.full(batch.n, -1, np.int32)
    pre = np.zeros(114, np.int64); post_sum = np.zeros(114, np.int64)
    total_ticks = int(cm.sum())
    from gencore_amd.shard import slice_contiguous
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        idx = np.nonzero((tid >= lo) & (tid < hi))[0]
        before = int(cm[tid < lo].sum()); mine = int(cm[idx].sum())
        ctx = dict(tick_offset=before, trailing_flush=int(total_ticks // 10000 > (before + mine) // 10000))
        sub = slice_contiguous(batch, int(idx[0]), int(idx[-1]) + 1)
        r = run_stream(sub, prm(**ctx), ref)
        flags[idx], fr[idx], nm[idx] = r.out_flag, r.fr, r.nm_new
        pre += r.pre.as_array(); post_sum += r.post.as_array()
    assert np.array_equal(flags, whole.out_flag) and np.array_equal(fr, whole.fr) and np.array_equal(nm, whole.nm_new)
    assert np.array_equal(pre, whole.pre.as_array()) and np.array_equal(post_sum, whole.post.as_array())




# ------------------------------------------------------------------------------------------------ the zero-copy entry points
def _hip_runtime():
    """The HIP runtime this process already uses (torch's), for plain hipMemcpy of the engine's device-side table."""
    import ctypes as C
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return C.CDLL(line.split()[-1])
    raise RuntimeError("no HIP runtime loaded")


def run_device(data, params, pad=64, shift=0):
    """The path bench.py times: stream built on the GPU (synth), gce_submit_device -> gce_process -> gce_result_device, table
    copied back with torch.  `shift` misaligns every blob by that many bytes; `pad` bytes stay readable behind each blob
    (include/gencore_amd.h: 16 are required)."""
    import ctypes as C
    import torch
    from gencore_amd import capi
    from gencore_amd.batch import table_from_rows
    from gencore_amd.capi import GceBatch, GceResult
    lib = capi.load_library()
    t = data.t
    keep = {}

    def blob(name, dtype=None):
        x = t[name] if dtype is None else t[name].to(dtype)
        y = torch.zeros(x.numel() * x.element_size() + pad + shift + 16, dtype=torch.uint8, device=x.device)
        v = y[shift:shift + x.numel() * x.element_size()]
        v.copy_(x.contiguous().view(torch.uint8).reshape(-1))
        keep[name] = y
        return y.data_ptr() + shift
    b = GceBatch()
    b.n_reads = data.n_reads
    b.core = t["core"].data_ptr()
    b.qname_off, b.cigar_off, b.seq_off, b.qual_off = (t[k].data_ptr() for k in ("qname_off", "cigar_off", "seq_off", "qual_off"))
    b.qname, b.seq, b.qual = blob("qname"), blob("seq"), blob("qual")
    b.cigar = t["cigar"].data_ptr()
    b.nm, b.nm_type, b.mi_off, b.mi, b.tick = t["nm"].data_ptr(), t["nm_type"].data_ptr(), None, None, None
    b.qname_bytes, b.cigar_words, b.seq_bytes, b.qual_bytes, b.mi_bytes = t["qname"].numel(), t["cigar"].numel(), t["seq"].numel(), t["qual"].numel(), 0
    eng = C.c_void_p()
    assert lib.gce_create(C.byref(params), C.byref(eng)) == 0
    try:
        for tid, (nib, ln) in enumerate(data.reference):
            assert lib.gce_set_reference(eng, tid, nib.data_ptr(), ln) == 0
        assert lib.gce_submit_device(eng, C.byref(b)) == 0
        rc = lib.gce_process(eng)
        assert rc == 0, lib.gce_last_error(eng)
        r = GceResult()
        assert lib.gce_result_device(eng, C.byref(r)) == 0
        n = int(r.n_out)

        hip = _hip_runtime()

        def dev(ptr, count, dt):
            if count == 0:
                return np.zeros(0, dt)
            out = np.empty(count, dt)
            assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(out.nbytes), 2) == 0     # hipMemcpyDeviceToHost
            return out
        torch.cuda.synchronize()
        rows = {k: dev(getattr(r, k), n, dt) for k, dt in (("src", np.uint32), ("kind", np.uint8), ("qname_src", np.uint32), ("nm_new", np.int32),
                                                           ("fr", np.int16), ("rr", np.int16), ("mate", np.uint32), ("seq_off", np.uint64), ("qual_off", np.uint64))}
        rows["seq"] = dev(r.seq, int(r.seq_bytes), np.uint8)
        rows["qual"] = dev(r.qual, int(r.qual_bytes), np.uint8)
        pre, post = capi.GceStats(), capi.GceStats()
        C.memmove(C.byref(pre), C.byref(r.pre), C.sizeof(capi.GceStats))
        C.memmove(C.byref(post), C.byref(r.post), C.sizeof(capi.GceStats))
        host = data.to_batch()
        return host, table_from_rows(host, rows, pre, post)
    finally:
        lib.gce_destroy(eng)


@pytest.mark.parametrize("name,n_pairs,shift", [("cfg2", 60000, 0), ("cfg3", 60000, 0), ("cfg5", 50000, 0), ("cfg3", 30000, 3), ("cfg1s", None, 1)])
def test_device_entry_points_against_oracle(built, name, n_pairs, shift):
    """gce_submit_device / gce_result_device (caller-owned HBM, zero copy) — the entry points bench.py times — against the oracle;
    shift != 0: every blob misaligned and exactly 16 readable bytes behind it (the documented contract)."""
    import torch
    from gencore_amd import synth
    from gencore_amd.capi import default_params
    from oracle import oracle_py
    d = synth.generate(name, n_pairs=n_pairs, device="cuda")
    tl = np.asarray(d.target_len, np.uint32)
    prm = default_params(n_targets=len(tl), target_len=tl.ctypes.data, umi_prefix=d.info["umi_prefix"], cluster_size_req=d.info["supporting_reads"])
    host, got = run_device(d, prm, pad=16 if shift else 64, shift=shift)
    want = oracle_py.run(host, prm, d.reference_host())
    assert want.status == 0
    diffs = diff_results(host, got, want) + check_output_order(host, got.rows)
    assert not diffs, "\n".join(diffs)
    assert len(got.emitted()) > 0


@pytest.mark.parametrize("seed,world,mode", [(700, 2, "range"), (701, 3, "range"), (702, 4, "lpt"), (703, 5, "range")])
def test_key_range_shards_equal_whole_stream(built, seed, world, mode):
    """Shards cut by cluster key INSIDE a contig (gencore_amd/shard.py: gce_batch.tick + gce_set_flush_events): every shard on the
    engine equals the oracle on the same shard, and the shards together equal the whole stream."""
    from gencore_amd.engine import run_stream
    from gencore_amd.shard import plan_shards, shard_by_plan, stream_context
    from oracle import oracle_py
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=90, umi_mode="prefix", period=[11, 29, 5][seed % 3])
    prm = fuzzgen.make_params(over, contig_len)
    whole, _ = run_both(batch, prm, reference)
    tick, et, ep = stream_context(batch.core, over["flush_period"])
    plan = plan_shards(batch.core, world, mode)
    flags = np.zeros(batch.n, np.uint8); fr = np.full(batch.n, -1, np.int16)
    pre = np.zeros(114, np.int64); post = np.zeros(114, np.int64)
    for r in range(world):
        sub, idx = shard_by_plan(batch, plan, r, tick)
        got = run_stream(sub, prm, reference, events=(et, ep))
        want = oracle_py.run(sub, prm, reference, events=(et, ep))
        assert not diff_results(sub, got, want)
        flags[idx], fr[idx] = got.out_flag, got.fr
        pre += got.pre.as_array(); post += got.post.as_array()
    assert np.array_equal(flags, whole.out_flag) and np.array_equal(fr, whole.fr)
    assert np.array_equal(pre, whole.pre.as_array()) and np.array_equal(post, whole.post.as_array())


def test_key_range_shards_at_scale(built):
    """cfg3 generator, 200 k pairs cut into 4 key ranges (the cuts fall inside contigs and inside targets)."""
    from gencore_amd import synth
    from gencore_amd.capi import default_params
    from gencore_amd.engine import run_stream
    from gencore_amd.shard import plan_shards, shard_by_plan, stream_context
    d = synth.generate("cfg3", n_pairs=200000)
    batch = d.to_batch()
    tl = np.asarray(d.target_len, np.uint32)
    prm = default_params(n_targets=len(tl), target_len=tl.ctypes.data, umi_prefix=d.info["umi_prefix"], cluster_size_req=2)
    ref = d.reference_host()
    whole = run_stream(batch, prm, ref)
    tick, et, ep = stream_context(batch.core, 10000)
    plan = plan_shards(batch.core, 4, "range")
    flags = np.zeros(batch.n, np.uint8); pre = np.zeros(114, np.int64); post = np.zeros(114, np.int64)
    for r in range(4):
        sub, idx = shard_by_plan(batch, plan, r, tick)
        got = run_stream(sub, prm, ref, events=(et, ep))
        flags[idx] = got.out_flag
        pre += got.pre.as_array(); post += got.post.as_array()
    assert np.array_equal(flags, whole.out_flag)
    assert np.array_equal(pre, whole.pre.as_array()) and np.array_equal(post, whole.post.as_array())


@pytest.mark.gpu
def test_reference_windows_equal_whole_contigs(built):
    """Per-shard reference staging: an engine that holds only the bases its reads can touch -- per contig the window
    [first read position (rounded down to even), last read end) -- gives the result of the engine that holds the whole contigs;
    a window that cuts a read off is refused (GCE_ERR_REF_WINDOW), never read past."""
    from gencore_amd import synth
    from gencore_amd.capi import GceError, default_params
    from gencore_amd.engine import Engine, run_stream
    from test_cabi_driver import ascii_of
    d = synth.generate("cfg3", n_pairs=20000)
    b = d.to_batch()
    tl = np.asarray(d.target_len, np.uint32)
    prm = default_params(n_targets=len(tl), target_len=tl.ctypes.data, umi_prefix=d.info["umi_prefix"], cluster_size_req=d.info["supporting_reads"])
    ref = d.reference_host()
    want = run_stream(b, prm, ref)
    asc = ascii_of(ref)
    tid, pos = b.core["tid"].astype(np.int64), b.core["pos"].astype(np.int64)
    end = pos + b.core["l_qseq"].astype(np.int64) + 64                       # (soft clips aside a read spans at most its length + deletions)
    for shrink in (0, 1):
        e = Engine(prm)
        try:
            for t in range(len(tl)):
                sel = (tid == t) & (pos >= 0)
                if asc[t] is None or not sel.any():
                    continue
                lo = int(pos[sel].min()) & ~1
                hi = min(int(end[sel].max()), int(tl[t]))
                if shrink:
                    hi = max(lo + 2, lo + (hi - lo) // 2)
                e.set_reference_window(t, int(tl[t]), lo, asc[t][lo:hi])
            e.add_reads(b)
            if shrink:
                with pytest.raises(GceError) as err:
                    e.finish()
                assert err.value.status == -15
            else:
                e.finish()
                got = e.output(b)
                assert not diff_results(b, got, want)
        finally:
            e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,n_pairs,world,mode,period", [("cfg3", 40000, 4, "range", 10000), ("cfg3", 20000, 3, "range", 977), ("cfg5", 20000, 4, "lpt", 10000),
                                                          ("cfg2", 30000, 8, "range", 1500), ("cfg1s", None, 2, "lpt", 10000)])
def test_c_planner_equals_the_python_spec(built, name, n_pairs, world, mode, period):
    """gce_stream_context / gce_plan_shards (GPU kernels behind the C-ABI) against gencore_amd/shard.py on whole streams: ticks, flush
    events and the shard of every read, both plan modes."""
    from gencore_amd import shard, synth
    d = synth.generate(name, n_pairs=n_pairs)
    core = d.to_batch().core
    tick, et, ep = shard.stream_context(core, period)
    gtick, get, gep = shard.stream_context_gpu(core, period)
    cm = shard.clustered_mask(core)
    assert np.array_equal(tick, gtick) and np.array_equal(et, get) and np.array_equal(ep, gep) and len(et) == int(cm.sum()) // period
    assert np.array_equal(shard.plan_shards(core, world, mode), shard.plan_shards_gpu(core, world, mode))


@pytest.mark.gpu
@pytest.mark.parametrize("hostcodec", [False, True])
@pytest.mark.parametrize("workload,n_pairs,shards,mode,period", [("cfg3", 30000, 3, 0, 2000), ("cfg5", 3000, 2, 1, 10000), ("cfg2", 20000, 4, 0, 700)])
def test_sharded_bam_run_equals_the_single_engine(built, tmp_path, monkeypatch, workload, n_pairs, shards, mode, period, hostcodec):
    """gce_run_bam_sharded writes the same records in the same order with the same Stats as gce_run_bam.  Default: the GPU codec -- every
    engine (all on device 0 here) receives the compressed pieces, inflates, indexes and plans the stream itself and keeps its shard
    (gce_raw_select_shard), the record streams are merged device to device (gce_raw_merge_outputs), Stats summed in device memory.
    hostcodec: round 2's runner (host inflate / index / cut, per-shard reference windows, host merge)."""
    import pybam
    from gencore_amd import synth
    from gencore_amd.bamio import run_bam, run_bam_sharded
    from gencore_amd.capi import default_params
    from test_bamio import records_of
    from test_cabi_driver import ascii_of
    d = synth.generate(workload, n_pairs=n_pairs)
    batch = d.to_batch()
    tl = np.asarray(d.target_len, np.uint32)
    targets = [("chr%d" % (i + 1), int(l)) for i, l in enumerate(tl)]
    src, one, many, fa = (str(tmp_path / x) for x in ("in.bam", "one.bam", "many.bam", "ref.fa"))
    pybam.write_bam(src, records_of(batch), targets)
    with open(fa, "wb") as f:
        for (nm, _), bases in zip(targets, ascii_of(d.reference_host())):
            if bases is not None:
                f.write(b">" + nm.encode() + b"\n" + bases + b"\n")
    prm = default_params(umi_prefix="auto", cluster_size_req=d.info["supporting_reads"], flush_period=period)
    r1 = run_bam(src, one, prm, fasta=fa, threads=4)
    if hostcodec:
        monkeypatch.setenv("GCE_BAM_HOSTCODEC", "1")
    r2 = run_bam_sharded(src, many, prm, [0] * shards, fasta=fa, plan_mode=mode, threads=4)
    monkeypatch.delenv("GCE_BAM_HOSTCODEC", raising=False)
    assert (r1.n_reads, r1.n_out) == (r2.n_reads, r2.n_out) and r1.n_out > 0
    assert bytes(r1.pre) == bytes(r2.pre) and bytes(r1.post) == bytes(r2.post)
    _, t1, g1 = pybam.read_bam(one)
    _, t2, g2 = pybam.read_bam(many)
    assert t1 == t2 and len(g1) == len(g2)
    for a, b in zip(g1, g2):                                                 # record by record, in file order
        assert a == b


@pytest.mark.gpu
@pytest.mark.parametrize("hostcodec", [False, True])
@pytest.mark.parametrize("max_contig,shards,mode", [(5, 3, 0), (4, 4, 1), (1, 2, 0)])
def test_sharded_bam_run_with_quit_after_contig(built, tmp_path, monkeypatch, max_contig, shards, mode, hostcodec):
    """--quit_after_contig (gce_params.max_contig, src/gencore.cpp:243-246) ends the loop ONCE: the first read of the whole stream with tid >= maxContig
    is counted by the pre-Stats, nothing behind it exists.  The sharded runners make that cut on the whole stream in front of the plan (every
    shard cutting its own part counted one extra read per shard that held a read of the later contigs): same records, same order, same Stats as
    gce_run_bam.  max_contig 1: only reads of the first contig are left, the plan gives most shards nothing."""
    import pybam
    from gencore_amd import synth
    from gencore_amd.bamio import run_bam, run_bam_sharded
    from gencore_amd.capi import default_params
    from test_bamio import records_of
    d = synth.generate("cfg3", n_pairs=30000)
    batch = d.to_batch()
    tl = np.asarray(d.target_len, np.uint32)
    targets = [("chr%d" % (i + 1), int(l)) for i, l in enumerate(tl)]
    src, one, many = (str(tmp_path / x) for x in ("in.bam", "one.bam", "many.bam"))
    pybam.write_bam(src, records_of(batch), targets)
    prm = default_params(umi_prefix="auto", cluster_size_req=d.info["supporting_reads"], flush_period=1500, max_contig=max_contig)
    r1 = run_bam(src, one, prm, threads=4)
    n_front = int((batch.core["tid"][batch.core["tid"] >= 0] < max_contig).sum())
    assert int(r1.pre.reads) == n_front + 1                                  # the cut read is counted, once
    if hostcodec:
        monkeypatch.setenv("GCE_BAM_HOSTCODEC", "1")
    r2 = run_bam_sharded(src, many, prm, [0] * shards, plan_mode=mode, threads=4)
    monkeypatch.delenv("GCE_BAM_HOSTCODEC", raising=False)
    assert (r1.n_reads, r1.n_out) == (r2.n_reads, r2.n_out) and r1.n_out > 0
    assert bytes(r1.pre) == bytes(r2.pre) and bytes(r1.post) == bytes(r2.post)
    _, t1, g1 = pybam.read_bam(one)
    _, t2, g2 = pybam.read_bam(many)
    assert t1 == t2 and len(g1) == len(g2)
    for a, b in zip(g1, g2):
        assert a == b


def test_stats_blocks_in_device_memory_equal_the_drained_ones(built):
    """gce_stats_device: the two Stats blocks as they lie in HBM (what a multi-GPU run all-reduces, bench.py) are the ones gce_drain returns."""
    import ctypes as C
    from gencore_amd import capi
    from gencore_amd.engine import Engine
    batch, over, reference, contig_len = fuzzgen.make_case(77, n_mol=120, umi_mode="duplex")
    E = Engine(fuzzgen.make_params(over, contig_len))
    try:
        E.run(batch, reference)
        rows, pre, post = E.rows()
        ptr = C.c_void_p()
        assert E.lib.gce_stats_device(E._h, C.byref(ptr)) == 0 and ptr.value
        hip = None
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                hip = C.CDLL(line.split()[-1]); break
        assert hip is not None
        host = np.zeros(2 * capi.GCE_STATS_WORDS, np.int64)
        assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), ptr, C.c_size_t(host.nbytes), 2) == 0
        assert np.array_equal(host[:capi.GCE_STATS_WORDS], pre.as_array()) and np.array_equal(host[capi.GCE_STATS_WORDS:], post.as_array())
        assert host[0] == batch.n
    finally:
        E.close()


@pytest.mark.parametrize("seed", [3, 8, 17, 26, 31, 44, 601, 606])
def test_two_stream_order_on_small_streams(built, seed, monkeypatch):
    """A stream of deep groups runs Pair::computeScore (k_score2) on a second HIP stream beside the hand-on and the preparation of the deep sides
    (gce_process).  GCE_FORCE_AUX_STREAM takes that order on the suite's small streams, deep or not."""
    monkeypatch.setenv("GCE_FORCE_AUX_STREAM", "1")
    kw = dict(exotic=seed >= 600)
    if seed % 2:
        kw["deep"] = 80
    batch, over, reference, contig_len = fuzzgen.make_case(seed, **kw)
    run_both(batch, fuzzgen.make_params(over, contig_len), reference)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,umi_mode,use_tick", [(21, "prefix", False), (22, "mi", False), (23, "mi", True), (24, "duplex", True)])
def test_streamed_submit_carries_mi_tags_and_ticks(built, seed, umi_mode, use_tick):
    """gce_reserve + gce_submit_async in several batches (the streamed host path of INTEGRATION 2) with MI:Z tags (src/bamutil.cpp:23-38) and with per-read
    global ticks + the stream's flush events (a key-range shard, src/gencore.cpp:319-322): the same result as the oracle on the whole stream.  Some batches
    of the `mi` streams carry no tag at all (their reads fall back to the name)."""
    import ctypes as C
    from gencore_amd import capi
    from gencore_amd.batch import table_from_rows
    from gencore_amd.engine import Engine
    from gencore_amd.shard import slice_batch, stream_context
    from oracle import oracle_py
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=150, umi_mode=umi_mode, period=17)
    prm = fuzzgen.make_params(over, contig_len)
    want = oracle_py.run(batch, prm, reference)
    assert want.status == 0
    tick = ev = None
    if use_tick:
        tick, et, ep = stream_context(batch.core, over["flush_period"])
        ev = (et, ep)
    E = Engine(prm)
    try:
        for tid, (nib, ln) in enumerate(reference):
            if nib is not None:
                E.set_reference(tid, nib, ln)
        if ev is not None:
            E.set_flush_events(*ev)
        st = batch.as_struct()
        assert E.lib.gce_reserve(E._h, batch.n, st.qname_bytes, st.cigar_words, st.seq_bytes, st.qual_bytes) == 0
        cuts = [0, batch.n // 5, batch.n // 2, batch.n // 2, (3 * batch.n) // 4, batch.n]
        keep = []
        for a, z in zip(cuts[:-1], cuts[1:]):
            sub = slice_batch(batch, np.arange(a, z))
            if tick is not None:
                sub.tick = np.ascontiguousarray(tick[a:z], np.uint64)
            keep.append(sub)
            s2 = sub.as_struct()
            t = C.c_int32(-1)
            rc = E.lib.gce_submit_async(E._h, C.byref(s2), C.byref(t))
            assert rc == 0, E.lib.gce_last_error(E._h)
        E.finish()
        got = E.output(batch)
    finally:
        E.close()
    diffs = diff_results(batch, got, want) + check_output_order(batch, got.rows)
    assert not diffs, "\n".join(diffs)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,umi_mode", [(31, "prefix"), (32, "mi")])
def test_several_device_batches_per_process(built, seed, umi_mode):
    """gce_submit_device more than once before gce_process: the engine appends the batches device to device into its own copy of the stream."""
    import ctypes as C
    import torch
    from gencore_amd.engine import Engine
    from oracle import oracle_py
    batch, over, reference, contig_len = fuzzgen.make_case(seed, n_mol=150, umi_mode=umi_mode, period=23)
    prm = fuzzgen.make_params(over, contig_len)
    want = oracle_py.run(batch, prm, reference)
    assert want.status == 0
    E = Engine(prm)
    try:
        for tid, (nib, ln) in enumerate(reference):
            if nib is not None:
                E.set_reference(tid, nib, ln)
        cuts = [0, batch.n // 3, (2 * batch.n) // 3, batch.n]
        from gencore_amd.capi import GceBatch
        from gencore_amd.shard import slice_batch

        def device_struct(sub):
            st, keep = GceBatch(), []
            st.n_reads = sub.n
            for f in sub.FIELDS:
                a_ = getattr(sub, f)
                if a_ is None or (a_.size == 0 and f in ("mi", "mi_off")):
                    setattr(st, f, None); continue
                raw = np.concatenate([a_.view(np.uint8).reshape(-1), np.zeros(64, np.uint8)])      # (device blobs must be readable past their end)
                t_ = torch.from_numpy(raw).cuda()
                keep.append(t_); setattr(st, f, t_.data_ptr())
            st.qname_bytes, st.cigar_words, st.seq_bytes, st.qual_bytes = sub.qname.size, sub.cigar.size, sub.seq.size, sub.qual.size
            st.mi_bytes = 0 if sub.mi is None else sub.mi.size
            st.tick = None
            return st, keep
        for a, z in zip(cuts[:-1], cuts[1:]):
            sub = slice_batch(batch, np.arange(a, z))
            st, keep = device_struct(sub)
            assert E.lib.gce_submit_device(E._h, C.byref(st)) == 0, E.lib.gce_last_error(E._h)
            if a > 0:
                del keep                                                     # (a batch behind the first may go as soon as the call returns)
            else:
                first_keep = keep
        E.finish()
        got = E.output(batch)
    finally:
        E.close()
    diffs = diff_results(batch, got, want) + check_output_order(batch, got.rows)
    assert not diffs, "\n".join(diffs)
