"""One dedicated, hand-derived case per quirk of SURVEY.md section 8's quirk register.

Every expectation below was worked out BY HAND from the cited reference lines (not taken from an oracle run), so the CPU tests
pin the oracle against an independent reading of the reference; the `gpu`-marked test then runs the same streams through the HIP
engine and requires bit-exact agreement with the oracle (tests/test_gpu_parity.py::run_both).

Conventions: one contig of 100 000 bp, 20 bp reads, defaults of src/options.cpp:4-40 (qualities 30/20/15, scores 8/6/4/2,
baseScoreReq 6, scorePercentReq 0.8, -s 1, -d 1, hidden EOF threshold 0, flush every 10 000 clustered reads).
"""
import numpy as np
import pytest

from gencore_amd.batch import ReadBatch
from gencore_amd.capi import default_params

TL = np.asarray([100000], np.uint32)
REF20 = "ACGTACGTACGTACGTACGT"


def prm(**over):
    return default_params(n_targets=1, target_len=TL.ctypes.data, **over)


def rec(qname, flag, pos, mpos, isize, seq=REF20, qual=None, cigar=None, nm=0, **kw):
    d = dict(qname=qname, flag=flag, tid=0, pos=pos, cigar=cigar or "%dM" % len(seq), mtid=0, mpos=mpos, isize=isize, seq=seq,
             qual=qual if qual is not None else [37] * len(seq), nm=nm)
    d.update(kw)
    return d


def pair(qname, left=100, right=130, lseq=REF20, rseq=REF20, lqual=None, rqual=None, **kw):
    """A proper FR pair of 20 bp reads; isize spans both mates."""
    isz = right + len(rseq) - left
    return [rec(qname, 99, left, right, isz, lseq, lqual, **kw), rec(qname, 147, right, left, -isz, rseq, rqual, **kw)]


def by_pos(recs):
    return sorted(recs, key=lambda r: (r["tid"] if r["tid"] >= 0 else 1 << 30, r["pos"]))


def out_records(rt, batch):
    return [r for r in rt.records(batch)]


# ------------------------------------------------------------------------------------------------------------------ cases
def case_q1_periodic_vs_eof(period):
    """Q1 (gencore.cpp:355 vs :409, options.cpp:12-13): two pairs of one cluster whose UMIs differ by one base.  Flushed by the
    periodic walk they merge (threshold -d = 1): ONE consensus pair with FR 2.  Left pending until finishConsensus they do not
    (threshold 0): TWO pairs with FR 1.  The far pair only supplies the 5th clustered read that fires the walk."""
    recs = pair("x:UMI_AAAA") + pair("y:UMI_AAAT") + pair("far:UMI_CCCC", left=5000, right=5030)
    return ReadBatch.from_records(by_pos(recs)), prm(umi_prefix="UMI", flush_period=period), []


def case_q2_tick_skips_mate_unmapped():
    """Q2 (gencore.cpp:307-309,319-322,344-354): a read whose mate is unmapped does not advance the tick, and a walk only takes
    clusters with left < pos AND right < pos of the firing read.  Period 6; stream order:
        A-left x2 (100), B-left (120), U (125, mate unmapped: no tick), A-right x2 (130), B-right (150: tick 6).
    The walk fires on B-right at 150: cluster A = (100..149) is taken there (149 < 150) with -d = 1 and its two UMIs (one base
    apart) merge.  Had U counted, tick 6 would have been A's second right mate at 130, where A is not yet flushable (149 >= 130):
    A would stay pending until finishConsensus (threshold 0) and come out as two pairs."""
    recs = pair("a1:UMI_AAAA") + pair("a2:UMI_AAAT") + pair("b:UMI_GGGG", left=120, right=150)
    recs.append(rec("u:UMI_TTTT", 73, 125, -1, 0, mtid=-1))
    return ReadBatch.from_records(by_pos(recs)), prm(umi_prefix="UMI", flush_period=6), []


def case_q4_dispositions():
    """Q4 (gencore.cpp:255-271,307-309): unmapped reads are counted by preStats and dropped; secondary / supplementary records are
    skipped; a read whose mate is unmapped passes through untouched and untagged."""
    recs = pair("p") + [rec("sec", 99 | 0x100, 200, 230, 50), rec("sup", 99 | 0x800, 300, 330, 50),
                        rec("mu", 73, 400, -1, 0, mtid=-1)]
    recs = by_pos(recs) + [dict(qname="un", flag=77, tid=-1, pos=-1, cigar="*", mtid=-1, mpos=-1, isize=0, seq=REF20, qual=[20] * 20, nm=None)]
    return ReadBatch.from_records(recs), prm(), []


def case_q5_q6_q11_overlap_mismatch():
    """Q5/Q6/Q11 (pair.cpp:155-168, group.cpp:394-417,421-428,503-525): a depth-1 pair still runs the vote (Q11).  The mates
    overlap by 8 bases and disagree at one of them with EQUAL qualities 37: both quals are rewritten to max(0, 37-37) = 0, the left
    score is qual2score(0) - 3 = 2 - 3 = -1 (Q5), the right score 0.  In the column vote every bin then scores <= 0 with qual-sum 0,
    so the `>=` tie rule walks the winner up to bin 15 = 'N' with count 0 (Q6); no reference => the base is written as 'N' with
    quality topQuals[15] = 0 on BOTH mates.  Every other column is unanimous (score 12 in the overlap, 8 outside) and keeps its
    base and its quality 37."""
    lseq = REF20[:14] + "T" + REF20[15:]          # left[14] = 'T' where the right mate (pos 112, same 4-periodic phase) shows REF20[2] = 'G'
    recs = [rec("q", 99, 100, 112, 32, lseq), rec("q", 147, 112, 100, -32, REF20)]
    return ReadBatch.from_records(recs), prm(), []


def _q7_records(n_cols):
    """Four pairs, no UMI, mates far apart (no overlap).  The template of the left side is the first read in qname order ('a').
    At `n_cols` columns 'a' shows the reference base with quality 20 while b, c, d show another base with quality 37."""
    ref = "G" * 100 + REF20 + "G" * 280 + REF20 + "G" * 100            # contig prefix; left reads at 100, right reads at 400
    cols = list(range(2, 2 + n_cols))
    alt = {"A": "C", "C": "A", "G": "T", "T": "G"}
    other = "".join(alt[ch] if i in cols else ch for i, ch in enumerate(REF20))
    qa = [20 if i in cols else 37 for i in range(20)]
    recs = []
    for name in "abcd":
        recs += pair(name, left=100, right=400, lseq=REF20 if name == "a" else other, lqual=qa if name == "a" else None,
                     nm=0 if name == "a" else n_cols)
    return by_pos(recs), ref, cols, other


def case_q7(n_cols):
    """Q7 (group.cpp:442-467,503-573): per such column the other base has 3 votes (score 24, top quality 37 >= high) against one
    moderate-quality reference base: no reference check, the template's base is overwritten and, because it WAS the reference base,
    mismatchInc++.  With 5 columns NM is patched from 0 to 5 (type 'C'); with 6 columns mismatchInc > 5 and the template's seq and
    qual are restored wholesale, NM untouched."""
    recs, ref, cols, other = _q7_records(n_cols)
    from oracle import oracle_py
    return ReadBatch.from_records(recs), prm(), [(oracle_py.pack_reference(ref), len(ref))]


def case_q13_mateless_scores_six():
    """Q13 (pair.cpp:88-107): a pair with only one mate scores every base 6, whatever its quality.  Group of two pairs, 'b' lost its
    right mate.  Column 5 of the left side: template 'a' shows A with quality 15 (score 4), 'b' shows C with quality 2 -- which
    scores 6, not 2 -- so C is the top base (6 > 4); the second base has one low-quality vote, the top has < 2 votes and no high
    quality, its quality 2 is <= low: reference check, but there is no reference => column 5 becomes 'C' with quality 2."""
    la = REF20[:5] + "A" + REF20[6:]
    qa = [37] * 20
    qa[5] = 15
    lb = REF20[:5] + "C" + REF20[6:]
    qb = [37] * 20
    qb[5] = 2
    recs = pair("a", lseq=la, lqual=qa) + [rec("b", 99, 100, 130, 50, lb, qb)]
    return ReadBatch.from_records(by_pos(recs)), prm(), []


# ------------------------------------------------------------------------------------------------------------------ CPU: oracle vs hand
def test_q1_threshold_depends_on_who_flushes(oracle):
    for period, n_out, fr in ((5, 2, 2), (10000, 4, 1)):
        batch, p, ref = case_q1_periodic_vs_eof(period)
        rt = oracle.run(batch, p, ref)
        assert rt.status == 0
        out = [r for r in out_records(rt, batch) if r["pos"] < 1000]
        assert len(out) == n_out and all(r["fr"] == fr for r in out), (period, out)


def test_q2_tick_and_walk_rule(oracle):
    batch, p, ref = case_q2_tick_skips_mate_unmapped()
    rt = oracle.run(batch, p, ref)
    assert rt.status == 0
    a = [r for r in out_records(rt, batch) if r["kind"] == 1 and r["pos"] in (100, 130)]
    assert len(a) == 2 and all(r["fr"] == 2 for r in a), a
    u = [r for r in out_records(rt, batch) if r["kind"] == 2]
    assert len(u) == 1 and u[0]["pos"] == 125


def test_q4_dispositions(oracle):
    batch, p, ref = case_q4_dispositions()
    rt = oracle.run(batch, p, ref)
    assert rt.status == 0
    kinds = {batch.qname_of(i): int(rt.out_flag[i]) for i in range(batch.n)}
    assert kinds == {"p": 1, "sec": 0, "sup": 0, "mu": 2, "un": 0}
    mu = [r for r in out_records(rt, batch) if r["qname"] == "mu"][0]
    assert mu["fr"] == -1 and mu["rr"] == -1 and mu["seq"] == REF20 and mu["qual"] == [37] * 20
    assert rt.pre.as_dict()["reads_unmapped"] == 1 and rt.pre.as_dict()["reads"] >= 4      # the unmapped read was counted, then dropped


def test_q5_q6_q11_overlap_mismatch_becomes_N(oracle):
    batch, p, ref = case_q5_q6_q11_overlap_mismatch()
    rt = oracle.run(batch, p, ref)
    assert rt.status == 0
    out = sorted(out_records(rt, batch), key=lambda r: r["pos"])
    assert len(out) == 2 and all(r["fr"] == 1 for r in out)
    left, right = out
    assert left["seq"] == REF20[:14] + "N" + REF20[15:] and right["seq"] == REF20[:2] + "N" + REF20[3:]
    assert left["qual"] == [37] * 14 + [0] + [37] * 5 and right["qual"] == [37] * 2 + [0] + [37] * 17


@pytest.mark.parametrize("n_cols", [5, 6])
def test_q7_nm_patch_or_restore(oracle, n_cols):
    batch, p, ref = case_q7(n_cols)
    recs, _, cols, other = _q7_records(n_cols)
    rt = oracle.run(batch, p, ref)
    assert rt.status == 0
    out = sorted(out_records(rt, batch), key=lambda r: r["pos"])
    assert len(out) == 2 and all(r["fr"] == 4 for r in out)
    left = out[0]
    assert left["qname"] == "a"
    if n_cols == 5:
        assert left["seq"] == other and left["qual"] == [37] * 20 and left["nm"] == 5
    else:
        assert left["seq"] == REF20 and left["qual"] == [20 if i in cols else 37 for i in range(20)] and left["nm"] == 0
    assert out[1]["seq"] == REF20 and out[1]["nm"] == 0


def test_q13_mateless_pair_scores_six(oracle):
    batch, p, ref = case_q13_mateless_scores_six()
    rt = oracle.run(batch, p, ref)
    assert rt.status == 0
    left = [r for r in out_records(rt, batch) if r["pos"] == 100]
    assert len(left) == 1 and left[0]["fr"] == 2
    assert left[0]["seq"] == REF20[:5] + "C" + REF20[6:] and left[0]["qual"][5] == 2


# ------------------------------------------------------------------------------------------------------------------ GPU: engine vs oracle
@pytest.mark.gpu
@pytest.mark.parametrize("case", ["q1p", "q1e", "q2", "q4", "q5", "q7a", "q7b", "q13"])
def test_quirk_cases_on_the_engine(built, case):
    from test_gpu_parity import run_both
    batch, p, ref = {"q1p": lambda: case_q1_periodic_vs_eof(5), "q1e": lambda: case_q1_periodic_vs_eof(10000),
                     "q2": case_q2_tick_skips_mate_unmapped, "q4": case_q4_dispositions, "q5": case_q5_q6_q11_overlap_mismatch,
                     "q7a": lambda: case_q7(5), "q7b": lambda: case_q7(6), "q13": case_q13_mateless_scores_six}[case]()
    got, want = run_both(batch, p, ref)
    assert want.status == 0 and got is not None


# ------------------------------------------------------------------------------------------------------------------ --quit_after_contig
TL3 = np.asarray([100000, 100000, 100000], np.uint32)


def case_quit_after_contig(maxc):
    """Options::maxContig (src/gencore.cpp:243-246): the read loop ends on the first read whose tid >= maxContig.  That read has been
    counted by the pre-Stats (:222) before (the sorted check of :233-241 cannot fail on it: every read in front has a smaller tid); it and everything behind it are ignored; pending clusters
    are finished as at the end of the file (threshold 0).  Three contigs, one pair on each; x and y on contig 0 have UMIs one base apart."""
    recs = pair("x:UMI_AAAA") + pair("y:UMI_AAAT")
    for t in (1, 2):
        for r in pair("c%d:UMI_CCCC" % t, left=500, right=530):
            r["tid"] = t; r["mtid"] = t
            recs.append(r)
    recs = by_pos(recs)
    p = default_params(n_targets=3, target_len=TL3.ctypes.data, umi_prefix="UMI", max_contig=maxc)
    p._keep = TL3
    return ReadBatch.from_records(recs), p, []


@pytest.mark.parametrize("maxc,reads_counted,names", [(0, 8, {"x:UMI_AAAA", "y:UMI_AAAT", "c1:UMI_CCCC", "c2:UMI_CCCC"}), (1, 5, {"x:UMI_AAAA", "y:UMI_AAAT"}),
                                                      (2, 7, {"x:UMI_AAAA", "y:UMI_AAAT", "c1:UMI_CCCC"}), (3, 8, {"x:UMI_AAAA", "y:UMI_AAAT", "c1:UMI_CCCC", "c2:UMI_CCCC"})])
def test_quit_after_contig(oracle, maxc, reads_counted, names):
    """By hand: with maxContig = 1 the loop sees the four reads of contig 0 and the FIRST read of contig 1 (counted: 5 reads, 100 bases),
    then breaks; x and y are finished with the end-of-file threshold 0 (two pairs, FR 1 each); nothing of contigs 1 and 2 is written."""
    batch, p, ref = case_quit_after_contig(maxc)
    rt = oracle.run(batch, p, ref)
    assert rt.status == 0
    assert rt.pre.as_dict()["reads"] == reads_counted and rt.pre.as_dict()["bases"] == 20 * reads_counted
    recs = out_records(rt, batch)
    assert {r["qname"].rstrip("\0") for r in recs} == names and len(recs) == 2 * len(names)
    assert all(r["fr"] == 1 for r in recs)


@pytest.mark.gpu
@pytest.mark.parametrize("maxc", [0, 1, 2, 3, 7])
def test_quit_after_contig_on_the_engine(built, maxc):
    from test_gpu_parity import run_both
    batch, p, ref = case_quit_after_contig(maxc)
    got, want = run_both(batch, p, ref)
    assert got.pre.as_dict()["reads"] == want.pre.as_dict()["reads"]
