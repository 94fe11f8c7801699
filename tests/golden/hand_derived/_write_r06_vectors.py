#!/usr/bin/env python
"""Writes the round-6 hand-derived vectors and adds hand-derived `expected_stats` blocks to four older ones.  A WRITING AID, not an
oracle: every `expected` block, every `expected_stats` block and every derivation was worked out by hand from the cited reference
lines; no oracle or engine run is involved."""
import json, os
HERE = os.path.dirname(os.path.abspath(__file__))
Q = lambda q, n: [q] * n


def rec(qname, flag, pos, cigar, mpos, isize, seq, qual, tid=0, mtid=0, nm=0, **kw):
    return dict(qname=qname, flag=flag, tid=tid, pos=pos, cigar=cigar, mtid=mtid, mpos=mpos, isize=isize, seq=seq, qual=qual, nm=nm, **kw)


def out(qname, flag, pos, cigar, seq, qual, nm, fr, rr=-1, tid=0):
    return dict(qname=qname, flag=flag, tid=tid, pos=pos, cigar=cigar, seq=seq, qual=qual, nm=nm, fr=fr, rr=rr)


def write(v):
    with open(os.path.join(HERE, v["name"] + ".json"), "w") as f:
        json.dump(v, f, indent=1)
        f.write("\n")


def stats(reads, bases, clusters, multi, molecules, hist, sscs=0, dcs=0, unmapped=(0, 0), mism=(0, 0), se=0):
    return dict(reads=reads, bases=bases, reads_unmapped=unmapped[0], bases_unmapped=unmapped[1], base_mismatches=mism[0], reads_with_mismatches=mism[1],
                clusters=clusters, multi_molecule_clusters=multi, molecules=molecules, molecules_se=se, molecules_pe=molecules - se, sscs=sscs, dcs=dcs,
                supporting_hist={str(k): v for k, v in hist.items()})


C0 = [dict(name="c0", length=100000)]
unmapped = lambda name: dict(qname=name, flag=4, tid=-1, pos=-1, cigar="*", mtid=-1, mpos=-1, isize=0, seq="ACGTACGTACGTACGTACGT", qual=Q(20, 20), nm=None)

# ------------------------------------------------------------------------------------------------ 1: posDis < 0 (pair.cpp:115-118)
# reference positions 480..519 of the (absent) contig, as the two reads show them
S = "ACGTTGCAAGCTTCGATGCAAGCTTGCATGCAAGCTAGCT"
R2 = S[0:30]                               # the read at 480, 30M: columns 20..29 lie on 500..509
R1 = S[20:23] + "A" + S[24:40]             # the read at 500, 20M: column 3 (position 503) shows A where the read at 480 shows T (its column 23)
assert S[23] == "T" and len(R1) == 20
q2 = Q(30, 30); q2[25] = 10
write(dict(
    name="overlap_with_the_right_read_left_of_the_left_read",
    cites=["src/gencore.cpp:232-266", "src/cluster.cpp:260-273", "src/pair.cpp:108-120", "src/pair.cpp:140-168", "src/pair.cpp:77-86", "src/group.cpp:395-428",
           "src/group.cpp:497-526", "src/gencore.cpp:319-352", "src/gencore.cpp:276-279"],
    derivation=(
        "How `posDis < 0` (pair.cpp:115-118) is reached at all: Cluster::addRead makes the FIRST read of a name the pair's mLeft (cluster.cpp:260-273), and in a sorted "
        "stream the first read has the smaller position -- but the sortedness test (gencore.cpp:232-241) compares with lastTid / lastPos, which EVERY read sets, an unmapped one "
        "(tid -1, pos -1: itself exempt from the test) included.  Flush period 3 (the literal 10000 of gencore.cpp:321).  Record 0: unmapped read u1: the output set was never "
        "cleared, so finishConsensus runs now, on nothing (:255-262): mProperClustersFinished = true, mOutSetCleared = true -- no later unmapped read and no end of file finishes "
        "anything again.  Record 1: x, the REVERSE read of its pair, at 500 (20M, isize -40, mate at 480): key left = mpos = 480, right = 480 + 40 - 1 = 519; tick 1; it is the "
        "pair's mLeft.  Record 2: unmapped read u2: lastTid = lastPos = -1, nothing else (:257 is false).  Record 3: x's forward read at 480 (30M, isize 40): 0:480 after -1:-1 "
        "passes the sortedness test; same key (480, 519); tick 2; it becomes mRight.  Record 4: y at 2000: tick 3 fires the walk with pos 2000: cluster (480, 519) has left < 2000 "
        "and right < 2000 and is taken; y's own cluster (left 2000 >= 2000) stops the walk and is never flushed (no finish at end of file): y is counted by mPreStats->addRead "
        "and by nothing else.  "
        "Pair x, computeScore: both reads are one M block (offsets 0, lengths 20 and 30); posDis = 480 - 500 = -20 < 0: leftStart = 0, rightStart = 0 + 20 = 20, cmpLen = "
        "min(20, 30 - 20) = 10: left columns 0..9 lie on right columns 20..29 (positions 500..509).  Left qualities 37, right qualities 30 except column 25 = 10.  Matches: "
        "q = (37 + 30) / 2 = 33 -> 8 + 4 = 12 on both reads; column 5 / 25: (37 + 10) / 2 = 23 -> 6 + 4 = 10.  Column 3 / 23: A against T, lq 37 >= rq 30: lqual[3] = 7, rqual[23] = "
        "max(0, 30 - 37) = 0, mLeftScore[3] = qual2score(7) - 3 = 2 - 3 = -1, mRightScore[23] = 0.  Outside the window: left 10..19 score 8, right 0..19 score 8.  "
        "One group of one pair (no UMI).  LEFT side (the read at 500): template = the only read, it votes alone.  Column 3: bin A holds count 1, score -1, quality sum 7; every "
        "other bin 0 / 0.  Top (group.cpp:395-403): bin 0 takes it with 0 > -inf, every later EMPTY bin ties at 0 with quals 0 >= 0 and takes over, A's -1 never does: topBase = "
        "15, topNum 0, topQual 0.  Second (:406-417): likewise the last empty bin but 15: secBase 14, secNum 0.  secNum == 0 but topScore 0 < baseScoreReq 6: needToCheckRef, "
        "there is no reference: nothing changes topBase.  outBase A != 15: the template's base is overwritten by code 15 = N, quality topQual = 0 (:497-526); no reference base: "
        "mismatchInc stays 0 and NM (1) is not touched.  Every other left column: one bin with score 12 / 10 / 8 >= 6 and top quality 37 >= 20, secNum 0: quality 37 kept (:421-428).  "
        "RIGHT side (the read at 480, the only right read: left-read mode): column 23: T with score 0 and the rewritten quality 0: ALL sixteen bins tie at score 0 / quals 0, the "
        "`>=` hands the top on to bin 15, second = bin 14, both absent: as on the left, N with quality 0 is written.  Column 25: T, score 10, top quality 10: secNum 0 and 10 >= 6 but "
        "topQual 10 < moderateQuality 20: needToCheckRef (no reference), topQual <= lowQuality: again needToCheckRef; the base is its own, outqual = 10: unchanged.  All other "
        "columns: quality 30 >= 20 kept.  Both records leave with FR 1; the mismatching base of a lone pair is masked to N / 0 on BOTH mates.  "
        "Stats: pre 5 reads (20 + 20 + 20 + 30 + 20 bases), 2 unmapped (40 bases), NM: 1 on x's reverse read; 1 cluster (one group), 1 molecule of 1 read, PE.  post: 2 records "
        "(50 bases, NM 1 once), 1 cluster, 1 SSCS, outputPair counts 1 molecule."),
    params=dict(flush_period=3), contigs=C0,
    records=[unmapped("u1"), rec("x", 147, 500, "20M", 480, -40, R1, Q(37, 20), nm=1), unmapped("u2"), rec("x", 99, 480, "30M", 500, 40, R2, q2),
             rec("y", 99, 2000, "20M", 2300, 320, S[:20], Q(37, 20))],
    expected_status=0,
    expected=[out("x", 147, 500, "20M", R1[:3] + "N" + R1[4:], Q(37, 3) + [0] + Q(37, 16), 1, 1),
              out("x", 99, 480, "30M", R2[:23] + "N" + R2[24:], Q(30, 23) + [0, 30, 10] + Q(30, 4), 0, 1)],
    # the order of the output set is not checked for this stream: it is NOT sorted (that is its point), the engine's output order (and the oracle's) is defined for streams
    # whose mapped reads ascend, and the reference itself writes such a stream "unordered" (gencore.cpp:86-103: the watermark drain of :113-143 is out of scope)
    skip_order_check="the stream is unsorted behind an unmapped read",
    expected_stats=dict(pre=stats(5, 110, 1, 0, 1, {1: 1}, unmapped=(2, 40), mism=(1, 1)),
                        post=stats(2, 50, 1, 0, 1, {1: 1}, sscs=1, mism=(1, 1)))))

# ------------------------------------------------------------------------------------------------ 2: odd-shaped duplex UMIs
LA, RA = "ACGTTGCAAGCTTCGATGCA", "TTGCAAGCTTCGATGCAAGC"
names = dict(a="a:UMI__AAAA_CCCC", b="b:UMI_CCCC_AAAA", c="c:UMI_GGGG__TTTT", d="d:UMI_TTTT_GGGG", e="e:UMI_AC_", f="f:UMI__AC")
order = "abcdef"
write(dict(
    name="odd_shaped_duplex_umis_go_through_split",
    cites=["src/bamutil.cpp:40-63", "src/util.h:59-88", "src/cluster.cpp:246-258", "src/cluster.cpp:55-100", "src/cluster.cpp:116-168", "src/cluster.cpp:183-185",
           "src/gencore.cpp:145-160", "src/gencore.cpp:409"],
    derivation=(
        "Prefix UMI.  getUMI in prefix mode (bamutil.cpp:40-63) starts two characters behind the last of the characters U / M / I of the name and takes the whole run of "
        "[ATCG_]: the six pairs of the one cluster (0, 100, 419) carry the UMIs a = '_AAAA_CCCC', b = 'CCCC_AAAA', c = 'GGGG__TTTT', d = 'TTTT_GGGG', e = 'AC_', f = '_AC'.  All reads "
        "of a side are identical (quality 37, no reference): every consensus is its input.  End of file: finishConsensus with unproperReadsUmiDiffThreshold 0 (gencore.cpp:409): "
        "six groups of one pair, in std::map order of the UMIs ('_' sorts behind the letters): e, b, c, d, a, f.  Cluster::isDuplex (cluster.cpp:246-258) tokenises with util.h's "
        "split (util.h:59-88): LEADING separators are skipped (find_first_not_of), a doubled separator yields an EMPTY token, a trailing separator yields an empty LAST token: "
        "a -> [AAAA, CCCC], b -> [CCCC, AAAA], c -> [GGGG, '', TTTT] (three tokens), d -> [TTTT, GGGG], e -> [AC, ''] (two tokens, the second empty), f -> [AC] (one).  The duplex loop "
        "(cluster.cpp:116-168) pops from the back.  p1 = f: one token, never a duplex: single strand, kept (clusterSizeReq 1): SSCS.  p1 = a: the rest is e, b, c, d in this order; "
        "e: a's first token AAAA is not e's second token '': no; b: AAAA == AAAA and CCCC == CCCC: DUPLEX although a's UMI string is one character longer than b's -- "
        "duplexMerge of identical reads: diff 0 <= 2, 1 + 1 >= 1: a is written as DCS with FR 1 (its own read) and RR 1 (b's), b is erased.  p1 = d: the rest is e, c; c has three "
        "tokens: no partner: SSCS.  p1 = c: three tokens: SSCS.  p1 = e: nothing left (and [AC, ''] could only pair with a UMI whose FIRST token is empty, which split never "
        "produces): SSCS.  Five pairs leave: a with FR 1 / RR 1, c, d, e, f with FR 1.  Stats: pre 12 reads x 20 bases; 1 cluster with several molecules; addMolecule: f 1, a + b 2, "
        "d 1, c 1, e 1 (all PE).  post: 10 records; 1 cluster with several molecules (cluster.cpp:183-185); 4 SSCS + 1 DCS; outputPair counts 5 molecules of 1."),
    params=dict(umi_prefix="UMI"), contigs=C0,
    records=[rec(names[k], 99, 100, "20M", 400, 320, LA, Q(37, 20)) for k in order] + [rec(names[k], 147, 400, "20M", 100, -320, RA, Q(37, 20)) for k in order],
    expected_status=0,
    expected=[out(names[k], 99, 100, "20M", LA, Q(37, 20), 0, 1, 1 if k == "a" else -1) for k in "acdef"] +
             [out(names[k], 147, 400, "20M", RA, Q(37, 20), 0, 1, 1 if k == "a" else -1) for k in "acdef"],
    expected_stats=dict(pre=stats(12, 240, 1, 1, 5, {1: 4, 2: 1}), post=stats(10, 200, 1, 1, 5, {1: 5}, sscs=4, dcs=1))))

# ------------------------------------------------------------------------------------------------ 3: Stats words of four older vectors (cluster.cpp:101,136,142,158,162,174-185; gencore.cpp:145-146; stats.cpp:101-139)
ADD = {
    # one cluster of two groups; the duplex (y + x) is ONE molecule of 2 reads in preStats; DCS; outputPair counts the pair as a molecule of 1; y's left read carries NM 2
    "duplex_merge_odd_phase": dict(pre=stats(4, 80, 1, 1, 1, {2: 1}, mism=(2, 1)), post=stats(2, 40, 1, 0, 1, {1: 1}, dcs=1, mism=(2, 1))),
    # k: a cluster of one group, one molecule of 1, SSCS; (x, y): two groups, the duplex is counted in preStats BEFORE the mismatch test drops it (cluster.cpp:136 in front of :137); nothing
    # survives of that cluster: no postStats->addCluster (:183); y's left read carries NM 3
    "duplex_mismatch_over_threshold_drops_both": dict(pre=stats(6, 120, 2, 1, 2, {1: 1, 2: 1}, mism=(3, 1)), post=stats(2, 40, 1, 0, 1, {1: 1}, sscs=1)),
    # k: counted in preStats (cluster.cpp:174), dropped by duplexOnly; z: addMolecule(1) then dropped (:158-159); y + x: a molecule of 2, DCS
    "duplex_only_drops_every_single_strand_consensus": dict(pre=stats(8, 160, 2, 1, 3, {1: 2, 2: 1}), post=stats(2, 40, 1, 0, 1, {1: 1}, dcs=1)),
    # solo: a molecule of 1 in preStats, filtered by clusterSizeReq 2; (m1, m2): one group, a molecule of 2, SSCS
    "cluster_size_req_filters_singletons": dict(pre=stats(6, 120, 2, 0, 2, {1: 1, 2: 1}), post=stats(2, 40, 1, 0, 1, {1: 1}, sscs=1)),
}
for name, es in ADD.items():
    path = os.path.join(HERE, name + ".json")
    v = json.load(open(path))
    v["expected_stats"] = es
    for c in ("src/cluster.cpp:101", "src/cluster.cpp:136-185", "src/gencore.cpp:145-146", "src/stats.cpp:101-139"):
        if c not in v["cites"]:
            v["cites"].append(c)
    write(v)
