#!/usr/bin/env python
"""Writes the round-4 hand-derived vectors (the three rules VERDICT r3 named as still without a vector).  A WRITING AID, not an oracle:
every `expected` block and every derivation was worked out by hand from the cited reference lines; no oracle or engine run is involved."""
import json, os
HERE = os.path.dirname(os.path.abspath(__file__))
Q37 = lambda n: [37] * n


def rec(qname, flag, pos, cigar, mpos, isize, seq, qual, tid=0, mtid=0, nm=0, **kw):
    return dict(qname=qname, flag=flag, tid=tid, pos=pos, cigar=cigar, mtid=mtid, mpos=mpos, isize=isize, seq=seq, qual=qual, nm=nm, **kw)


def out(qname, flag, pos, cigar, seq, qual, nm, fr, rr=-1, tid=0):
    return dict(qname=qname, flag=flag, tid=tid, pos=pos, cigar=cigar, seq=seq, qual=qual, nm=nm, fr=fr, rr=rr)


def write(v):
    with open(os.path.join(HERE, v["name"] + ".json"), "w") as f:
        json.dump(v, f, indent=1)
        f.write("\n")


C0 = [dict(name="c0", length=100000)]
L20, R20 = "ACGTACGTACGTACGTACGT", "TTGCAAGCTTCGATGCAAGC"

# ------------------------------------------------------------------------------------------------ 1: a template without CIGAR
qa = Q37(7) + [11] + Q37(2) + [25, 11]
qb = Q37(7) + [25] + Q37(2)
qc = Q37(7) + [11] + Q37(2)
write(dict(
    name="template_without_cigar_votes_the_shortest_voter_length",
    cites=["src/group.cpp:354-360", "src/bamutil.cpp:204-211", "src/group.cpp:196-266", "src/group.cpp:287-313", "src/pair.cpp:88-107", "src/bamutil.cpp:316-336", "src/group.cpp:394-428,442-467,503-525"],
    derivation=(
        "Three pairs a, b, c without UMI in one cluster (0, 100, 419), one group, taken by finishConsensus.  Left reads: a is mapped with CIGAR * "
        "(n_cigar 0), 12 bases; b and c are 10M, 10 bases.  consensusMergeBam(left): isPartOf(a, x) is true for every x (cigarNumPart 0: the loop of "
        "bamutil.cpp:213 never runs), isPartOf(b, a) is false (cigarNumWhole 0 < 1, :210-211), isPartOf(b, c) = isPartOf(c, b) = true: containedBy = "
        "3, 2, 2 -> template a (3 >= 0.4 x 3), voters a, b, c (isPartOf(a, .) again).  makeConsensus: out->core.n_cigar == 0, so len = min(12, 10, 10) "
        "= 10 (group.cpp:354-360): columns 10 and 11 of the template are never looked at and keep base and quality (G/25, T/11) -- a loop to "
        "l_qseq = 12 would read b and c past their end.  Scores: pair a has a left read without M block (getMOffsetAndLen: MLen 0), so both of its "
        "reads keep the memset constant 6 (pair.cpp:89-107); b and c score qual2score(q) (right reads start 300 bases later: cmpLen < 0).  "
        "Columns 0-4, 6, 8, 9: A everywhere, secNum 0, topScore 6 + 8 + 8 = 22 >= 6, topQual 37 -> quality 37.  Column 7: A everywhere, "
        "qualities 11 / 25 / 11, scores 6 + 6 + 2 = 14 >= 6, topQual 25 >= 20 -> kept, quality 25 (the template's own 11 is replaced).  Column 5: "
        "a says A (score 6, quality 37), b and c say C (16, 74): top C, topNum 2, second A with secNum 1 and quals[A] = 37 > lowQuality: "
        "'high quality secondary', topNum < 3 -> needToCheckRef, but there is no reference (refbase 0): C is written into the template, "
        "quality 37, diff 1, mismatchInc 0 (NM stays).  Right reads are identical 20M: template = first in qname order (a), unchanged.  FR = 3."),
    params={}, contigs=C0,
    records=[rec("a", 99, 100, "*", 400, 320, "AAAAAAAAAAGT", qa), rec("b", 99, 100, "10M", 400, 320, "AAAAACAAAA", qb), rec("c", 99, 100, "10M", 400, 320, "AAAAACAAAA", qc),
             rec("a", 147, 400, "20M", 100, -320, R20, Q37(20)), rec("b", 147, 400, "20M", 100, -320, R20, Q37(20)), rec("c", 147, 400, "20M", 100, -320, R20, Q37(20))],
    expected_status=0,
    expected=[out("a", 99, 100, "*", "AAAAACAAAAGT", Q37(7) + [25] + Q37(2) + [25, 11], 0, 3), out("a", 147, 400, "20M", R20, Q37(20), 0, 3)]))

# ------------------------------------------------------------------------------------------------ 2a: a third read with another UMI takes over
write(dict(
    name="third_read_with_another_umi_replaces_right_and_names_the_pair",
    cites=["src/cluster.cpp:260-272", "src/pair.cpp:196-216", "src/bamutil.cpp:23-38,45-63", "src/cluster.cpp:55-100", "src/gencore.cpp:409", "src/group.cpp:122-131"],
    derivation=(
        "Prefix UMI.  One cluster (0, 100, 419).  Name a occurs three times: the left read and the first right read carry no MI tag and a name "
        "without any of the characters U, M, I (find_last_of -> npos: UMI \"\", bamutil.cpp:45-48); the second right read (the stream's 5th record) "
        "carries MI:Z:UMI_CCCC.  Cluster::addRead: setLeft -> mUMI \"\"; setRight(first right) -> mUMI empty, so no check, mUMI = \"\"; setRight(third) "
        "destroys the first right read (pair.cpp:197-199) and, mUMI still being empty, sets mUMI = CCCC (pair.cpp:201-214): the pair now carries the "
        "third read's UMI.  Pair b carries CCCC on both reads, pair c CCCA.  End of file: finishConsensus groups with unproperReadsUmiDiffThreshold 0 "
        "(gencore.cpp:409): umiCount {CCCA: 1, CCCC: 2}; top CCCC takes a and b (umiDiff 0), then CCCA takes c.  Had pair a kept the empty UMI of its "
        "second read it would have formed a group of its own and all three pairs would come out with FR 1.  Group {a, b}: left reads identical, "
        "template = first in qname order = a's left; right reads = a's THIRD record and b's right, identical (...GC) -> template a's third record; "
        "the record ending in ...GA is gone.  The merged pair: setLeft(a's left) -> \"\", setRight(third) -> CCCC, no mismatch (mUMI empty).  Duplex "
        "stage (hasUMI): CCCA and CCCC have no '_' -> never duplex; both groups are written as SSCS: FR 2 for a, FR 1 for c; b is not written."),
    params=dict(umi_prefix="UMI"), contigs=C0,
    records=[rec("a", 99, 100, "20M", 400, 320, L20, Q37(20)), rec("b", 99, 100, "20M", 400, 320, L20, Q37(20), mi="UMI_CCCC"), rec("c", 99, 100, "20M", 400, 320, L20, Q37(20), mi="UMI_CCCA"),
             rec("a", 147, 400, "20M", 100, -320, R20[:19] + "A", Q37(20), nm=1), rec("a", 147, 400, "20M", 100, -320, R20, Q37(20), mi="UMI_CCCC"),
             rec("b", 147, 400, "20M", 100, -320, R20, Q37(20), mi="UMI_CCCC"), rec("c", 147, 400, "20M", 100, -320, R20, Q37(20), mi="UMI_CCCA")],
    expected_status=0,
    expected=[out("a", 99, 100, "20M", L20, Q37(20), 0, 2), out("a", 147, 400, "20M", R20, Q37(20), 0, 2),
              out("c", 99, 100, "20M", L20, Q37(20), 0, 1), out("c", 147, 400, "20M", R20, Q37(20), 0, 1)]))

# ------------------------------------------------------------------------------------------------ 2b: ... and is fatal when the pair already has a UMI
write(dict(
    name="third_read_with_another_umi_is_fatal_when_the_pair_has_one",
    cites=["src/pair.cpp:196-212", "src/cluster.cpp:260-272", "src/util.h:250"],
    derivation=(
        "Prefix UMI.  Name a three times: left MI:Z:UMI_AAAA, first right MI:Z:UMI_AAAA (setRight: equal, fine), a second right read with "
        "MI:Z:UMI_AAAT: mUMI = AAAA is not empty and differs -> error_exit(\"The UMI of a read pair should be identical ...\") (pair.cpp:201-212).  "
        "The engine reports GCE_ERR_UMI_MISMATCH (-11)."),
    params=dict(umi_prefix="UMI"), contigs=C0,
    records=[rec("a", 99, 100, "20M", 400, 320, L20, Q37(20), mi="UMI_AAAA"), rec("a", 147, 400, "20M", 100, -320, R20, Q37(20), mi="UMI_AAAA"),
             rec("a", 147, 400, "20M", 100, -320, R20, Q37(20), mi="UMI_AAAT")],
    expected_status=-11, expected=[]))

# ------------------------------------------------------------------------------------------------ 3: duplexOnly
write(dict(
    name="duplex_only_drops_every_single_strand_consensus",
    cites=["src/cluster.cpp:116-152", "src/cluster.cpp:155-167", "src/cluster.cpp:169-181", "src/cluster.cpp:246-258", "src/pair.cpp:38-68"],
    derivation=(
        "Prefix UMI, duplexOnly.  Pair k (no UMI) is a cluster of its own, (0, 50, 369): hasUMI is false, the plain branch (cluster.cpp:169-181) "
        "keeps a pair only if !duplexOnly -> deleted, nothing written.  Cluster (0, 100, 419): x = AAAA_CCCC, y = CCCC_AAAA, z = GGGG_TTTT, threshold 0 "
        "at the end of the file: three groups of one pair in std::map order of their UMIs (x, y, z), each a depth-1 consensus that changes nothing "
        "(quality 37, no reference).  The duplex loop pops from the back: p1 = z finds no partner (isDuplex needs the two halves swapped) -> "
        "single strand, and `!duplexOnly && ...` (cluster.cpp:159) is false: deleted.  p1 = y: isDuplex(CCCC_AAAA, AAAA_CCCC) is true, "
        "duplexMerge of identical reads gives diff 0 <= 2, 1 + 1 >= clusterSizeReq 1 -> y is written as DCS with FR = its own 1 read and "
        "RR = x's 1 read (setDuplex(p2->mMergeReads), pair.cpp:38-41,57-66); x is erased and deleted.  Output: y's two records only."),
    params=dict(umi_prefix="UMI", duplex_only=1), contigs=C0,
    records=[rec("k", 99, 50, "20M", 350, 320, L20, Q37(20)),
             rec("x:UMI_AAAA_CCCC", 99, 100, "20M", 400, 320, L20, Q37(20)), rec("y:UMI_CCCC_AAAA", 99, 100, "20M", 400, 320, L20, Q37(20)), rec("z:UMI_GGGG_TTTT", 99, 100, "20M", 400, 320, L20, Q37(20)),
             rec("k", 147, 350, "20M", 50, -320, R20, Q37(20)),
             rec("x:UMI_AAAA_CCCC", 147, 400, "20M", 100, -320, R20, Q37(20)), rec("y:UMI_CCCC_AAAA", 147, 400, "20M", 100, -320, R20, Q37(20)), rec("z:UMI_GGGG_TTTT", 147, 400, "20M", 100, -320, R20, Q37(20))],
    expected_status=0,
    expected=[out("y:UMI_CCCC_AAAA", 99, 100, "20M", L20, Q37(20), 0, 1, 1), out("y:UMI_CCCC_AAAA", 147, 400, "20M", R20, Q37(20), 0, 1, 1)]))
