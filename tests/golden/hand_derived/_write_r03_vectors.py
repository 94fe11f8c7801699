#!/usr/bin/env python
"""Writes the six round-3 hand-derived vectors.  A WRITING AID, not an oracle: the inputs are spelled out here because three of the
files hold hundreds of records; every `expected` block and every derivation was worked out by hand from the cited reference lines
(no oracle or engine run is involved in producing them)."""
import json, os
HERE = os.path.dirname(os.path.abspath(__file__))
Q37 = lambda n: [37] * n
REF4 = "ACGT"


def rec(qname, flag, pos, cigar, mpos, isize, seq, qual, tid=0, mtid=0, nm=0, **kw):
    return dict(qname=qname, flag=flag, tid=tid, pos=pos, cigar=cigar, mtid=mtid, mpos=mpos, isize=isize, seq=seq, qual=qual, nm=nm, **kw)


def out(qname, flag, pos, cigar, seq, qual, nm, fr, rr=-1, tid=0):
    return dict(qname=qname, flag=flag, tid=tid, pos=pos, cigar=cigar, seq=seq, qual=qual, nm=nm, fr=fr, rr=rr)


def write(v):
    with open(os.path.join(HERE, v["name"] + ".json"), "w") as f:
        json.dump(v, f, indent=1)
        f.write("\n")


# ------------------------------------------------------------------------------------------------ 1: > 1000 pairs
S20 = "ACGTTGCAAGCTTCGATGCA"
records = [rec("a0000", 99, 100, "19M", 400, 420, S20[:19], Q37(19))]
records.append(dict(rec("b{i:04d}", 99, 100, "20M", 400, 420, S20, Q37(20)), repeat=600))
records.append(dict(rec("c{i:04d}", 99, 100, "18M", 400, 420, S20[:18], Q37(18)), repeat=400))
records.append(rec("a0000", 147, 400, "120M", 100, -420, "A" * 120, Q37(120)))
for k in range(1, 111):
    records.append(rec("b%04d" % (k - 1), 147, 400, "%dS%dM" % (k, 120 - k), 100, -420, "A" * 120, Q37(120)))
records.append(dict(rec("b{i:04d}", 147, 400, "120M", 100, -420, "A" * 120, Q37(120)), repeat=490, repeat_from=110))
records.append(dict(rec("c{i:04d}", 147, 400, "120M", 100, -420, "A" * 120, Q37(120)), repeat=400))
write(dict(
    name="deep_group_early_break_and_low_complexity_skip",
    cites=["src/group.cpp:142-175", "src/group.cpp:196-233", "src/group.cpp:235-266", "src/group.cpp:287-313", "src/group.cpp:104-106,122-131", "src/pair.cpp:57-61"],
    derivation=(
        "1001 proper pairs without UMI, all with the cluster key (0, 100, 519): one cluster, one group, taken by finishConsensus (2002 reads < 10000). "
        "mPairs.size() = 1001 > skipLowComplexityClusterThreshold (1000), so both special rules of consensusMergeBam apply. "
        "LEFT side: the left reads carry three CIGAR strings (19M, 20M, 18M): 3 is not > 100.1, no skip (group.cpp:161). containedBy in qname order "
        "(allPairs = map order: a0000, b0000..b0599, c0000..c0399): read 0 (a0000, 19M) is part of every 20M read (one op, 19 <= 20, last op may be shorter: "
        "bamutil.cpp:204-255) and of no 18M read: containedBy = 1 + 600 = 601 >= 1001/2 = 500, and the loop BREAKS after i = 0 (group.cpp:231-232): every other "
        "entry of containedByList stays 0. Without the break the 18M reads would score 1 + 399 + 1 + 600 = 1001 and c0000 would be the template; with it the "
        "maximum is entry 0, 601 is not < 400.4, and the template is a0000's 19-base read. Voters (group.cpp:287-313): the template and the 600 reads it is part "
        "of (the 20M ones). All voters agree on every column with quality 37 and score 8 (no mate overlap: the mates are 300 bases apart): secNum == 0, "
        "topScore = 4808 >= 6, topQual 37 >= 20: the column keeps its base, quality 37 (group.cpp:421-428). Nothing changes, mismatchInc = 0, NM untouched. "
        "RIGHT side: 111 distinct CIGAR strings (120M and kS(120-k)M for k = 1..110) > 100.1, and the first right read in map order (a0000's) is 120 x 'A': "
        "diffNeighbor = 0 < 60, so consensusMergeBam returns NULL (group.cpp:161-174): the result Pair has no right read. "
        "mMergeReads = 1001 (group.cpp:105) >= clusterSizeReq; FR is the low byte of min(1001, 65535) = 0x03E9 -> 0xE9 = 233 (pair.cpp:57-61). "
        "One record is written: a0000's left read, unchanged, with FR:C:233."),
    params={}, contigs=[dict(name="c0", length=100000)], records=records, expected_status=0,
    expected=[out("a0000", 99, 100, "19M", S20[:19], Q37(19), 0, 233)]))

# ------------------------------------------------------------------------------------------------ 2: quals >= 128 and `char refBaseQual`
ref100 = (REF4 * 5)                                         # contig = "ACGT" x 250: the 20 bases at 100 and at 200
qa = Q37(20); qa[4] = 200
qb = Q37(20); qb[4] = 30
sc = list(ref100); sc[4] = "C"; sc = "".join(sc)
qexp = Q37(20); qexp[4] = 30
write(dict(
    name="ref_base_qual_is_a_signed_char",
    cites=["src/group.cpp:394-417", "src/group.cpp:442-457", "src/group.cpp:470-501", "src/pair.cpp:77-86"],
    derivation=(
        "Three pairs a, b, c (no UMI) in one cluster (0, 100, 219), one group; all left reads 20M at 100, identical CIGARs: containedBy = 3 each, equal lengths, "
        "template = a (first in qname order), voters in the order a, b, c (group.cpp:287-313). Column 4 (reference base 'A' at 104; the contig is ACGT repeated, the "
        "template's isize = 120 != 0 so the reference is consulted, group.cpp:363): a has 'A' with quality 200, b has 'A' with quality 30, c has 'C' with quality 37. "
        "qual2score takes a uint8_t: 200 >= 30 scores 8 (pair.cpp:77-86). Bins: A count 2, score 16, quality sum 230, topQuals (uint8_t) = 200; C count 1, score 8, quality 37. "
        "top = A (topNum 2, topQual 200), second = C with secNum 1 and quals[C] = 37 > lowQuality: 'high quality secondary', topNum 2 < 3 -> needToCheckRef (group.cpp:452-456). "
        "Reference integration (group.cpp:470-501) keeps the best ref-consistent quality in `char refBaseQual`: voter a: 200 > 0, refBaseQual = (char)200 = -56 (char is signed "
        "on x86), and 200 >= highQuality sets topBase = ref; voter b: 30 > -56 is TRUE, refBaseQual = 30 -- the smaller quality overwrites the larger one; c is not ref-consistent. "
        "topBase == ref, so topQual = refBaseQual = 30. The template's base is already 'A': no base change, diff 0, mismatchInc 0. outqual[4] = 30, where a true maximum would "
        "give 200. Every other column is unanimous (3 x 8 = 24 >= 6, quality 37). The right reads (20M at 200) are identical: unchanged. FR = 3."),
    params={}, contigs=[dict(name="c0", length=1000, sequence=dict(repeat=REF4, times=250))],
    records=[rec("a", 99, 100, "20M", 200, 120, ref100, qa), rec("b", 99, 100, "20M", 200, 120, ref100, qb), rec("c", 99, 100, "20M", 200, 120, sc, Q37(20)),
             rec("a", 147, 200, "20M", 100, -120, ref100, Q37(20)), rec("b", 147, 200, "20M", 100, -120, ref100, Q37(20)), rec("c", 147, 200, "20M", 100, -120, ref100, Q37(20))],
    expected_status=0,
    expected=[out("a", 99, 100, "20M", ref100, qexp, 0, 3), out("a", 147, 200, "20M", ref100, Q37(20), 0, 3)]))

# ------------------------------------------------------------------------------------------------ 3: NM whose aux type is not 'C'
recs3 = []
for tag, base, typ in (("x", 100, "S"), ("y", 400, "C")):
    recs3 += [rec(tag + "a", 99, base, "20M", base + 100, 120, sc, Q37(20), nm=1, nm_type=typ), rec(tag + "b", 99, base, "20M", base + 100, 120, ref100, Q37(20)),
              rec(tag + "c", 99, base, "20M", base + 100, 120, ref100, Q37(20))]
    recs3 += [rec(tag + k, 147, base + 100, "20M", base, -120, ref100, Q37(20)) for k in "abc"]
write(dict(
    name="nm_is_patched_only_when_stored_as_type_C",
    cites=["src/group.cpp:470-501", "src/group.cpp:503-524", "src/group.cpp:528-573"],
    derivation=(
        "Two clusters of three pairs each (no UMI), x at (0, 100, 219) and y at (0, 400, 519), both built the same way: the template (xa / ya, first in qname order, identical "
        "CIGARs) carries 'C' at column 4 where the reference (ACGT repeated; 104 and 404 are 'A') and the two other voters have 'A', all qualities 37. Column 4: A count 2 score 16, "
        "C count 1 score 8; second base single with quality 37 > lowQuality and topNum 2 < 3 -> needToCheckRef (group.cpp:452-456); voters b and c are ref-consistent with quality "
        "37 >= highQuality: topBase = ref 'A', topQual = refBaseQual = 37 (group.cpp:470-501). outBase 'C' != 'A': the template's base becomes 'A', diff = 1, and because "
        "topBase == ref (outBase was not) mismatchInc = -1 (group.cpp:503-524). mismatchInc != 0 enters the NM block (group.cpp:528-573): valNM = 1, newValNM = 0, not > 5, so the "
        "byte is rewritten ONLY `if(typeNM == 'C' && ...)`. xa stores NM as type 'S' (uint16): it keeps NM = 1 although the consensus now matches the reference; ya stores it as "
        "type 'C': NM becomes 0. All other columns and the right reads are unanimous and unchanged. FR = 3 for every record."),
    params={}, contigs=[dict(name="c0", length=1000, sequence=dict(repeat=REF4, times=250))], records=sorted(recs3, key=lambda r: r["pos"]), expected_status=0,
    expected=[out("xa", 99, 100, "20M", ref100, Q37(20), 1, 3), out("xa", 147, 200, "20M", ref100, Q37(20), 0, 3),
              out("ya", 99, 400, "20M", ref100, Q37(20), 0, 3), out("ya", 147, 500, "20M", ref100, Q37(20), 0, 3)]))

# ------------------------------------------------------------------------------------------------ 4: cross-contig key under a periodic flush
SEQ = "ACGTTGCAAGCTTCGATGCA"
x = lambda name, pos, mpos: rec(name, 65, pos, "20M", mpos, 0, SEQ, Q37(20), mtid=1)
write(dict(
    name="cross_contig_key_is_taken_by_every_periodic_flush",
    cites=["src/gencore.cpp:299-313", "src/gencore.cpp:319-322", "src/gencore.cpp:333-362", "src/gencore.cpp:392-434", "src/cluster.cpp:55-102", "src/group.cpp:68-131", "src/options.cpp:12-13"],
    derivation=(
        "Contigs c0 and c1, 1000 bases each; flush period 4 (the literal 10000 of gencore.cpp:321). Six reads on c0 whose mates lie on c1: the key's right is "
        "-target_len[0] * (mtid + 1) + mpos (gencore.cpp:311): -1500 for mpos 500, -1499 for mpos 501, -1498 for mpos 502 -- NEGATIVE, so `iter3->first >= b->core.pos` "
        "(gencore.cpp:352) never stops the walk: a cross-contig cluster is taken by the first flush whose read lies right of its left position. "
        "Ticks: q1, q2 (pos 100, mpos 500: one cluster (0,100,-1500)) = 1, 2; s3 (pos 150, mpos 501) = 3; s4 (pos 160, mpos 502) = 4 -> 4 % 4 == 0: the walk runs with pos = 160 and "
        "takes every cluster with left < 160 (gencore.cpp:345-349): (100,-1500) and (150,-1499), with properReadsUmiDiffThreshold = 1 and crossContig = true (right < 0, :355). "
        "s4's own cluster has left 160 >= 160: it stays. q5, q6 (pos 300, mpos 500) = ticks 5, 6; no further event; finishConsensus takes (160,-1498) and (300,-1500) with "
        "unproperReadsUmiDiffThreshold = 0 (gencore.cpp:409, options.cpp:13). "
        "Cluster (100,-1500) under threshold 1: UMIs AAAAAAAA (q1) and AAAAAAAT (q2), one pair each (every read is its Pair's left; there are no right reads); the top UMI is "
        "AAAAAAAA (equal counts: first in map order) and absorbs AAAAAAAT (distance 1): ONE group of two pairs. consensusMerge(crossContig): nameToCopy = the shortest, then "
        "smallest left name = q1's (group.cpp:80-99); left side: identical 20M reads, template q1, both vote; isize == 0 -> no reference; a Pair without a right read scores the "
        "constant 6 (pair.cpp:89-105): 12 >= 6, quality 37: unchanged. Right side: no reads, containedByList = {0, 0}, 0 < 0.8 -> NULL. mMergeReads = 2: q1's record leaves with FR:C:2. "
        "Cluster (150,-1499): one Pair without right read: returned untouched (group.cpp:73-77), FR 1. The same for s4 at the end of the file. "
        "Cluster (300,-1500) under threshold 0: AAAAAAAA absorbs only itself: TWO groups, two untouched singletons, FR 1 each -- the same two UMIs that merged in front of the flush."),
    params=dict(umi_prefix="UMI", flush_period=4), contigs=[dict(name="c0", length=1000), dict(name="c1", length=1000)],
    records=[x("q1:UMI_AAAAAAAA", 100, 500), x("q2:UMI_AAAAAAAT", 100, 500), x("s3:UMI_CCCCCCCC", 150, 501), x("s4:UMI_GGGGGGGG", 160, 502),
             x("q5:UMI_AAAAAAAA", 300, 500), x("q6:UMI_AAAAAAAT", 300, 500)],
    expected_status=0,
    expected=[out("q1:UMI_AAAAAAAA", 65, 100, "20M", SEQ, Q37(20), 0, 2), out("s3:UMI_CCCCCCCC", 65, 150, "20M", SEQ, Q37(20), 0, 1), out("s4:UMI_GGGGGGGG", 65, 160, "20M", SEQ, Q37(20), 0, 1),
              out("q5:UMI_AAAAAAAA", 65, 300, "20M", SEQ, Q37(20), 0, 1), out("q6:UMI_AAAAAAAT", 65, 300, "20M", SEQ, Q37(20), 0, 1)]))

# ------------------------------------------------------------------------------------------------ 5: several rights on one left
recs5, exp5 = [], []
for k, (isz, rpos) in enumerate(((150, 230), (250, 330), (350, 430)), start=1):
    for nm_, umi in (("a", "AAAAAAAA"), ("b", "AAAAAAAT")):
        name = "k%d%s:UMI_%s" % (k, nm_, umi)
        recs5.append(rec(name, 99, 100, "20M", rpos, isz, SEQ, Q37(20)))
        recs5.append(rec(name, 147, rpos, "20M", 100, -isz, SEQ, Q37(20)))
        if k > 1:
            exp5 += [out(name, 99, 100, "20M", SEQ, Q37(20), 0, 1), out(name, 147, rpos, "20M", SEQ, Q37(20), 0, 1)]
exp5 += [out("k1a:UMI_AAAAAAAA", 99, 100, "20M", SEQ, Q37(20), 0, 2), out("k1a:UMI_AAAAAAAA", 147, 230, "20M", SEQ, Q37(20), 0, 2)]
write(dict(
    name="flush_walk_stops_at_the_first_right_not_left_of_the_read",
    cites=["src/gencore.cpp:299-304", "src/gencore.cpp:319-322", "src/gencore.cpp:350-362", "src/gencore.cpp:409", "src/cluster.cpp:55-102", "src/options.cpp:12-13"],
    derivation=(
        "Three clusters share tid 0 and left 100 and differ in right = left + |isize| - 1: 249, 349, 449 (isize 150 / 250 / 350; right mates at 230 / 330 / 430). Each holds two "
        "pairs whose UMIs differ in one base (AAAAAAAA / AAAAAAAT). Flush period 9. Stream order: the six left reads at 100 (ticks 1-6), k1's right reads at 230 (7, 8), k2's first "
        "right read at 330 = tick 9 -> the walk runs with pos = 330 (gencore.cpp:319-322). Contig 0, left 100 < 330, then the rights of that left in ascending order "
        "(map<long, Cluster*>): 249 < 330 is taken (properReadsUmiDiffThreshold = 1); 349 >= 330 BREAKS the loop (gencore.cpp:352-354): 349 and 449 stay, although the read that "
        "triggered the walk was just added to 349's cluster. No other event (12 ticks). "
        "k1 under threshold 1: AAAAAAAA absorbs AAAAAAAT: one group of two pairs; identical reads, template k1a on both sides, nothing changes; the names are equal, FR = 2: two records. "
        "k2 and k3 are taken by finishConsensus with threshold 0 (gencore.cpp:409, options.cpp:13): two groups each, i.e. singleton pairs WITH right reads -- not the "
        "untouched case of group.cpp:73-77: they vote alone (one voter, score 8 >= 6, quality 37: unchanged) -- written as they came with FR = 1: eight records."),
    params=dict(umi_prefix="UMI", flush_period=9), contigs=[dict(name="c0", length=100000)], records=sorted(recs5, key=lambda r: r["pos"]), expected_status=0, expected=exp5))

# ------------------------------------------------------------------------------------------------ 6: MI:Z
recs6 = []
for name, mi in (("a:UMI_AAAAAAAA", None), ("b:UMI_CCCCCCCC", "zz:UMI_AAAAAAAA"), ("c:UMI_AAAAAAAA", "UMI_GGGGGGGG")):
    for flag, pos, mpos, isz in ((99, 100, 200, 120), (147, 200, 100, -120)):
        r = rec(name, flag, pos, "20M", mpos, isz, SEQ, Q37(20))
        if mi:
            r["mi"] = mi
        recs6.append(r)
write(dict(
    name="mi_tag_overrides_the_read_name",
    cites=["src/bamutil.cpp:23-38", "src/bamutil.cpp:40-63", "src/pair.cpp:188-216", "src/cluster.cpp:55-102", "src/cluster.cpp:116-188"],
    derivation=(
        "BamUtil::getUMI(b, prefix) looks for an MI tag first and, if it is there, runs the SAME string parser on the tag's value instead of the name (bamutil.cpp:23-38); with "
        "prefix \"UMI\" that parser takes the run of [ATCG_] two characters behind the last 'U', 'M' or 'I' (bamutil.cpp:45-63). Three pairs in one cluster (0, 100, 219): "
        "a: no MI tag, name ...UMI_AAAAAAAA -> AAAAAAAA. b: name says CCCCCCCC, both mates carry MI:Z:zz:UMI_AAAAAAAA -> AAAAAAAA. c: name says AAAAAAAA, both mates carry "
        "MI:Z:UMI_GGGGGGGG -> GGGGGGGG. (setRight's check, pair.cpp:201-212, compares like with like: both mates of a pair carry the same tag.) "
        "Six reads: finishConsensus, threshold 0. umiCount: AAAAAAAA 2 (a, b), GGGGGGGG 1 (c): the first group is {a, b}, the second {c}. Had the names been used, a and c "
        "would have met and b stayed alone. Group {a, b}: identical reads, template a on both sides, unchanged, FR = 2. Group {c}: a singleton WITH a right read votes alone, "
        "unchanged, FR = 1. The duplex stage finds no partner (no '_' in the UMIs, cluster.cpp:246-258): no RR."),
    params=dict(umi_prefix="UMI"), contigs=[dict(name="c0", length=100000)], records=sorted(recs6, key=lambda r: r["pos"]), expected_status=0,
    expected=[out("a:UMI_AAAAAAAA", 99, 100, "20M", SEQ, Q37(20), 0, 2), out("a:UMI_AAAAAAAA", 147, 200, "20M", SEQ, Q37(20), 0, 2),
              out("c:UMI_AAAAAAAA", 99, 100, "20M", SEQ, Q37(20), 0, 1), out("c:UMI_AAAAAAAA", 147, 200, "20M", SEQ, Q37(20), 0, 1)]))
print("written")
