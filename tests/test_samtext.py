"""SAM text in / out (the reference reads SAM or BAM through sam_open(in, "r") and writes SAM text when the output name
ends in "sam", src/gencore.cpp:164-173).  The library's line functions (gencore_amd/csrc/gce_samtext.hpp) against an independent
pure-Python SAM / BAM writer and reader (tests/pybam.py): field by field, both directions, plus the htslib conventions the consensus
path can see (integer tags in the smallest type -- NM 0..255 as 'C', src/group.cpp:569 --, bin, QUAL '*', unknown RNAME)."""
import gzip
import random

import pytest

import pybam
from gencore_amd import bamio
from gencore_amd.capi import GceError

TARGETS = [("chr1", 100000), ("chr2", 250000000), ("chrUn_x", 5000)]


def records(seed=0, n=400):
    rng = random.Random(seed)
    out = []
    for k in range(n):
        ls = rng.choice([0, 1, 2, 7, 36, 75, 150, 151])
        cig_kind = rng.randrange(5)
        if ls == 0:
            cigar = "*"
        elif cig_kind == 0 or ls < 8:
            cigar = "%dM" % ls
        elif cig_kind == 1:
            cigar = "3S%dM" % (ls - 3)
        elif cig_kind == 2:
            cigar = "%dM2D%dM1I%dM" % (2, 3, ls - 6)
        elif cig_kind == 3:
            cigar = "2H%d=1X%dM70000N1M" % (2, ls - 4)
        else:
            cigar = "*"
        tid = rng.choice([0, 0, 1, 2, -1])
        mtid = rng.choice([tid, tid, 1, -1])
        flag = rng.choice([99, 147, 83, 163, 4, 77, 141, 0, 16, 1024 + 99])
        pos = -1 if tid < 0 else rng.choice([0, 1, 16383, 16384, 131071, 131072, rng.randrange(0, TARGETS[tid][1] - 200)])
        r = dict(qname="r%d:%s" % (k, "x" * rng.randrange(0, 40)), flag=flag, tid=tid, pos=pos, mapq=rng.randrange(0, 256), cigar=cigar,
                 mtid=mtid, mpos=-1 if mtid < 0 else rng.randrange(0, 99000), isize=rng.choice([0, 150, -150, 2 ** 31 - 1, -(2 ** 31 - 1), rng.randrange(-1000, 1000)]),
                 seq="".join(rng.choice("ACGTNRYKM=") for _ in range(ls)), qual=[rng.randrange(0, 94) for _ in range(ls)])
        if tid < 0 or cigar == "*":
            r["flag"] |= 4                       # what htslib's sam_parse1 makes of such a line (a read without contig or CIGAR carries BAM_FUNMAP): the text round trip keeps it
        if ls and rng.random() < 0.1:
            r["qual"] = [0xFF] * ls
        if ls == 1 and r["qual"] == [9]:
            r["qual"] = [10]                     # a one-base read of quality 9 prints QUAL as "*": SAM text cannot tell it from "no qualities" (htslib reads 0xFF too)
        nm = rng.choice([None, 0, 3, 255, 256, 65535, 65536, -1, -128, -129, -32768, -32769, 4000000000])
        if nm is not None:
            r["nm"] = nm
            r["nm_type"] = ("c" if nm >= -128 else "s" if nm >= -32768 else "i") if nm < 0 else ("C" if nm <= 255 else "S" if nm <= 65535 else "I")
        if rng.random() < 0.3:
            r["mi"] = "UMI_%d" % k
        r["aux_pre"] = [("RG", "Z", "grp1")] if rng.random() < 0.5 else []
        r["aux_post"] = rng.choice([[], [("XA", "A", b"Q")], [("XB", "B", ("s", [-3, 7, 300]))], [("XF", "B", ("f", [1.5, -2.0]))], [("XH", "H", "1AE301")],
                                    [("XX", "f", 2.5)], [("ZB", "B", ("C", []))], [("XI", "B", ("I", [0, 4000000000]))]])
        r["bin"] = pybam.expected_bin(r)
        out.append(r)
    return out


def same_record(want, got):
    cig = pybam.parse_cigar(want.get("cigar", "*"))
    assert got["qname"] == want["qname"] and got["flag"] == want["flag"] and got["tid"] == want["tid"] and got["pos"] == want["pos"]
    assert got["mapq"] == want["mapq"] and got["bin"] == want["bin"], (got["bin"], want["bin"], want)
    assert got["cigar"] == list(cig) and got["mtid"] == want["mtid"] and got["mpos"] == want["mpos"] and got["isize"] == want["isize"]
    assert got["seq"] == want["seq"].upper() and got["qual"] == list(want["qual"])
    exp = list(want.get("aux_pre", []))
    if want.get("nm") is not None:
        exp.append(("NM", want["nm_type"], want["nm"]))
    if want.get("mi") is not None:
        exp.append(("MI", "Z", want["mi"]))
    exp += want.get("aux_post", [])
    assert got["aux_order"] == [t for t, _, _ in exp]
    for tag, typ, val in exp:
        gt, gv = got["aux"][tag]
        if typ == "A":
            val = val if isinstance(val, bytes) else val.encode()
        if typ == "B":
            assert gt == "B" and gv[0] == val[0] and list(gv[1]) == list(val[1])
        else:
            assert (gt, gv) == (typ, val), (tag, gt, gv, typ, val)


@pytest.mark.parametrize("threads", [1, 3, 16])
def test_sam_to_bam_field_by_field(built, tmp_path, threads):
    recs = records(1)
    sam, bam = tmp_path / "in.sam", tmp_path / "out.bam"
    text = "@HD\tVN:1.6\tSO:coordinate\n@RG\tID:grp1\tSM:s\n"
    pybam.write_sam(sam, recs, TARGETS, text=text)
    bamio.sam_to_bam(sam, bam, threads=threads, level=1)
    got_text, got_targets, got = pybam.read_bam(bam)
    assert got_targets == TARGETS
    assert got_text == text + "".join("@SQ\tSN:%s\tLN:%d\n" % t for t in TARGETS)
    assert len(got) == len(recs)
    for w, g in zip(recs, got):
        same_record(w, g)


def test_bam_to_sam_line_by_line(built, tmp_path):
    recs = records(2)
    bam, sam = tmp_path / "in.bam", tmp_path / "out.sam"
    pybam.write_bam(bam, recs, TARGETS, text="@HD\tVN:1.6\tSO:coordinate\n")         # no @SQ lines in the text: the writer makes them (sam_hdr_write)
    bamio.bam_to_sam(bam, sam, threads=4)
    text, lines = pybam.read_sam(sam)
    assert text == "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % t for t in TARGETS)
    assert len(lines) == len(recs)
    for r, ln in zip(recs, lines):
        assert ln == pybam.sam_line(r, TARGETS).split("\t")


def test_round_trip_is_the_identity_on_the_record_stream(built, tmp_path):
    recs = records(3, n=1500)
    a, s, b2 = tmp_path / "a.bam", tmp_path / "a.sam", tmp_path / "b.bam"
    pybam.write_bam(a, recs, TARGETS, text="@HD\tVN:1.6\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % t for t in TARGETS))
    bamio.bam_to_sam(a, s)
    bamio.sam_to_bam(s, b2, level=-1)
    assert gzip.decompress(open(a, "rb").read()) == gzip.decompress(open(b2, "rb").read())


def test_sam_quirks(built, tmp_path):
    """CRLF line ends, no line feed at the end of the file, lower-case bases, '=' / '*' / an unknown contig name, empty lines."""
    sam, bam = tmp_path / "q.sam", tmp_path / "q.bam"
    body = ("@HD\tVN:1.6\r\n@SQ\tSN:c1\tLN:1000\r\n@SQ\tLN:2000\tSN:c2\r\n"
            "a\t99\tc1\t11\t60\t4M\t=\t21\t14\tacgt\tIIII\tNM:i:0\r\n"
            "\r\n"
            "b\t0\tnosuch\t5\t0\t*\tc2\t7\t0\t*\t*\r\n"
            "c\t4\t*\t0\t0\t*\t*\t0\t0\tNN\t*")
    open(sam, "w", newline="").write(body)
    bamio.sam_to_bam(sam, bam, threads=2)
    text, targets, got = pybam.read_bam(bam)
    assert targets == [("c1", 1000), ("c2", 2000)]
    assert [g["qname"] for g in got] == ["a", "b", "c"]
    assert got[0]["seq"] == "ACGT" and got[0]["qual"] == [40] * 4 and got[0]["mtid"] == 0 and got[0]["aux"]["NM"] == ("C", 0) and got[0]["bin"] == 4681
    assert got[1]["tid"] == -1 and got[1]["mtid"] == 1 and got[1]["seq"] == "" and got[1]["cigar"] == []
    assert got[1]["flag"] == 4                                # an unknown RNAME: treated as unmapped, BAM_FUNMAP set (sam_parse1)
    assert got[2]["tid"] == -1 and got[2]["pos"] == -1 and got[2]["qual"] == [255, 255] and got[2]["bin"] == 4680


def test_sam_lines_htslib_treats_as_unmapped(built, tmp_path):
    """sam_parse1: "mapped query cannot have zero coordinate; treated as unmapped" (contig dropped, BAM_FUNMAP), "mapped query must have a CIGAR;
    treated as unmapped" (BAM_FUNMAP only: contig and position stay -- the consensus path goes by contig and position, src/gencore.cpp:255)."""
    sam, bam = tmp_path / "u.sam", tmp_path / "u.bam"
    open(sam, "w").write("@SQ\tSN:c1\tLN:1000\n"
                         "z\t99\tc1\t0\t60\t4M\t=\t21\t14\tACGT\tIIII\n"
                         "n\t99\tc1\t11\t60\t*\t=\t21\t14\tACGT\tIIII\n"
                         "m\t99\tc1\t11\t60\t4M\t=\t21\t14\tACGT\tIIII\n")
    bamio.sam_to_bam(sam, bam, threads=1)
    _, _, got = pybam.read_bam(bam)
    assert (got[0]["tid"], got[0]["pos"], got[0]["flag"]) == (-1, -1, 103)
    assert got[0]["mtid"] == -1                    # RNEXT "=" copies the tid as it stands after the reset (sam_parse1 parses RNEXT behind POS)
    assert got[1]["mtid"] == 0 and got[2]["mtid"] == 0
    assert (got[1]["tid"], got[1]["pos"], got[1]["flag"], got[1]["cigar"]) == (0, 10, 103, [])
    assert (got[2]["tid"], got[2]["pos"], got[2]["flag"]) == (0, 10, 99)


@pytest.mark.parametrize("line", ["a\t99\tc1\t11\t60\t4M\t=\t21\t14\tACGT", "a\tx\tc1\t11\t60\t4M\t=\t21\t14\tACGT\tIIII", "a\t99\tc1\t11\t60\t4Q\t=\t21\t14\tACGT\tIIII",
                                  "a\t99\tc1\t11\t60\t4M\t=\t21\t14\tACGT\tIII", "a\t99\tc1\t11\t60\t4M\t=\t21\t14\tACGT\tIIII\tNM:i", "a\t99\tc1\t11\t60\t4M\t=\t21\t14\tACGT\tIIII\tNM:q:1",
                                  "a\t99\tc1\t11\t60\t5M\t=\t21\t14\tACGT\tIIII",             # "CIGAR and query sequence are of different length" (sam_parse1)
                                  "a\t99\tc1\t11\t60\t2S1M\t=\t21\t14\tACGT\tIIII", "a\t99\tc1\t11\t60\t4M\t=\t21\t14\tACGT\tIIII\tXB:B:c,1,200", "a\t99\tc1\t11\t60\t4M\t=\t21\t14\tACGT\tIIII\tXB:B:S,-1"])
def test_malformed_sam_lines_are_refused(built, tmp_path, line):
    sam = tmp_path / "bad.sam"
    open(sam, "w").write("@SQ\tSN:c1\tLN:1000\n" + line + "\n")
    with pytest.raises(GceError):
        bamio.sam_to_bam(sam, tmp_path / "bad.bam")


@pytest.mark.gpu
@pytest.mark.parametrize("workload,n_pairs", [("cfg3", 6000), ("cfg2", 4000)])
def test_sam_end_to_end(built, tmp_path, workload, n_pairs):
    """gce_run_bam takes SAM text and writes SAM text for an output name that ends in "sam" (src/gencore.cpp:164-173): the four
    combinations of BAM / SAM in and out give the same records (compared as text, printed by an independent formatter) and Stats."""
    import numpy as np
    from gencore_amd import synth
    from gencore_amd.capi import default_params
    from test_bamio import records_of
    d = synth.generate(workload, n_pairs=n_pairs)
    batch = d.to_batch()
    targets = [("chr%d" % (i + 1), int(l)) for i, l in enumerate(np.asarray(d.target_len, np.uint32))]
    recs = records_of(batch)
    in_b, in_s = str(tmp_path / "in.bam"), str(tmp_path / "in.sam")
    pybam.write_bam(in_b, recs, targets)
    pybam.write_sam(in_s, recs, targets)
    prm = default_params(umi_prefix="auto", cluster_size_req=d.info["supporting_reads"])
    runs, outs = {}, {}
    for src, ext in ((in_b, "bam"), (in_b, "sam"), (in_s, "bam"), (in_s, "sam")):
        out = str(tmp_path / ("out_%s.%s" % (src[-3:], ext)))
        runs[(src[-3:], ext)] = bamio.run_bam(src, out, prm, threads=4, chunk_reads=5000 if ext == "sam" else 1 << 21, level=1)
        outs[(src[-3:], ext)] = out
    base = runs[("bam", "bam")]
    assert base.n_reads == batch.n and base.n_out > 100
    for k, r in runs.items():
        assert (r.n_reads, r.n_out) == (base.n_reads, base.n_out), k
        assert bytes(r.pre) == bytes(base.pre) and bytes(r.post) == bytes(base.post), k
    _, tg, got = pybam.read_bam(outs[("bam", "bam")])
    want_lines = [pybam.sam_fields_of_read(g, tg) for g in got]
    for k in (("bam", "sam"), ("sam", "sam")):
        text, lines = pybam.read_sam(outs[k])
        assert "@SQ\tSN:chr1\tLN:%d\n" % targets[0][1] in text
        assert lines == want_lines, k
    _, tg2, got2 = pybam.read_bam(outs[("sam", "bam")])
    assert tg2 == tg and [pybam.sam_fields_of_read(g, tg2) for g in got2] == want_lines


@pytest.mark.gpu
def test_sam_without_contigs_and_header_only(built, tmp_path):
    """src/gencore.cpp:186-189: a SAM file whose header names no contig is refused ("this SAM file has no header"); a header without
    alignments gives an output that holds the header and nothing else."""
    from gencore_amd.capi import default_params
    prm = default_params(umi_prefix="auto")
    bad = tmp_path / "nohdr.sam"
    open(bad, "w").write("a\t99\tc1\t11\t60\t4M\t=\t21\t14\tACGT\tIIII\n")
    with pytest.raises(GceError) as ei:
        bamio.run_bam(str(bad), str(tmp_path / "o.bam"), prm, threads=2)
    assert "no header" in str(ei.value)
    only = tmp_path / "only.sam"
    open(only, "w").write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c1\tLN:1000\n")
    for ext in ("sam", "bam"):
        out = str(tmp_path / ("o2." + ext))
        r = bamio.run_bam(str(only), out, prm, threads=2)
        assert r.n_reads == 0 and r.n_out == 0
        if ext == "sam":
            text, lines = pybam.read_sam(out)
            assert text == "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c1\tLN:1000\n" and lines == []
        else:
            _, tg, recs = pybam.read_bam(out)
            assert tg == [("c1", 1000)] and recs == []
