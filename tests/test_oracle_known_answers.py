"""The oracle against EVERY known-answer vector the reference's own `gencore test` holds for this path
(SURVEY.md section 4 / 8c): BamUtil::test (src/bamutil.cpp:385-423) and Cluster::test (src/cluster.cpp:275-288).
These are the only outputs of the path the reference pins; everything else is 'parity unpinned'."""
import numpy as np
import pytest

GETUMI = [  # (qname, prefix, expected)   src/bamutil.cpp:386-411
    ("NB551106:8:H5Y57BGX2:1:13304:3538:1404", "", ""),
    ("NB551106:8:H5Y57BGX2:1:13304:3538:1404:UMI_GAGCATAC", "UMI", "GAGCATAC"),
    ("NB551106:8:H5Y57BGX2:1:13304:3538:1404:UMI_GAGC_ATAC", "UMI", "GAGC_ATAC"),
    ("NB551106:8:H5Y57BGX2:1:13304:3538:1404:GAGC_ATAC", "", "GAGC_ATAC"),
    ("NB551106:8:H5Y57BGX2:1:13304:3538:1404:UMI_X", "UMI", ""),
    ("@V300034954L1C001R0040000002/1:UMI_ATG_AAT", "UMI", "ATG_AAT"),
    ("@V300034954L1C001R0040000002:UMI_ATG_AAT /1", "UMI", "ATG_AAT"),
]
UMIDIFF = [("ATCGATCG", "ATCGATCG", 0), ("ATCGATCG", "ATCGTTC", 2), ("ATCGATCG", "ATCGTTCG", 1), ("AAAA_ATCG", "AAAA_ATCG", 0)]  # cluster.cpp:277-280
ISDUPLEX = [("ATCG_CTAG", "CTAG_ATCG", True), ("AGC_TGA", "TGA_AGC", True), ("AAAA_AAAA", "AAAA_AAAA", True),
            ("CTAG", "CTAG_ATCG", False), ("CTAG", "CCCAGG", False), ("", "", False)]  # cluster.cpp:281-286


@pytest.mark.parametrize("qname,prefix,want", GETUMI)
def test_get_umi_reference_vectors(oracle, qname, prefix, want):
    assert oracle.get_umi(qname, prefix) == want


@pytest.mark.parametrize("a,b,want", UMIDIFF)
def test_umi_diff_reference_vectors(oracle, a, b, want):
    assert oracle.umi_diff(a, b) == want


@pytest.mark.parametrize("a,b,want", ISDUPLEX)
def test_is_duplex_reference_vectors(oracle, a, b, want):
    assert oracle.is_duplex(a, b) is want


def test_get_umi_edge_semantics(oracle):
    # find_last_of(prefix) matches ANY char of the prefix (quirk Q10): the 'M' of "1Mxy" wins over the real tag
    assert oracle.get_umi("r:UMI_ACGT:1Mxy", "UMI") == ""
    assert oracle.get_umi("r:UMI_ACGT:1MxAC", "UMI") == "AC"
    assert oracle.get_umi("readUx", "UMI") == ""         # start == len -> substr(len, 0) == ""
    assert oracle.get_umi("x:ACGTN", "") == ""           # N invalidates the colon form
    assert oracle.get_umi("x:A_C_G", "") == ""           # more than one underscore
    assert oracle.get_umi("x:_ACG", "") == "ACG"         # one leading underscore is skipped
    assert oracle.get_umi("x:", "") == ""
    assert oracle.get_umi("nocolon", "") == ""


def test_get_umi_throw_is_reported(oracle):
    # start = pos+2 beyond the end of the name: std::string::substr throws std::out_of_range in the reference
    for name in ("readU", "read_U", "readUI", "r:UMI_ACGT:1M"):
        assert oracle.get_umi(name, "UMI") is None


def test_split_semantics_behind_is_duplex(oracle):
    # util.h:59-88: leading separators are skipped, a trailing one yields an extra empty token
    assert oracle.is_duplex("A_", "_A") is False         # "_A" has ONE token
    assert oracle.is_duplex("A_", "A_") is False         # ["A",""] vs ["A",""]: A != "" -> False
    assert oracle.is_duplex("A__B", "B__A") is False     # three tokens
    assert oracle.is_duplex("_A_B", "B_A") is True       # leading '_' skipped


def cig(s):
    from gencore_amd.batch import parse_cigar
    return parse_cigar(s)


@pytest.mark.parametrize("part,whole,left,want", [
    ("100M", "100M", True, True),
    ("90M", "100M", True, True),               # shorter last op
    ("100M", "90M", True, False),
    ("50M2I48M", "50M2I48M", True, True),
    ("50M2I40M", "50M2I48M", True, True),
    ("40M2I48M", "50M2I48M", True, False),     # length mismatch before the last op
    ("40M10H", "50M10H", True, True),          # shorter op followed by a trailing hard clip
    ("40M10S", "50M10S", True, False),
    ("5S95M", "5S95M", False, True),
    ("90M", "5S95M", False, True),             # right aligned: compared from the end
    ("5S95M", "95M", False, False),            # whole has fewer ops
    ("", "10M", True, True),                   # no CIGAR: trivially part of anything
])
def test_is_part_of(oracle, part, whole, left, want):
    assert oracle.is_part_of(cig(part), cig(whole), left) is want


@pytest.mark.parametrize("cigar,pos,want", [
    ("10M", 0, 0), ("10M", 9, 9), ("10M", 10, -1), ("3S7M", 1, -1), ("3S7M", 3, 0), ("5M2I5M", 5, -1), ("5M2I5M", 7, 5),
    ("5M2D5M", 5, 7), ("5H5M", 0, 0), ("", 0, -1)])
def test_ref_offset(oracle, cigar, pos, want):
    assert oracle.ref_offset(cig(cigar), pos) == want


def test_reference_packing_matches_fastareader_layout(oracle):
    # FastaReader::to4bits: A=1,T=2,C=3,G=4, other=0; LOW nibble = even position (src/fastareader.cpp:106-113,139-152)
    got = oracle.pack_reference("ATCGN")
    assert got.tolist() == [1 | (2 << 4), 3 | (4 << 4), 0]


def test_is_duplex_against_the_reference_tokenizer(oracle):
    """The one reference file on the path that compiles without htslib is src/util.h; oracle/ref_probe builds its `split`
    (the tokenizer behind Cluster::isDuplex, cluster.cpp:246-258) from the reference source in place into
    oracle/_ref/libref_util.so.  The oracle's restatement must agree with it on every UMI shape, including the odd ones
    (leading / trailing / doubled underscores, empty strings)."""
    import ctypes as C
    import itertools
    import os
    import random
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_util.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libref_util.so not built (needs /root/reference at build time)")
    ref = C.CDLL(so)
    ref.ref_is_duplex.argtypes = [C.c_char_p, C.c_char_p]
    ref.ref_split.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(256)
    assert ref.ref_split(b"_A__B_", b"_", buf, 256) == 4 and buf.value == b"A\n\nB\n"      # leading sep skipped, trailing one adds ""
    alphabet = ["", "A", "AC", "_", "__", "A_", "_A", "A_C", "AC_GT", "GT_AC", "A__C", "_A_C", "A_C_", "A_C_G", "ACGT_TTGA", "TTGA_ACGT"]
    cases = list(itertools.product(alphabet, alphabet))
    rng = random.Random(7)
    for _ in range(3000):
        mk = lambda: "".join(rng.choice("ACGT_") for _ in range(rng.randint(0, 9)))
        a = mk()
        b = mk() if rng.random() < 0.5 else "_".join(reversed(a.split("_")))
        cases.append((a, b))
    for a, b in cases:
        assert oracle.is_duplex(a, b) == bool(ref.ref_is_duplex(a.encode(), b.encode())), (a, b)
