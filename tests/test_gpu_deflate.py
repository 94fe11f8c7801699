"""The GPU BGZF encoder (gce_deflate.hpp: greedy LZ77, fixed Huffman codes, one lane per block) against zlib / gzip as DECODERS: every kind of
data, every block size, the stored fallback for bytes that fixed codes would expand; and gce_run_bam with level -2 against the same run at level 1,
read back by the independent Python BAM reader (gzip.decompress checks every member's CRC-32 and ISIZE)."""
import ctypes as C
import gzip
import struct
import zlib

import numpy as np
import pytest

from gencore_amd import capi

pytestmark = pytest.mark.gpu


def gpu_deflate(lib, data, block):
    buf = np.frombuffer(data, np.uint8) if data else np.zeros(1, np.uint8)
    nb = (len(data) + block - 1) // block
    cap = len(data) + len(data) // 8 + 64 * (nb + 1)
    out = np.zeros(cap + 8, np.uint8)
    got = C.c_size_t(0)
    lib.gce_bgzf_deflate.argtypes = [C.c_int32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    rc = lib.gce_bgzf_deflate(0, buf.ctypes.data, len(data), block, out.ctypes.data, cap, C.byref(got))
    return rc, out[:got.value].tobytes()


def members_of(blob):
    """the BGZF members of a blob: (BSIZE from the BC subfield, raw deflate bytes, CRC, ISIZE) -- walked by the framing alone"""
    out, p = [], 0
    while p < len(blob):
        assert blob[p:p + 4] == b"\x1f\x8b\x08\x04" and blob[p + 12:p + 16] == b"BC\x02\x00", p
        (bs,) = struct.unpack_from("<H", blob, p + 16)
        bs += 1
        crc, isize = struct.unpack_from("<II", blob, p + bs - 8)
        out.append((bs, blob[p + 18:p + bs - 8], crc, isize))
        p += bs
    assert p == len(blob)
    return out


def payloads(rng):
    text = (b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"read%d\t99\tchr1\t%d\t60\t150M\t=\t%d\t300\tACGT\tFFFF\tNM:i:%d\n" % (i, 1000 + i, 1200 + i, i % 3) for i in range(6000)))
    return [b"A", b"AC", b"ACG", b"ACGT", b"ACGTA", bytes(rng.integers(0, 256, 200000, dtype=np.uint8)),               # incompressible: stored blocks
            text, b"\0" * 300000, b"ab" * 100000, bytes(rng.integers(0, 4, 150000, dtype=np.uint8)),
            bytes(np.repeat(rng.integers(0, 256, 3000, dtype=np.uint8), rng.integers(1, 300, 3000))),                 # runs: distance 1, lengths up to 258 and beyond
            bytes(rng.integers(33, 74, 100000, dtype=np.uint8)),                                                      # quality-like
            (bytes(rng.integers(0, 256, 3000, dtype=np.uint8)) * 60),                                                 # distances of 3000
            (bytes(rng.integers(0, 256, 40000, dtype=np.uint8)) * 3),                                                 # distances beyond 32 768: no match allowed there
            bytes(rng.integers(144, 256, 50000, dtype=np.uint8)) + b"\x90" * 70000]                                   # 9-bit literals


@pytest.mark.parametrize("block", [1, 7, 300, 4096, 16384, 65279, 65280])
def test_gpu_deflate_round_trips_through_zlib(built, block):
    lib = capi.load_library()
    rng = np.random.default_rng(11)
    for k, d in enumerate(payloads(rng)):
        if block < 300 and len(d) > 3000:
            d = d[:3000 + k]
        rc, blob = gpu_deflate(lib, d, block)
        assert rc == 0, (k, block)
        mem = members_of(blob)
        assert len(mem) == (len(d) + block - 1) // block
        at = 0
        for bs, raw, crc, isize in mem:
            piece = d[at:at + isize]
            assert isize == min(block, len(d) - at) and bs <= 0x10000
            assert zlib.decompress(raw, -15) == piece, (k, block, at)           # a valid raw-deflate stream, ended by its final block
            assert crc == (zlib.crc32(piece) & 0xFFFFFFFF)
            at += isize
        assert at == len(d) and gzip.decompress(blob) == d                      # a series of gzip members (what a BAM reader sees)


def test_gpu_deflate_compresses(built):
    """not a ratio claim, a sanity bound: BAM-like bytes shrink, random bytes cost 5 + 26 bytes per block at most"""
    lib = capi.load_library()
    rng = np.random.default_rng(5)
    qual = bytes(rng.choice(np.array([37, 25, 11], np.uint8), 400000, p=[0.8, 0.12, 0.08]))
    rc, blob = gpu_deflate(lib, qual, 0xff00)
    assert rc == 0 and len(blob) < 0.6 * len(qual)
    rnd = bytes(rng.integers(0, 256, 300000, dtype=np.uint8))
    rc, blob = gpu_deflate(lib, rnd, 0xff00)
    assert rc == 0 and len(blob) <= len(rnd) + 31 * 5


@pytest.mark.parametrize("workload,n_pairs", [("cfg3", 30000), ("cfg2", 20000), ("cfg5", 3000)])
def test_run_bam_level_minus_2_equals_level_1(built, tmp_path, workload, n_pairs):
    """gce_run_bam with the GPU encoder (level -2) writes the records the host encoder writes, in the same order, with the same header"""
    import pybam
    from gencore_amd import synth
    from gencore_amd.bamio import run_bam
    from gencore_amd.capi import default_params
    from test_bamio import records_of
    d = synth.generate(workload, n_pairs=n_pairs)
    batch = d.to_batch()
    tl = np.asarray(d.target_len, np.uint32)
    targets = [("chr%d" % (i + 1), int(l)) for i, l in enumerate(tl)]
    src, a, b = (str(tmp_path / x) for x in ("in.bam", "host.bam", "gpu.bam"))
    pybam.write_bam(src, records_of(batch), targets)
    prm = default_params(umi_prefix="auto", cluster_size_req=d.info["supporting_reads"])
    r1 = run_bam(src, a, prm, threads=4, level=1)
    r2 = run_bam(src, b, prm, threads=4, level=-2)
    assert (r1.n_reads, r1.n_out) == (r2.n_reads, r2.n_out) and r1.n_out > 0
    t1, g1, x1 = pybam.read_bam(a)
    t2, g2, x2 = pybam.read_bam(b)
    assert (t1, g1) == (t2, g2) and x1 == x2
