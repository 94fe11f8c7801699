"""The GPU BGZF decoder (gce_inflate.hpp) against zlib: every deflate block type, sizes 0 .. 65 280, data that compresses in every way
(random, text, runs, short and long distances), damaged members."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from gencore_amd import capi

pytestmark = pytest.mark.gpu


def member(data, level, strategy=zlib.Z_DEFAULT_STRATEGY, extra=b""):
    """one BGZF member (SAM spec 4.1): gzip header with the BC subfield (and optionally more subfields in front), raw deflate, CRC-32, ISIZE"""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    body = co.compress(data) + co.flush()
    xlen = len(extra) + 6
    bsize = 12 + xlen + len(body) + 8
    assert bsize <= 0x10000
    return b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", xlen) + extra + b"BC\x02\0" + struct.pack("<H", bsize - 1) + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def gpu_inflate(lib, members, sizes):
    blob = b"".join(members)
    coff = np.cumsum([0] + [len(m) for m in members[:-1]]).astype(np.uint64) if members else np.zeros(0, np.uint64)
    csize = np.array([len(m) for m in members], np.uint32)
    usize = np.array(sizes, np.uint32)
    out = np.zeros(int(usize.sum()) + 8, np.uint8)
    bad = C.c_int32(-2)
    buf = np.frombuffer(blob, np.uint8) if blob else np.zeros(1, np.uint8)
    rc = lib.gce_bgzf_inflate(0, buf.ctypes.data, len(blob), len(members), coff.ctypes.data, csize.ctypes.data, usize.ctypes.data, out.ctypes.data, C.byref(bad))
    return rc, bad.value, out[:int(usize.sum())].tobytes()


def payloads(rng):
    text = (b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"read%d\t99\tchr1\t%d\t60\t150M\t=\t%d\t300\tACGT\tFFFF\tNM:i:%d\n" % (i, 1000 + i, 1200 + i, i % 3) for i in range(900)))
    out = [b"", b"A", b"AC", bytes(rng.integers(0, 256, 1, dtype=np.uint8)), bytes(rng.integers(0, 256, 65280, dtype=np.uint8)), text[:65280], b"\0" * 65280, b"ab" * 30000,
           bytes(rng.integers(0, 4, 65000, dtype=np.uint8)),                                   # long Huffman codes are rare here, short ones dominate
           bytes(np.repeat(rng.integers(0, 256, 700, dtype=np.uint8), rng.integers(1, 200, 700)))[:65280],   # runs: distance 1, overlapping copies
           bytes(rng.integers(33, 74, 40000, dtype=np.uint8)),                                  # quality-like
           (bytes(rng.integers(0, 256, 3000, dtype=np.uint8)) * 22)[:65280]]                    # distances of 3000
    for n in (2, 3, 7, 8, 9, 255, 256, 257, 258, 259, 4095, 32768, 32769, 65279):
        out.append(text[:n])
    return out


def test_every_block_type_equals_zlib(built):
    lib = capi.load_library()
    rng = np.random.default_rng(7)
    datas, members = [], []
    for d in payloads(rng):
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            if level == 0 and len(d) > 65000:
                d2 = d[:65000]                                                                   # (a stored member needs its five header bytes per 65 535)
            else:
                d2 = d
            try:
                m = member(d2, level, strategy)
            except AssertionError:
                continue                                                                       # incompressible at this level: does not fit a BGZF member
            datas.append(d2); members.append(m)
    assert len(members) > 150
    rc, bad, got = gpu_inflate(lib, members, [len(d) for d in datas])
    assert rc == 0 and bad == -1, (rc, bad)
    assert got == b"".join(datas)


def test_extra_subfields_and_several_deflate_blocks_in_one_member(built):
    lib = capi.load_library()
    rng = np.random.default_rng(11)
    a, b, c = bytes(rng.integers(0, 256, 20000, dtype=np.uint8)), b"ACGT" * 6000, bytes(rng.integers(65, 70, 15000, dtype=np.uint8))
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(a) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(b) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(c) + co.flush()      # stored (empty) blocks between dynamic ones
    data = a + b + c
    extra = b"XY\x03\0abc"
    xlen = len(extra) + 6
    bsize = 12 + xlen + len(body) + 8
    m = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", xlen) + extra + b"BC\x02\0" + struct.pack("<H", bsize - 1) + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))
    rc, bad, got = gpu_inflate(lib, [m, member(b"tail", 6)], [len(data), 4])
    assert rc == 0 and bad == -1 and got == data + b"tail"


@pytest.mark.parametrize("what", ["crc", "isize_short", "isize_long", "body", "truncated", "btype3"])
def test_damaged_members_are_reported(built, what):
    lib = capi.load_library()
    rng = np.random.default_rng(13)
    good = bytes(rng.integers(0, 64, 30000, dtype=np.uint8))
    m0, m2 = member(b"first member" * 100, 6), member(b"third" * 1000, 1)
    m1 = bytearray(member(good, 6)); n1 = len(good)
    if what == "crc":
        m1[-8] ^= 1
    elif what == "isize_short":
        n1 -= 1
    elif what == "isize_long":
        n1 += 1
    elif what == "body":
        for k in range(40, len(m1) - 8, 97):
            m1[k] ^= 0x55
    elif what == "truncated":
        m1 = bytearray(m1[:18] + m1[18:len(m1) // 2 - 8] + m1[-8:])                            # half of the deflate data gone (BSIZE kept consistent by the caller's csize)
    elif what == "btype3":
        m1[18] = (m1[18] & ~0x06) | 0x06
    rc, bad, got = gpu_inflate(lib, [m0, bytes(m1), m2], [1200, n1, 5000])
    assert rc == -1 and bad == 1, (rc, bad)
    assert got[:1200] == b"first member" * 100                                                 # the members around it are delivered


def _members(blob):
    """(offset, csize, isize) of every BGZF member of a file image"""
    out, off = [], 0
    while off < len(blob):
        xlen = struct.unpack_from("<H", blob, off + 10)[0]
        x, bsize = 0, None
        while x + 4 <= xlen:
            si1, si2, sl = blob[off + 12 + x], blob[off + 13 + x], struct.unpack_from("<H", blob, off + 14 + x)[0]
            if si1 == 66 and si2 == 67 and sl == 2:
                bsize = struct.unpack_from("<H", blob, off + 16 + x)[0] + 1
            x += 4 + sl
        out.append((off, bsize, struct.unpack_from("<I", blob, off + bsize - 4)[0]))
        off += bsize
    return out


def test_raw_stream_of_host_windows_and_gpu_members_equals_the_batch(built, tmp_path):
    """gce_raw_push (bytes the host inflated) and gce_raw_push_bgzf (members for the GPU) interleaved, with a capacity hint so small that the
    raw stream and the compressed staging both grow on the way: the stream that comes out is the file's, record for record."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import fuzzgen
    from gencore_amd.bamio import write_batch_as_bam
    from gencore_amd.engine import run_stream
    lib = capi.load_library()
    batch, over, reference, contig_len = fuzzgen.make_case(4242, n_mol=400)
    params = fuzzgen.make_params(over, contig_len)
    want = run_stream(batch, params, reference)
    path = tmp_path / "in.bam"
    write_batch_as_bam(path, batch, contig_len, level=6)
    blob = path.read_bytes()
    mem = [m for m in _members(blob) if m[2] > 0]
    assert len(mem) >= 6
    stream = b"".join(zlib.decompress(blob[o + 18:o + c - 8], -15) for o, c, _ in mem)
    l_text = struct.unpack_from("<I", stream, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<I", stream, p)[0]; p += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<I", stream, p)[0]; p += 4 + ln + 4
    hdr_end = p
    assert hdr_end < mem[0][2]                                                                  # the header lies in the first member

    lib.gce_raw_begin.argtypes = [C.c_void_p, C.c_size_t]
    lib.gce_raw_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32)]
    lib.gce_raw_push_bgzf.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    lib.gce_raw_finish.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.POINTER(C.c_int64)]
    from gencore_amd.engine import Engine
    E = Engine(params)
    eng = E._h
    try:
        for tid, (nib, ln) in enumerate(reference):
            if nib is not None:
                E.set_reference(tid, nib, ln)
        assert lib.gce_raw_begin(eng, 1024) == 0

        def push_host(k):
            o, c, u = mem[k]
            data = np.frombuffer(zlib.decompress(blob[o + 18:o + c - 8], -15), np.uint8).copy()
            tk = C.c_int32()
            assert lib.gce_raw_push(eng, data.ctypes.data, len(data), C.byref(tk)) == 0
            assert lib.gce_submit_wait(eng, tk.value) == 0

        def push_gpu(k0, k1):
            o0 = mem[k0][0]; o1 = mem[k1 - 1][0] + mem[k1 - 1][1]
            piece = np.frombuffer(blob[o0:o1], np.uint8).copy()
            coff = np.array([m[0] - o0 for m in mem[k0:k1]], np.uint64); cs = np.array([m[1] for m in mem[k0:k1]], np.uint32); us = np.array([m[2] for m in mem[k0:k1]], np.uint32)
            tk = C.c_int32()
            assert lib.gce_raw_push_bgzf(eng, piece.ctypes.data, len(piece), k1 - k0, coff.ctypes.data, cs.ctypes.data, us.ctypes.data, C.byref(tk)) == 0
            assert lib.gce_submit_wait(eng, tk.value) == 0

        n = len(mem)
        push_host(0); push_gpu(1, 3); push_host(3); push_gpu(4, n - 1); push_gpu(n - 1, n)
        n_rec = C.c_int64()
        assert lib.gce_raw_finish(eng, hdr_end, n_ref, C.byref(n_rec)) == 0, lib.gce_last_error(eng)
        assert n_rec.value == batch.n
        assert lib.gce_process(eng) == 0, lib.gce_last_error(eng)
        rows, pre, post = E.rows()
        for k in ("src", "kind", "qname_src", "nm_new", "fr", "rr", "mate"):
            assert np.array_equal(rows[k], want.rows[k]), k
        from gencore_amd.batch import table_from_rows
        from parity_helpers import diff_results
        got = table_from_rows(batch, rows, pre, post)                                            # (bases and qualities record by record: the pad bytes between records differ by construction)
        d = diff_results(batch, got, want)
        assert not d, d[:3]
    finally:
        E.close()
