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
