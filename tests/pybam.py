"""Independent pure-Python BAM writer / reader for the tests (struct + zlib + gzip only, SAMv1 section 4): the writer makes the
input files of tests/test_bamio.py from python records, the reader checks what gencore_amd/csrc/bamio.cpp writes."""
import gzip
import struct
import zlib

from gencore_amd.batch import pack_seq, parse_cigar  # noqa: F401 (parse_cigar re-exported for the tests)


def bgzf_block(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    comp = co.compress(data) + co.flush()
    bsize = 18 + len(comp) + 8 - 1
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


EOF_BLOCK = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def aux_bytes(tag, typ, val):
    t = tag.encode()
    if typ == "Z":
        return t + b"Z" + val.encode() + b"\0"
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "A": "<c", "f": "<f"}[typ]
    return t + typ.encode() + struct.pack(fmt, val)


def record_bytes(r):
    """r: dict as ReadBatch.from_records takes, plus optional 'aux_pre' / 'aux_post' lists of (tag, type, value) around NM / MI."""
    name = r["qname"].encode() + b"\0"
    cig = parse_cigar(r.get("cigar", "*"))
    seq, qual = r["seq"], r["qual"]
    if isinstance(qual, str):
        qual = [ord(c) - 33 for c in qual]
    aux = b"".join(aux_bytes(*a) for a in r.get("aux_pre", []))
    if r.get("nm") is not None:
        aux += aux_bytes("NM", r.get("nm_type", "C"), r["nm"])
    if r.get("mi") is not None:
        aux += aux_bytes("MI", "Z", r["mi"])
    aux += b"".join(aux_bytes(*a) for a in r.get("aux_post", []))
    core = struct.pack("<iiBBHHHiiii", r["tid"], r["pos"], len(name), r.get("mapq", 60), r.get("bin", 4680), len(cig), r["flag"], len(seq),
                       r["mtid"], r["mpos"], r["isize"])
    body = core + name + b"".join(struct.pack("<I", w) for w in cig) + bytes(pack_seq(seq)) + bytes(qual) + aux
    return struct.pack("<i", len(body)) + body


def write_bam(path, records, targets, text="@HD\tVN:1.6\tSO:coordinate\n", block=0xff00, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    """targets: list of (name, length)."""
    stream = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(targets))
    for nm, ln in targets:
        stream += struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln)
    stream += b"".join(record_bytes(r) for r in records)
    with open(path, "wb") as f:
        for o in range(0, len(stream), block):
            f.write(bgzf_block(stream[o:o + block], level, strategy))
        f.write(EOF_BLOCK)


def read_bam(path):
    """-> (text, targets, records); a record is a dict with the fixed fields, qname, cigar words, seq (str), qual (list), aux {tag: (type, value)}."""
    raw = open(path, "rb").read()
    assert raw.endswith(EOF_BLOCK), "no BGZF EOF marker"
    u = gzip.decompress(raw)                                   # BGZF is a series of gzip members
    assert u[:4] == b"BAM\1"
    p = 4
    (lt,) = struct.unpack_from("<i", u, p); p += 4
    text = u[p:p + lt].decode(); p += lt
    (nref,) = struct.unpack_from("<i", u, p); p += 4
    targets = []
    for _ in range(nref):
        (ln,) = struct.unpack_from("<i", u, p); p += 4
        nm = u[p:p + ln - 1].decode(); p += ln
        (tl,) = struct.unpack_from("<i", u, p); p += 4
        targets.append((nm, tl))
    recs = []
    code = "=ACMGRSVTWYHKDBN"
    while p < len(u):
        (bs,) = struct.unpack_from("<i", u, p); p += 4
        e = p + bs
        tid, pos, lq, mapq, bn, nc, flag, ls, mtid, mpos, isize = struct.unpack_from("<iiBBHHHiiii", u, p)
        q = p + 32
        qname = u[q:q + lq - 1].decode(); assert u[q + lq - 1] == 0; q += lq
        cig = list(struct.unpack_from("<%dI" % nc, u, q)); q += 4 * nc
        sb = u[q:q + (ls + 1) // 2]; q += (ls + 1) // 2
        seq = "".join(code[(sb[i >> 1] >> (0 if i & 1 else 4)) & 15] for i in range(ls))
        qual = list(u[q:q + ls]); q += ls
        aux, order = {}, []
        while q < e:
            tag, typ = u[q:q + 2].decode(), chr(u[q + 2]); q += 3
            if typ == "Z":
                z = u.index(b"\0", q); val = u[q:z].decode(); q = z + 1
            else:
                fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "A": "<c", "f": "<f"}[typ]
                (val,) = struct.unpack_from(fmt, u, q); q += struct.calcsize(fmt)
            aux[tag] = (typ, val); order.append(tag)
        assert q == e
        recs.append(dict(tid=tid, pos=pos, mapq=mapq, bin=bn, flag=flag, mtid=mtid, mpos=mpos, isize=isize, qname=qname, cigar=cig, seq=seq,
                         qual=qual, aux=aux, aux_order=order))
        p = e
    return text, targets, recs
