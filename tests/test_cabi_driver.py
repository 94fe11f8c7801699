"""The C-ABI exercised by a C caller (tests/cabi_driver.c, gcc): create -> set_reference_ascii -> submit x2 -> process -> drain.
CPU: the driver compiles and links against libgencore_amd.so and, with no device, fails loudly with GCE_ERR_NO_DEVICE (no CPU
fallback).  GPU: its table equals the oracle's, row by row."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = 0x3142414345434721


def build_driver(tmp):
    exe = os.path.join(tmp, "cabi_driver")
    libdir = os.path.join(ROOT, "gencore_amd", "csrc")
    subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi_driver.c"),
                    "-L" + libdir, "-lgencore_amd", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    return exe


def dump(path, batch, target_len, contigs, cluster_size_req, flush_period, umi_prefix):
    s = batch
    with open(path, "wb") as f:
        f.write(struct.pack("<7Q", MAGIC, s.n, len(target_len), s.qname.size, s.cigar.size, s.seq.size, s.qual.size))
        f.write(struct.pack("<2i", cluster_size_req, flush_period))
        f.write(umi_prefix.encode().ljust(32, b"\0"))
        f.write(np.asarray(target_len, np.uint32).tobytes())
        for t in range(len(target_len)):
            c = contigs[t] if t < len(contigs) else None
            f.write(struct.pack("<Q", 0 if c is None else len(c)))
            if c is not None:
                f.write(c if isinstance(c, bytes) else c.encode())
        for a, dt in ((s.core, None), (s.qname_off, np.uint64), (s.cigar_off, np.uint64), (s.seq_off, np.uint64), (s.qual_off, np.uint64),
                      (s.nm, np.int32), (s.nm_type, np.uint8), (s.qname, np.uint8), (s.cigar, np.uint32), (s.seq, np.uint8), (s.qual, np.uint8)):
            f.write(np.ascontiguousarray(a if dt is None else np.asarray(a, dt)).tobytes())


def read_table(path, batch):
    raw = open(path, "rb").read()
    status, n_out = struct.unpack_from("<2q", raw, 0)
    o = 16
    rows = []
    for _ in range(n_out):
        src, kind, qsrc, nm_new, fr, rr, mate, lq = struct.unpack_from("<IBIihhII", raw, o)
        o += struct.calcsize("<IBIihhII")
        seq = raw[o:o + (lq + 1) // 2]; o += (lq + 1) // 2
        qual = raw[o:o + lq]; o += lq
        rows.append((src, kind, qsrc, nm_new, fr, rr, mate, seq, qual))
    stats = np.frombuffer(raw, np.int64, offset=o)
    return status, rows, stats


def ascii_of(reference):
    """(FastaReader nibbles, length) per contig -> the ASCII bases gce_set_reference_ascii takes; None = contig absent."""
    code = np.frombuffer(b"NATCG" + b"N" * 11, np.uint8)
    out = []
    for nib, ln in reference or []:
        if nib is None:
            out.append(None)
            continue
        both = np.empty(len(nib) * 2, np.uint8)
        both[0::2] = nib & 0xF; both[1::2] = nib >> 4
        out.append(code[both[:ln]].tobytes())
    return out


def small_case():
    import fuzzgen
    batch, over, reference, contig_len = fuzzgen.make_case(7, n_mol=80, umi_mode="prefix", period=50)
    return batch, over, reference, contig_len


def test_driver_builds_and_fails_loudly_without_a_device(built, tmp_path):
    import torch
    exe = build_driver(str(tmp_path))
    if torch.cuda.is_available():
        pytest.skip("a device is present: covered by the gpu test")
    batch, over, reference, contig_len = small_case()
    dump(str(tmp_path / "in.bin"), batch, list(contig_len), [], 1, over.get("flush_period", 10000), over.get("umi_prefix", ""))
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 3 and "gce status -2" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["fuzz", "cfg3"])
def test_c_caller_matches_oracle(built, tmp_path, case):
    from gencore_amd.capi import default_params
    from oracle import oracle_py
    exe = build_driver(str(tmp_path))
    if case == "fuzz":
        batch, over, reference, contig_len = small_case()
        tl, contigs, csr, period, prefix = list(contig_len), [], over.get("cluster_size_req", 1), over.get("flush_period", 10000), over.get("umi_prefix", "")
        ref = reference
        ascii_contigs = ascii_of(reference)
        contigs = ascii_contigs
    else:
        from gencore_amd import synth
        d = synth.generate("cfg3", n_pairs=30000)
        batch = d.to_batch()
        tl, csr, period, prefix = list(d.target_len), d.info["supporting_reads"], 10000, d.info["umi_prefix"]
        ref = d.reference_host()
        contigs = ascii_of(ref)
    tla = np.asarray(tl, np.uint32)
    prm = default_params(n_targets=len(tla), target_len=tla.ctypes.data, umi_prefix=prefix, cluster_size_req=csr, flush_period=period)
    want = oracle_py.run(batch, prm, ref)
    dump(str(tmp_path / "in.bin"), batch, tl, contigs, csr, period, prefix)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr)
    status, rows, stats = read_table(str(tmp_path / "out.bin"), batch)
    assert status == 0 and want.status == 0
    em = want.emitted()
    assert sorted(x[0] for x in rows) == em.tolist()
    src_row = {x[0]: k for k, x in enumerate(rows)}
    for src, kind, qsrc, nm_new, fr, rr, mate, seq, qual in rows:
        assert kind == want.out_flag[src] and qsrc == want.qname_src[src] and nm_new == want.nm_new[src] and fr == want.fr[src] and rr == want.rr[src]
        wm = int(want.mate[src])
        assert (mate == 0xFFFFFFFF) == (wm == 0xFFFFFFFF) and (mate == 0xFFFFFFFF or rows[mate][0] == wm)
        lq = int(batch.core["l_qseq"][src]); so = int(batch.seq_off[src]); qo = int(batch.qual_off[src])
        wseq = bytearray(want.seq[so:so + (lq + 1) // 2].tobytes())
        gseq = bytearray(seq)
        if lq & 1:
            wseq[-1] &= 0xF0; gseq[-1] &= 0xF0
        assert gseq == wseq and qual == want.qual[qo:qo + lq].tobytes(), src
    want_stats = np.concatenate([np.frombuffer(bytes(want.pre), np.int64), np.frombuffer(bytes(want.post), np.int64)])
    assert np.array_equal(stats, want_stats)
    keys = [(int(batch.core["tid"][s]), int(batch.core["pos"][s])) for s in (x[0] for x in rows)]
    assert keys == sorted(keys)                    # bamComp's leading keys (the full order: batch.check_output_order in test_gpu_parity)
