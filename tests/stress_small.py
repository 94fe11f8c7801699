"""Stress: tiny streams (the error-path cases and a few small fuzz cases) hundreds of times in one process, to flush out
races / uninitialised reads that a single pytest pass rarely hits.  Prints every mismatch in full."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import fuzzgen
from gencore_amd.batch import ReadBatch
from parity_helpers import diff_results
from gencore_amd.capi import GceError, default_params
from gencore_amd.engine import run_stream
from oracle import oracle_py

tl = np.asarray([100000], np.uint32)
base = dict(flag=99, tid=0, cigar="20M", mtid=0, isize=50, seq="ACGTACGTACGTACGTACGT", qual=[37] * 20, nm=0)
ref_seq = "ACGTACGTACGTACGTACGT"
lowq = [37] * 20; lowq[3] = 2
bad = ref_seq[:3] + "A" + ref_seq[4:]
refnib = oracle_py.pack_reference("G" * 100 + ref_seq + "G" * 200)
cases = [
    ("unsorted", [dict(base, qname="a", pos=500, mpos=530), dict(base, qname="b", pos=100, mpos=130)], {}, []),
    ("mi_mismatch", [dict(base, qname="a", pos=100, mpos=130, mi="x:AAAA"), dict(base, qname="a", flag=147, pos=130, mpos=100, isize=-50, mi="x:CCCC")], {}, []),
    ("mi_ok", [dict(base, qname="a", pos=100, mpos=130, mi="AAAA"), dict(base, qname="a", flag=147, pos=130, mpos=100, isize=-50, mi="CCCC")], {}, []),
    ("umi_parse", [dict(base, qname="readUI", pos=100, mpos=130)], dict(umi_prefix="UMI"), []),
    ("nm_missing", [dict(base, qname="a", pos=100, mpos=130, seq=bad, qual=lowq, nm=None), dict(base, qname="a", flag=147, pos=130, mpos=100, isize=-50, nm=None)], {}, [(refnib, 320)]),
    ("single", [dict(base, qname="a", pos=100, mpos=130)], {}, []),
    ("pair", [dict(base, qname="a", pos=100, mpos=130), dict(base, qname="a", flag=147, pos=130, mpos=100, isize=-50)], {}, []),
]
prepared = []
for name, recs, over, ref in cases:
    b = ReadBatch.from_records(recs)
    p = default_params(n_targets=1, target_len=tl.ctypes.data, **over)
    prepared.append((name, b, p, ref, oracle_py.run(b, p, ref)))
for seed in (3, 7, 29, 41, 600):
    b, over, ref, cl = fuzzgen.make_case(seed, n_mol=25, exotic=seed >= 600)
    p = fuzzgen.make_params(over, cl)
    prepared.append(("fuzz%d" % seed, b, p, ref, oracle_py.run(b, p, ref)))
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad_n = 0
import torch
for it in range(n_iter):
    if it % 5 == 0:                      # dirty freed device memory so that uninitialised reads see garbage, not zeros
        g = torch.empty(1 << 28, dtype=torch.uint8, device='cuda').fill_(0xAB + it % 7); del g; torch.cuda.empty_cache()
    for name, b, p, ref, want in prepared:
        try:
            got = run_stream(b, p, ref)
            st = 0
        except GceError as e:
            got, st = None, e.status
        if st != want.status:
            bad_n += 1; print("ITER", it, name, "status", st, "want", want.status, flush=True)
        elif got is not None:
            d = diff_results(b, got, want)
            if d:
                bad_n += 1; print("ITER", it, name, "DIFF", d[:3], flush=True)
print("stress done: %d iterations, %d mismatches" % (n_iter, bad_n))
