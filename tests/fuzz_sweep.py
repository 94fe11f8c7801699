"""Fuzz sweep: tests/fuzzgen.py cases over a range of seeds the test-suite does not hold, engine vs oracle (every record, both
Stats blocks, order, error status).  Run on the GPU box:  python tests/fuzz_sweep.py [first_seed] [count] [n_mol]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import fuzzgen
from parity_helpers import diff_results
from gencore_amd.capi import GceError
from gencore_amd.engine import run_stream
from oracle import oracle_py

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n_mol = int(sys.argv[3]) if len(sys.argv) > 3 else 60
bad = 0
for seed in range(first, first + count):
    kw = dict(n_mol=n_mol + seed % 40, exotic=seed % 3 == 0)
    if seed % 5 == 0: kw["period"] = 3 + seed % 50
    if seed % 7 == 0: kw["deep"] = 20 + seed % 90
    if seed % 4 == 1: kw["umi_mode"] = ("none", "prefix", "colon", "duplex")[seed // 4 % 4]
    b, over, ref, cl = fuzzgen.make_case(seed, **kw)
    p = fuzzgen.make_params(over, cl)
    want = oracle_py.run(b, p, ref)
    try:
        got, st = run_stream(b, p, ref), 0
    except GceError as e:
        got, st = None, e.status
    if st != want.status:
        bad += 1; print("SEED", seed, kw, "status", st, "want", want.status, flush=True)
    elif got is not None:
        d = diff_results(b, got, want)
        if d:
            bad += 1; print("SEED", seed, kw, "DIFF", d[:3], flush=True)
print("fuzz sweep: seeds %d..%d, %d mismatches" % (first, first + count - 1, bad))
