"""Stress: the contig-sharded streams of test_sharded_stream_context (tick_offset / trailing_flush) and a few fuzz seeds with short
flush periods, hundreds of times in one process with freed device memory dirtied in between -- to reproduce an intermittent mismatch
a single pytest pass rarely hits.  Prints every mismatch in full.   python tests/stress_shard.py [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import fuzzgen
from parity_helpers import check_output_order, diff_results
from gencore_amd.capi import GceError
from gencore_amd.engine import run_stream
from gencore_amd.shard import shard_by_contig
from oracle import oracle_py

prepared = []
plain = [int(x) for x in os.environ.get("STRESS_SEEDS", "11,23").split(",") if x]
for seed, kw in [(300, dict(n_mol=80, umi_mode="prefix", period=17)), (302, dict(n_mol=60, umi_mode="prefix", period=5))] + [(x, {}) for x in plain]:
    batch, over, reference, contig_len = fuzzgen.make_case(seed, **kw)
    p = fuzzgen.make_params(over, contig_len)
    prepared.append(("whole%d" % seed, batch, p, reference, oracle_py.run(batch, p, reference)))
    if kw:
        for rank in range(2):
            sub, idx, ctx = shard_by_contig(batch, 2, rank, over["flush_period"])
            ps = fuzzgen.make_params(dict(over, **ctx), contig_len)
            prepared.append(("seed%d_rank%d" % (seed, rank), sub, ps, reference, oracle_py.run(sub, ps, reference)))
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad_n = 0
for it in range(n_iter):
    if it % 3 == 0:
        g = torch.empty(1 << 28, dtype=torch.uint8, device="cuda").fill_(0xA5 + it % 11); del g; torch.cuda.empty_cache()
    for name, b, p, ref, want in prepared:
        try:
            got = run_stream(b, p, ref); st = 0
        except GceError as e:
            got, st = None, e.status
        if st != want.status:
            bad_n += 1; print("ITER", it, name, "status", st, "want", want.status, flush=True)
        elif got is not None:
            d = diff_results(b, got, want) + check_output_order(b, got.rows)
            if d:
                bad_n += 1; print("ITER", it, name, "DIFF", d[:4], flush=True)
print("stress_shard done: %d iterations x %d cases, %d mismatches" % (n_iter, len(prepared), bad_n))
