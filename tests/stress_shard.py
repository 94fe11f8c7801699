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
from gencore_amd.engine import Engine
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
        eng = None
        try:
            eng = Engine(p)
            got = eng.run(b, ref); st = 0
        except GceError as e:
            got, st = None, e.status
        if st != want.status:
            bad_n += 1; print("ITER", it, name, "status", st, "want", want.status, flush=True)
        elif got is not None:
            d = diff_results(b, got, want) + check_output_order(b, got.rows)
            if d:
                bad_n += 1; print("ITER", it, name, "DIFF", [x[:400] for x in d[:4]], flush=True)
                # where does the wrong word live?  Drain the same engine again: a clean second copy means the engine's own host copy of the Stats was
                # right and the FIRST result struct (this process's heap) was written to behind our back; the same wrong word means the engine's copy has it
                _, pre2, post2 = eng.rows()
                import numpy as _np
                w1, w2, ww = got.post.as_array(), post2.as_array(), want.post.as_array()
                print("   second drain of the same engine: post equals the oracle's: %s; equals the first drain's: %s; wrong words first drain %s, second drain %s" % (
                    bool(_np.array_equal(w2, ww)), bool(_np.array_equal(w2, w1)), _np.nonzero(w1 != ww)[0].tolist(), _np.nonzero(w2 != ww)[0].tolist()), flush=True)
        if eng is not None:
            eng.close()
print("stress_shard done: %d iterations x %d cases, %d mismatches" % (n_iter, len(prepared), bad_n))
