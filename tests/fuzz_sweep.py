"""Fuzz sweep: tests/fuzzgen.py cases over a range of seeds the test-suite does not hold, engine vs oracle (every record, both
Stats blocks, order, error status).  Run on the GPU box:  python tests/fuzz_sweep.py [first_seed] [count] [n_mol] [summary.json]
The summary (seed range, streams, reads, records, error-path streams, mismatches with their seeds) is the artefact that goes to profiles/."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import fuzzgen
from parity_helpers import check_output_order, diff_results
from gencore_amd.capi import GceError
from gencore_amd.engine import run_stream
from oracle import oracle_py

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n_mol = int(sys.argv[3]) if len(sys.argv) > 3 else 60
summary_path = sys.argv[4] if len(sys.argv) > 4 else None
bad, bad_seeds, n_reads, n_rec, n_err, modes, t0 = 0, [], 0, 0, 0, {}, time.time()
for seed in range(first, first + count):
    kw = dict(n_mol=n_mol + seed % 40, exotic=seed % 3 == 0)
    if seed % 5 == 0: kw["period"] = 3 + seed % 50
    if seed % 7 == 0: kw["deep"] = 20 + seed % 90
    if seed % 4 == 1: kw["umi_mode"] = ("none", "prefix", "colon", "duplex", "mi")[seed // 4 % 5]
    b, over, ref, cl = fuzzgen.make_case(seed, **kw)
    p = fuzzgen.make_params(over, cl)
    want = oracle_py.run(b, p, ref)
    n_reads += b.n; n_err += want.status != 0
    for k_ in ("period", "deep", "umi_mode", "exotic"):
        if kw.get(k_): modes[k_] = modes.get(k_, 0) + 1
    try:
        got, st = run_stream(b, p, ref), 0
    except GceError as e:
        got, st = None, e.status
    if st != want.status:
        bad += 1; bad_seeds.append(seed); print("SEED", seed, kw, "status", st, "want", want.status, flush=True)
    elif got is not None:
        n_rec += len(got.rows["src"])
        d = diff_results(b, got, want) + check_output_order(b, got.rows)
        if d:
            bad += 1; bad_seeds.append(seed); print("SEED", seed, kw, "DIFF", d[:3], flush=True)
print("fuzz sweep: seeds %d..%d, %d mismatches" % (first, first + count - 1, bad))
if summary_path:
    import subprocess
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("GCE_HEAD", "?")
    json.dump(dict(what="tests/fuzz_sweep.py: engine (C-ABI) vs oracle on tests/fuzzgen.py streams outside the suite's seeds: every emitted record, both Stats blocks, "
                        "bamComp order, error status", head=head, first_seed=first, streams=count, molecules_per_stream="%d..%d" % (n_mol, n_mol + 39),
                   reads=int(n_reads), records_compared=int(n_rec), streams_on_an_error_path=int(n_err), streams_by_feature=modes, mismatches=bad, mismatching_seeds=bad_seeds,
                   seconds=round(time.time() - t0, 1)), open(summary_path, "w"), indent=1)
