#!/usr/bin/env python
"""Freezes oracle outputs for a few fuzz cases into oracle_regression.json (sha256 over the canonical record list +
both Stats blocks).  Run from the repo root:  python tests/golden/make_golden.py
Regression vectors of the ORACLE (test infrastructure): they detect drift of the restatement, they do not pin it to
the reference — see DESIGN.md section 5."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]

CASES = [dict(seed=1, kwargs={}), dict(seed=2, kwargs={}), dict(seed=7, kwargs=dict(umi_mode="duplex", period=11)),
         dict(seed=9, kwargs=dict(umi_mode="colon", period=3)), dict(seed=11, kwargs=dict(umi_mode="none", period=2, deep=80))]


def digest_case(seed, kwargs):
    import fuzzgen
    from oracle import oracle_py
    batch, over, reference, contig_len = fuzzgen.make_case(seed, **kwargs)
    r = oracle_py.run(batch, fuzzgen.make_params(over, contig_len), reference)
    h = hashlib.sha256()
    h.update(json.dumps(r.records(batch), sort_keys=True).encode())
    h.update(json.dumps([r.pre.as_dict(), r.post.as_dict(), r.status], sort_keys=True).encode())
    return h.hexdigest()


if __name__ == "__main__":
    out = dict(note="sha256 digests of oracle outputs (regression vectors of the oracle, not reference goldens)",
               cases=[dict(c, digest=digest_case(c["seed"], c["kwargs"])) for c in CASES])
    json.dump(out, open(os.path.join(HERE, "oracle_regression.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
