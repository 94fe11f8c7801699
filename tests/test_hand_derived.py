"""Hand-derived golden vectors (tests/golden/hand_derived/*.json): the expected records were worked out from the cited reference
lines, not from an oracle run.  CPU: oracle vs vector.  GPU: engine (C-ABI) vs vector."""
import glob
import json
import os

import numpy as np
import pytest

from gencore_amd.batch import ReadBatch
from parity_helpers import check_output_order
from gencore_amd.capi import default_params

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hand_derived")
CASES = sorted(os.path.basename(f)[:-5] for f in glob.glob(os.path.join(HERE, "*.json")))      # (_write_r03_vectors.py only spells out the inputs of the long ones)
KEYS = ("qname", "flag", "tid", "pos", "cigar", "seq", "qual", "nm", "fr", "rr")


def load(name):
    v = json.load(open(os.path.join(HERE, name + ".json")))
    recs = []
    for r in v["records"]:
        r = dict(r)
        n, r0 = r.pop("repeat", None), r.pop("repeat_from", 0)
        if n is None:
            recs.append(r)
        else:
            recs += [dict(r, qname=r["qname"].format(i=i)) for i in range(r0, r0 + n)]
    batch = ReadBatch.from_records(recs)
    tl = np.asarray([c["length"] for c in v["contigs"]], np.uint32)
    prm = default_params(n_targets=len(tl), target_len=tl.ctypes.data, **v["params"])
    prm._keep = tl
    contigs = []
    for c in v["contigs"]:
        s = c.get("sequence")
        contigs.append(None if s is None else s["repeat"] * s["times"])
    return v, batch, prm, contigs


def canon(recs):
    out = []
    for r in recs:
        d = {k: r[k] for k in KEYS}
        d["qname"] = d["qname"].rstrip("\0")
        d["qual"] = list(d["qual"])
        out.append(d)
    return sorted(out, key=lambda d: (d["tid"], d["pos"], d["qname"], d["flag"], d["seq"]))


def check(v, rt, batch):
    assert rt.status == v["expected_status"], (rt.status, rt.message)
    if v["expected_status"] != 0:
        return
    got, want = canon(rt.records(batch)), canon(v["expected"])
    assert len(got) == len(want), (len(got), len(want), got)
    for g, w in zip(got, want):
        assert g == w, "%s\n got  %s\n want %s" % (v["name"], g, w)
    # optional: hand-derived words of the two Stats blocks (src/stats.h:47-65; gce_stats in include/gencore_amd.h)
    for blk in ("pre", "post"):
        got_s = getattr(rt, blk).as_dict()
        for k, want_v in v.get("expected_stats", {}).get(blk, {}).items():
            got_v = {str(i): x for i, x in enumerate(got_s[k]) if x} if k == "supporting_hist" else got_s[k]
            assert got_v == want_v, "%s %s.%s: got %s want %s" % (v["name"], blk, k, got_v, want_v)


def oracle_reference(contigs):
    from oracle import oracle_py
    if all(c is None for c in contigs):
        return []
    return [(oracle_py.pack_reference(c), len(c)) if c is not None else (None, 0) for c in contigs]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_hand_derivation(oracle, name):
    v, batch, prm, contigs = load(name)
    rt = oracle.run(batch, prm, oracle_reference(contigs))
    check(v, rt, batch)


def test_vectors_present():
    assert len(CASES) >= 34


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_engine_matches_hand_derivation(built, name):
    """The HIP engine against the vector itself (no oracle in the loop); the reference goes in as ASCII (gce_set_reference_ascii)."""
    from gencore_amd.capi import GceError
    from gencore_amd.engine import Engine
    v, batch, prm, contigs = load(name)
    eng = Engine(prm)
    try:
        for tid, c in enumerate(contigs):
            if c is not None:
                eng.set_reference_ascii(tid, c)
        if v["expected_status"] != 0:
            with pytest.raises(GceError) as ei:
                eng.add_reads(batch)
                eng.finish()
            assert ei.value.status == v["expected_status"]
            return
        eng.add_reads(batch)
        eng.finish()
        rt = eng.output(batch)
        check(v, rt, batch)
        if not v.get("skip_order_check"):
            assert not check_output_order(batch, rt.rows)
    finally:
        eng.close()
