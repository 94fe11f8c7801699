"""Test infrastructure: bit-exact comparison of two result tables and the check of the output order.  Used by tests/, by
bench.py's `parity_checked` leg and by __graft_entry__.smoke(); not part of the product package."""
import numpy as np

GCE_NONE = 0xFFFFFFFF


def check_output_order(batch, rows):
    """The table must come in the order of the reference's output set: bamComp (gencore.h:19-47) with the input index as the
    final tie-break; mates must point at each other.  Returns a list of complaints."""
    bad = []
    c = batch.core[rows["src"].astype(np.int64)]
    key = list(zip(c["tid"].tolist(), c["pos"].tolist(), c["mtid"].tolist(), c["mpos"].tolist(), c["isize"].tolist(), rows["src"].tolist()))
    if key != sorted(key):
        bad.append("rows are not in bamComp order")
    m = rows["mate"]
    has = np.nonzero(m != GCE_NONE)[0]
    if len(has) and not np.array_equal(m[m[has].astype(np.int64)], has.astype(np.uint32)):
        bad.append("mate rows do not point back")
    return bad



def _differing_reads(batch, a, b, reads):
    """Subset of `reads` whose record bytes (packed bases incl. the pad nibble of odd reads masked, qualities) differ."""
    lq = batch.core["l_qseq"].astype(np.int64)[reads]
    so, qo = batch.seq_off.astype(np.int64)[reads], batch.qual_off.astype(np.int64)[reads]
    bad = np.zeros(len(reads), bool)
    for lo in range(0, len(reads), 100000):
        ln = lq[lo:lo + 100000]
        tot = int(ln.sum())
        if tot == 0:
            continue
        rid = np.repeat(np.arange(len(ln)), ln)
        within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(ln) - ln, ln)
        qi = np.repeat(qo[lo:lo + 100000], ln) + within
        dq = a.qual[qi] != b.qual[qi]
        si = np.repeat(so[lo:lo + 100000], ln) + within // 2
        sh = np.where(within % 2 == 0, 4, 0)
        ds = ((a.seq[si] >> sh) & 0xF) != ((b.seq[si] >> sh) & 0xF)
        np.logical_or.at(bad, lo + rid[dq | ds], True)
    return reads[bad]


def diff_results(batch, a, b, max_report=5):
    """Bit-exact comparison of two ResultTables over the same batch.  Returns a list of difference strings."""
    diffs = []
    for name in ("out_flag", "qname_src", "nm_new", "fr", "rr", "mate"):
        x, y = getattr(a, name), getattr(b, name)
        em = (a.out_flag != 0) | (b.out_flag != 0) if name != "out_flag" else np.ones(len(x), bool)
        bad = np.nonzero((x != y) & em)[0]
        if len(bad):
            diffs.append("%s differs at %d reads, first %s: %s vs %s" % (name, len(bad), bad[:max_report].tolist(),
                                                                       x[bad[:max_report]].tolist(), y[bad[:max_report]].tolist()))
    both = np.nonzero((a.out_flag != 0) & (b.out_flag != 0))[0]
    nbad = 0
    if len(both) > 20000:      # large streams: vectorised screen first (equal-length records), the loop below only reports
        bad_reads = _differing_reads(batch, a, b, both)
        both = bad_reads
    for i in both:
        i = int(i)
        so, n = int(batch.seq_off[i]), int(batch.core["l_qseq"][i])
        qo = int(batch.qual_off[i])
        # compare nibble-exact (ignore the pad nibble of an odd-length read)
        sa, sb = a.seq[so:so + (n + 1) // 2].copy(), b.seq[so:so + (n + 1) // 2].copy()
        if n % 2:
            sa[-1] &= 0xF0; sb[-1] &= 0xF0
        if not np.array_equal(sa, sb) or not np.array_equal(a.qual[qo:qo + n], b.qual[qo:qo + n]):
            nbad += 1
            if nbad <= max_report:
                diffs.append("read %d seq/qual differ: %s | %s ; q %s | %s" % (
                    i, batch.seq_of(i, a.seq), batch.seq_of(i, b.seq),
                    a.qual[qo:qo + n].tolist(), b.qual[qo:qo + n].tolist()))
    if nbad > max_report:
        diffs.append("... %d emitted reads differ in seq/qual" % nbad)
    pa, pb = a.pre.as_array(), b.pre.as_array()
    if not np.array_equal(pa, pb):
        diffs.append("pre stats differ: %s vs %s" % (a.pre.as_dict(), b.pre.as_dict()))
    pa, pb = a.post.as_array(), b.post.as_array()
    if not np.array_equal(pa, pb):
        diffs.append("post stats differ: %s vs %s" % (a.post.as_dict(), b.post.as_dict()))
    return diffs
