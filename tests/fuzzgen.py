"""Adversarial small-stream generator for parity tests (engine vs oracle) — exercises the quirk register of
SURVEY.md section 8 (Q1..Q14): flush cadence, unmapped / secondary / mate-unmapped reads, negative scores, absent-bin
winners, NM revert, FR wrap, UMI parsing, singleton pairs, reference edge cases, odd CIGARs, trimmed duplicates."""
import random

import numpy as np

from gencore_amd.batch import ReadBatch

QUALS = [0, 2, 10, 14, 15, 16, 19, 20, 21, 29, 30, 31, 37, 40]


def _mutate(seq, rate, rng):
    out = []
    for ch in seq:
        if rng.random() < rate:
            out.append(rng.choice([c for c in "ACGT" if c != ch]) if rng.random() < 0.9 else "N")
        else:
            out.append(ch)
    return "".join(out)


def make_case(seed, n_mol=40, umi_mode=None, period=None, deep=None, exotic=False, umi_lens=(4, 6, 8)):
    """Returns (ReadBatch, params overrides dict, reference list [(nibble array|None, n_bases)], contig lengths)."""
    from oracle import oracle_py
    rng = random.Random(seed)
    contig_len = [rng.randint(4000, 9000), 260000, rng.randint(3000, 6000)]
    contigs = []
    for ln in contig_len:
        s = [rng.choice("ACGT") for _ in range(ln)]
        for _ in range(3):                       # a few N runs
            a = rng.randrange(ln - 50)
            for i in range(a, a + rng.randint(1, 30)):
                s[i] = "N"
        contigs.append("".join(s))
    umi_mode = umi_mode if umi_mode is not None else rng.choice(["none", "prefix", "colon", "duplex", "prefix"])
    recs = []
    serial = 0
    for m in range(n_mol):
        tid = rng.choice([0, 0, 1, 2])
        L = rng.randint(40, 110)
        ins = rng.randint(max(30, L // 2), 3 * L)
        far = tid == 1 and rng.random() < 0.15
        start = rng.randint(10, contig_len[tid] - (ins if not far else 150000) - 3 * L - 20)
        depth = deep if (deep and m == 0) else rng.choice([1, 1, 2, 3, 4, 6, 9])
        umi_a = "".join(rng.choice("ACGT") for _ in range(rng.choice(list(umi_lens))))
        umi_b = "".join(rng.choice("ACGT") for _ in range(len(umi_a)))
        cross = rng.random() < 0.08
        cross_tid = rng.choice([t for t in (0, 1, 2) if t != tid])
        cross_pos = rng.randint(10, contig_len[cross_tid] - 200)
        for d in range(depth):
            serial += 1
            strand = rng.randint(0, 1)
            la = L if rng.random() < 0.8 else rng.randint(max(20, L - 15), L)      # trimmed duplicates
            lb = L if rng.random() < 0.8 else rng.randint(max(20, L - 15), L)
            fpos = start
            rend = start + (ins if not far else 140000 + ins)
            rpos = rend - lb
            if rpos < 0:
                rpos = 0

            def aligned(pos, ln):
                """returns (pos, cigar, query seq, nm)"""
                kind = rng.choices(["M", "SL", "ST", "I", "D", "H"], [70, 8, 8, 5, 5, 4])[0]
                ref = contigs[tid]
                if kind == "M" or ln < 30:
                    q = ref[pos:pos + ln]; cg = "%dM" % ln; rp = pos
                elif kind == "SL":
                    k = rng.randint(1, 8); q = "".join(rng.choice("ACGT") for _ in range(k)) + ref[pos + k:pos + ln]; cg = "%dS%dM" % (k, ln - k); rp = pos + k
                elif kind == "ST":
                    k = rng.randint(1, 8); q = ref[pos:pos + ln - k] + "".join(rng.choice("ACGT") for _ in range(k)); cg = "%dM%dS" % (ln - k, k); rp = pos
                elif kind == "I":
                    k = rng.randint(1, 3); a = rng.randint(5, ln - k - 5)
                    q = ref[pos:pos + a] + "".join(rng.choice("ACGT") for _ in range(k)) + ref[pos + a:pos + ln - k]; cg = "%dM%dI%dM" % (a, k, ln - a - k); rp = pos
                elif kind == "D":
                    k = rng.randint(1, 3); a = rng.randint(5, ln - 5)
                    q = ref[pos:pos + a] + ref[pos + a + k:pos + ln + k]; cg = "%dM%dD%dM" % (a, k, ln - a); rp = pos
                else:
                    k = rng.randint(1, 20); q = ref[pos:pos + ln]; cg = "%dM%dH" % (ln, k) if rng.random() < 0.5 else "%dH%dM" % (k, ln); rp = pos
                q = q.replace("N", "A") if rng.random() < 0.5 else q
                q2 = _mutate(q, 0.04, rng)
                nm = sum(1 for x, y in zip(q, q2) if x != y)
                return rp, cg, q2, nm

            fp, fcg, fseq, fnm = aligned(fpos, la)
            rp_, rcg, rseq, rnm = aligned(rpos, lb)
            tlen = (rend - fp) if not far else (rend - fp)
            # qname
            name = "SIM:%d:%d:%d" % (rng.randint(1, 8), rng.randint(1, 99999), serial)
            if rng.random() < 0.1:
                name += "x" * rng.randint(1, 6)            # different padded lengths inside a group
            ua = _mutate(umi_a, 0.06, rng).replace("N", "A"); ub = _mutate(umi_b, 0.06, rng).replace("N", "A")
            if umi_mode == "prefix":
                name += ":UMI_" + ua
            elif umi_mode == "colon":
                name += ":" + ua + ("_" + ub if rng.random() < 0.5 else "")
            elif umi_mode == "duplex":
                du = ua + "_" + ub if strand == 0 else ub + "_" + ua
                if rng.random() < 0.06:                     # odd shapes util.h's split sees differently (leading / doubled / trailing separator, one token, none): cluster.cpp:246-258
                    du = rng.choice(["_" + du, ua + "__" + ub, du + "_", ua + "_", "_" + ua, "_", ua])
                name += ":UMI_" + du
            mi_tag = ua if (umi_mode == "mi" and rng.random() < 0.85) else None      # MI:Z tag (bamutil.cpp:23-38); a pair without one falls back to its name
            qf = [rng.choice(QUALS) for _ in fseq]
            qr = [rng.choice(QUALS) for _ in rseq]
            if exotic:          # IUPAC codes (BAM nibbles outside A,C,G,T,N) and out-of-spec quals: generic-kernel paths
                fseq = "".join(rng.choice("MRWSYKVHDB=") if rng.random() < 0.02 else ch for ch in fseq)
                rseq = "".join(rng.choice("MRWSYKVHDB=") if rng.random() < 0.02 else ch for ch in rseq)
                qf = [rng.choice([128, 200, 255]) if rng.random() < 0.01 else x for x in qf]
                qr = [rng.choice([128, 200, 255]) if rng.random() < 0.01 else x for x in qr]
            if rng.random() < 0.7:                          # mostly-good reads so consensus paths vary
                qf = [37 if rng.random() < 0.8 else x for x in qf]
                qr = [37 if rng.random() < 0.8 else x for x in qr]
            f_flag, r_flag = (99, 147) if strand == 0 else (163, 83)
            nm_type = rng.choice(["C", "C", "C", "S", "i"])
            r1 = dict(qname=name, flag=f_flag, tid=tid, pos=fp, cigar=fcg, mtid=tid, mpos=rp_, isize=tlen, seq=fseq, qual=qf, nm=fnm, nm_type=nm_type)
            r2 = dict(qname=name, flag=r_flag, tid=tid, pos=rp_, cigar=rcg, mtid=tid, mpos=fp, isize=-tlen, seq=rseq, qual=qr, nm=rnm, nm_type=nm_type)
            if mi_tag is not None:
                r1["mi"] = mi_tag; r2["mi"] = mi_tag
            roll = rng.random()
            if cross:                                       # mate on another contig: each read clusters alone (negative right)
                r1.update(mtid=cross_tid, mpos=cross_pos, isize=0)
                recs.append(r1)
                if rng.random() < 0.5:
                    r3 = dict(r2); r3.update(tid=cross_tid, pos=cross_pos, mtid=tid, mpos=fp, isize=0, cigar="%dM" % len(rseq))
                    recs.append(r3)
                continue
            if roll < 0.04:                                 # mate unmapped: pass-through (gencore.cpp:307-309)
                r1.update(mtid=-1, mpos=-1, isize=0); recs.append(r1); continue
            if roll < 0.08:                                 # unmapped mate placed at its mate's coordinate, no CIGAR
                r1.update(isize=0, mpos=fp)
                r2.update(flag=r_flag | 4, pos=fp, cigar="*", mpos=fp, isize=0)
                recs += [r1, r2]; continue
            if roll < 0.11:                                 # isize == 0: no reference arbitration (group.cpp:363)
                r1.update(isize=0); r2.update(isize=0)
            if roll < 0.14:
                r2.update(flag=r_flag | 0x100)              # secondary: skipped
            if roll > 0.97:
                recs.append(dict(r1, flag=f_flag | 0x800))  # supplementary copy: skipped
            if 0.14 <= roll < 0.17:
                recs.append(r1); continue                   # lost mate
            if 0.17 <= roll < 0.19:
                recs.append(dict(r2))                       # third read with the same name in the cluster (replaces right)
            recs += [r1, r2]
    n_unmapped = rng.choice([0, 0, 3])
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    for k in range(n_unmapped):
        recs.append(dict(qname="UNM:%d" % k, flag=77, tid=-1, pos=-1, cigar="*", mtid=-1, mpos=-1, isize=0, seq="ACGTN" * 6, qual=[20] * 30, nm=None))
    batch = ReadBatch.from_records(recs)
    reference = []
    for i, s in enumerate(contigs):
        if i == 2 and rng.random() < 0.3:
            reference.append((None, 0))                      # contig missing from the FASTA (reference.cpp:46-53)
        else:
            reference.append((oracle_py.pack_reference(s), len(s)))
    over = dict(
        umi_prefix={"none": "", "prefix": "UMI", "colon": "", "duplex": "UMI", "mi": ""}[umi_mode],
        flush_period=period if period is not None else rng.choice([7, 23, 50, 200, 10000]),
        cluster_size_req=rng.choice([1, 1, 2, 3]),
        proper_umi_diff_threshold=rng.choice([0, 1, 1, 2]),
        duplex_mismatch_threshold=rng.choice([0, 2, 5]),
        score_percent_req=rng.choice([0.5, 0.8, 1.0]),
        base_score_req=rng.choice([6, 6, 1, 10]),
        duplex_only=1 if (umi_mode == "duplex" and rng.random() < 0.2) else 0,
        disable_duplex=1 if (umi_mode == "duplex" and rng.random() < 0.15) else 0,
        skip_low_complexity_cluster_threshold=rng.choice([1000, 1000, 4]),
    )
    if over["duplex_only"] and over["disable_duplex"]:
        over["disable_duplex"] = 0
    return batch, over, reference, contig_len


def make_params(over, contig_len):
    from gencore_amd.capi import default_params
    tl = np.asarray(contig_len, np.uint32)
    p = default_params(n_targets=len(tl), target_len=tl.ctypes.data, **over)
    p._keep = tl
    return p
