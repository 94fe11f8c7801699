"""Deterministic synthetic read-cluster generator — runs on CPU or on the GPU (torch tensors).

Counter-based RNG (splitmix64 finaliser over int64 tensors, wrap-around arithmetic), so the same (seed, config)
yields bit-identical data on CPU and on the MI355X.  Output is a coordinate-sorted stream in the gce_batch
struct-of-arrays layout plus the reference contigs in FastaReader's 4-bit code.

Workloads (BASELINE.json `configs`):
  cfg1s : plumbing substitute for configs[0] (2k pairs, depth 2, no UMI, 1 contig)
  cfg2  : 1 M pairs, 150 bp, no UMI, mean depth 4, single 10 Mb contig, -s 1
  cfg3  : 10 M pairs, 150 bp, 8 bp UMI (":UMI_XXXXXXXX"), mean CLUSTER depth 8 (molecule depth 10.4), -s 2; 24 hg19-shaped contigs (`scale` x hg19 lengths:
          0.1 = 300 Mb by default, bench.py uses 1.0 = 3.04 Gb); molecules concentrated on 5 k x 200 bp BED targets (the target
          count follows n_pairs so that a down-scaled stream keeps ~250 molecules per target)
  cfg4s : per-GPU eighth of configs[3] (12.5 M pairs, UMI, depth 16, whole-genome coverage of `scale` x hg19; the 8-GPU stream is
          n_pairs = 100 M at scale = 1.0, cut into key ranges by generate(shard=(rank, world)))
  cfg5  : ultra-deep hotspots, 250 bp, duplex UMIs AAAA_BBBB, depth U[500,2000]
Every size can be scaled with `n_pairs=`.
"""
import math

import numpy as np
import torch

from .batch import ReadBatch
from .capi import CORE_DTYPE

SEED0 = 0x67656E63  # "genc"
_REF_CACHE = {}      # (seed, contig lengths) -> (packed contigs, base codes): host tensors, at most two genomes (generate())
_M64 = (1 << 64) - 1


def _s64(x):
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


_K1, _K2, _K3 = _s64(0x9E3779B97F4A7C15), _s64(0xBF58476D1CE4E5B9), _s64(0x94D049BB133111EB)


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix(x):
    x = (x ^ _lsr(x, 30)) * _K2
    x = (x ^ _lsr(x, 27)) * _K3
    return x ^ _lsr(x, 31)


def rnd(seed, stream, idx):
    """64 random bits per element of idx (int64 tensor); pure function of (seed, stream, idx)."""
    return _mix(idx * _K1 + _s64((seed * 0x100000001B3 + stream * 0xD6E8FEB86659FD93) & _M64))


def rnd_u(seed, stream, idx, n):
    """uniform integer in [0, n)."""
    return (_lsr(rnd(seed, stream, idx), 11) % n)


def rnd_f(seed, stream, idx):
    return _lsr(rnd(seed, stream, idx), 11).to(torch.float64) * (1.0 / (1 << 53))


CONFIGS = {
    "cfg1s": dict(n_pairs=2000, L=150, umi=0, duplex=False, mean_depth=2, contigs=[200_000], ins_mu=300, ins_sd=30, ins_max=600,
                  supporting_reads=1),
    "cfg2": dict(n_pairs=1_000_000, L=150, umi=0, duplex=False, mean_depth=4, contigs=[10_000_000], ins_mu=300, ins_sd=30,
                 ins_max=600, supporting_reads=1),
    "cfg3": dict(n_pairs=10_000_000, L=150, umi=8, duplex=False, mean_depth=10.4, scale=0.1,   # molecules of 10.4 reads -> CLUSTERS of 8.0 pairs (soft clips, indels and UMI errors split molecules)
                 contigs=[w * 1_000_000 for w in (249, 243, 198, 191, 181, 171, 159, 146, 141, 136, 135, 134, 115, 107, 103,
                                                  90, 81, 78, 59, 63, 48, 51, 155, 59)],      # hg19 chr1..22, X, Y (Mb)
                 bed_targets=5000, bed_len=200, ins_mu=300, ins_sd=30, ins_max=600, supporting_reads=2),
    "cfg4s": dict(n_pairs=12_500_000, L=150, umi=8, duplex=False, mean_depth=16, scale=0.125,      # one GPU's eighth of configs[3]:
                  contigs=[w * 1_000_000 for w in (249, 243, 198, 191, 181, 171, 159, 146, 141, 136, 135, 134, 115, 107, 103,
                                                   90, 81, 78, 59, 63, 48, 51, 155, 59)],          # hg19-shaped, whole-genome coverage;
                  ins_mu=300, ins_sd=30, ins_max=600, supporting_reads=2),                         # bench.py scales the genome with the GPUs
    "cfg5": dict(n_pairs=0, n_molecules=1250, L=250, umi=4, duplex=True, depth_lo=500, depth_hi=2000, contigs=[50_000_000],
                 ins_mu=450, ins_sd=50, ins_max=900, supporting_reads=1, shard_mode="lpt"),
}


def _poisson_table(lam, kmax=96):
    """CDF of 1 + Poisson(lam) for inverse-transform sampling."""
    p = [math.exp(-lam)]
    for k in range(1, kmax):
        p.append(p[-1] * lam / k)
    return np.cumsum(p)


class SynthData:
    """tensors: dict name -> torch tensor (gce_batch fields, `core` as int32 [n,8]); reference: list of (nibbles, n_bases)."""

    def __init__(self, tensors, reference, target_len, cfg, info):
        self.t, self.reference, self.target_len, self.cfg, self.info = tensors, reference, target_len, cfg, info

    @property
    def n_reads(self):
        return int(self.t["core"].shape[0])

    def to_batch(self):
        t = {k: v.cpu().numpy() for k, v in self.t.items()}
        core = np.ascontiguousarray(t["core"]).view(CORE_DTYPE).reshape(-1)
        return ReadBatch(core=core, qname_off=t["qname_off"].astype(np.uint64), qname=t["qname"],
                         cigar_off=t["cigar_off"].astype(np.uint64), cigar=t["cigar"].view(np.uint32),
                         seq_off=t["seq_off"].astype(np.uint64), seq=t["seq"], qual_off=t["qual_off"].astype(np.uint64),
                         qual=t["qual"], nm=t["nm"], nm_type=t["nm_type"], mi_off=None, mi=None)

    def reference_host(self):
        return [(r.cpu().numpy(), n) for r, n in self.reference]


def generate(name="cfg2", n_pairs=None, seed=0, device="cpu", chunk_reads=1 << 21, align=1, shard=None, flush_period=10000, legacy_shard=False, **over):
    """align: start every read's seq / qual slice on a multiple of `align` bytes (the gce_batch offsets are free-form;
    an aligned layout lets the kernels' dword accesses stay inside cache lines).
    shard=(rank, world): plan the WHOLE stream (molecules, pairs, the sorted order of all reads: cheap), cut it into `world` ranges
    of the cluster key (tid, left) with equal read counts — the cuts fall inside contigs — and materialise only this rank's reads
    (bases, qualities, names).  The result carries `stream_context`: every read's global tick (device tensor) and the flush events
    of the whole stream (gce_batch.tick / gce_set_flush_events), and `global_index`, the reads' positions in the whole stream.
    The records are byte-identical to those of the unsharded stream."""
    cfg = dict(CONFIGS[name])
    cfg.update(over)
    if n_pairs is not None:
        cfg["n_pairs"] = int(n_pairs)
    seed = SEED0 + 1000003 * list(CONFIGS).index(name) + seed
    dev = torch.device(device)
    L = cfg["L"]
    i64 = dict(dtype=torch.int64, device=dev)
    contigs = [max(2000, int(c * cfg.get("scale", 1.0))) for c in cfg["contigs"]]
    ncont = len(contigs)

    # ---------------------------------------------------------------- molecules and their duplicate counts
    if cfg.get("duplex") and cfg.get("n_molecules"):
        M = int(cfg["n_molecules"]) if n_pairs is None else max(1, int(n_pairs) // ((cfg["depth_lo"] + cfg["depth_hi"]) // 2))
        mid = torch.arange(M, **i64)
        depth = cfg["depth_lo"] + rnd_u(seed, 1, mid, cfg["depth_hi"] - cfg["depth_lo"] + 1)
    else:
        M = max(1, int(round(cfg["n_pairs"] / cfg["mean_depth"])))
        mid = torch.arange(M, **i64)
        cdf = torch.tensor(_poisson_table(cfg["mean_depth"] - 1.0), dtype=torch.float64, device=dev)
        depth = 1 + torch.searchsorted(cdf, rnd_f(seed, 1, mid)).to(torch.int64)
    P = int(depth.sum().item())                              # total pairs (~ n_pairs)
    clen = torch.tensor(contigs, **i64)
    cum = torch.cumsum(clen, 0)
    total = int(cum[-1].item())
    # insert size ~ N(mu, sd) via Irwin-Hall(12), clipped to [L, ins_max]
    z = sum(rnd_f(seed, 10 + k, mid) for k in range(12)) - 6.0
    ins = torch.clamp((cfg["ins_mu"] + cfg["ins_sd"] * z).round().to(torch.int64), L, cfg["ins_max"])
    bed = None
    if cfg.get("bed_targets"):
        # capture-panel shape: T targets of bed_len bases at random places of the genome; a molecule picks a target
        # and starts where it still overlaps it.  ~250 molecules per target => clusters with several molecules occur naturally.
        T = max(1, int(round(cfg["bed_targets"] * cfg["n_pairs"] / float(CONFIGS[name]["n_pairs"]))))
        tix = torch.arange(T, **i64)
        tg = rnd_u(seed, 40, tix, total - 2000) + 1000           # target start, genome-linear
        t_of = rnd_u(seed, 41, mid, T)
        g = tg[t_of] - (L - 1) + rnd_u(seed, 2, mid, cfg["bed_len"] + L - 1)
        bed = (tg, cfg["bed_len"])
    else:
        g = rnd_u(seed, 2, mid, total)                       # global start, then mapped to (contig, offset)
    m_tid = torch.searchsorted(cum, g, right=True)
    m_start = g - (cum[m_tid] - clen[m_tid])
    m_start = torch.minimum(m_start, clen[m_tid] - ins - 16).clamp_(min=8)
    m_strand = rnd_u(seed, 3, mid, 2)
    U = int(cfg["umi"])
    Utot = 2 * U if cfg["duplex"] else U

    # ---------------------------------------------------------------- duplicates (pairs): everything about a pair is a pure function of its GLOBAL id
    cdepth = torch.cumsum(depth, 0)                          # pair id -> molecule: the first molecule whose cumulative depth exceeds it

    def ref_span(kind, k):
        return torch.where(kind == 0, torch.full_like(k, L), torch.where((kind == 1) | (kind == 2), L - k, torch.where(kind == 3, L - k, L + k)))

    def pair_level(ids, light=False):
        """pair-level fields of the pairs `ids` (ascending global pair ids); light: only (tid, forward position, reverse position)"""
        mol_ = torch.searchsorted(cdepth, ids, right=True)
        p_tid_, p_start_, p_ins_ = m_tid[mol_], m_start[mol_], ins[mol_]
        # alignment variant per read: 0 = LM (97%), 1 = leading S, 2 = trailing S (2%), 3 = I, 4 = D (1%)
        def variant(stream):
            r = rnd_u(seed, stream, ids, 1000)
            kind = torch.zeros_like(ids)
            kind = torch.where(r >= 970, 1 + (r & 1), kind)
            kind = torch.where(r >= 990, 3 + (r & 1), kind)
            k = torch.where((kind == 1) | (kind == 2), 3 + rnd_u(seed, stream + 1, ids, 8), 1 + rnd_u(seed, stream + 1, ids, 3))
            k = torch.where(kind == 0, torch.zeros_like(k), k)
            a = None if light else 20 + rnd_u(seed, stream + 2, ids, L - 60)       # split point for I/D
            return kind, k, a
        fk_, fklen_, fa_ = variant(20)
        rk_, rklen_, ra_ = variant(30)
        f_pos_ = p_start_ + torch.where(fk_ == 1, fklen_, torch.zeros_like(fklen_))
        r0_ = p_start_ + p_ins_ - L
        r_pos_ = r0_ + torch.where(rk_ == 1, rklen_, torch.zeros_like(rklen_))
        if light:
            return p_tid_, f_pos_, r_pos_
        strand_ = rnd_u(seed, 4, ids, 2) if cfg["duplex"] else m_strand[mol_]
        f_end = f_pos_ + ref_span(fk_, fklen_)
        r_end = r_pos_ + ref_span(rk_, rklen_)
        tlen = torch.maximum(f_end, r_end) - torch.minimum(f_pos_, r_pos_)
        f_isize_ = torch.where(f_pos_ <= r_pos_, tlen, -tlen)
        return dict(pid=ids, mol=mol_, p_tid=p_tid_, p_start=p_start_, p_ins=p_ins_, strand=strand_, fk=fk_, fklen=fklen_, fa=fa_, rk=rk_, rklen=rklen_, ra=ra_,
                    f_pos=f_pos_, r_pos=r_pos_, r0=r0_, f_isize=f_isize_, r_isize=-f_isize_)

    lean = shard is not None and not legacy_shard and cfg.get("shard_mode", "range") == "range"
    stream_context = global_index = None
    n_pairs_stream = P
    if lean:
        # Key-range shard WITHOUT the whole stream at pair level in memory (round 6; the legacy path below holds ~25 int64 arrays of all pairs: 48 GB and 2 s at 8 x 10 M pairs):
        # the stream is walked in chunks of pairs for (tid, forward position, reverse position) alone -- 9 bytes per pair kept --, ONE stable sort of the reads' packed
        # (tid, pos, tie) keys gives the stream order (the legacy path's two stable argsorts order by exactly that: key, then tie, then forward-before-reverse and pair id),
        # and everything else is computed for this rank's pairs only.  The records are byte-identical (tests/test_host_logic.py).
        rank, world = shard
        assert max(contigs) < (1 << 28) and ncont < 32
        CH = 1 << 23
        t8 = torch.empty(P, dtype=torch.uint8, device=dev); fp32 = torch.empty(P, dtype=torch.int32, device=dev); rp32 = torch.empty(P, dtype=torch.int32, device=dev)
        for a0 in range(0, P, CH):
            ids = torch.arange(a0, min(P, a0 + CH), **i64)
            tt, ff, rr = pair_level(ids, light=True)
            t8[a0:a0 + ids.numel()] = tt.to(torch.uint8); fp32[a0:a0 + ids.numel()] = ff.to(torch.int32); rp32[a0:a0 + ids.numel()] = rr.to(torch.int32)
            del ids, tt, ff, rr
        N_all = 2 * P
        key2 = torch.empty(N_all, **i64)
        for a0 in range(0, P, CH):
            b0 = min(P, a0 + CH)
            ids = torch.arange(a0, b0, **i64)
            tid_c = t8[a0:b0].to(torch.int64) << 28
            key2[a0:b0] = ((tid_c | fp32[a0:b0].to(torch.int64)) << 20) | rnd_u(seed, 5, ids * 2, 1 << 20)
            key2[P + a0:P + b0] = ((tid_c | rp32[a0:b0].to(torch.int64)) << 20) | rnd_u(seed, 5, ids * 2 + 1, 1 << 20)
            del ids, tid_c
        order = torch.sort(key2, stable=True).indices        # original index = rev * P + pair: ties of (key, tie) keep forward reads first, then the pair id
        del key2
        is_rev_all = order >= P
        rd_pair_all = order - is_rev_all.to(torch.int64) * P
        del order
        evi = torch.arange(flush_period - 1, N_all, flush_period, **i64) if N_all >= flush_period else torch.zeros(0, **i64)
        ev_pair, ev_rev = rd_pair_all[evi], is_rev_all[evi]
        ev_pos = torch.where(ev_rev, rp32[ev_pair], fp32[ev_pair]).to(torch.int64)
        ev_tid = t8[ev_pair].to(torch.int64)
        kp = (t8.to(torch.int64) << 40) | torch.minimum(fp32, rp32).to(torch.int64)
        srt = torch.sort(kp).values
        cuts = srt[torch.tensor([min(P - 1, (P * r) // world) for r in range(1, world)], **i64)] if world > 1 else srt[:0]
        del srt
        mine_p = torch.searchsorted(cuts, kp, right=True) == rank
        del kp, t8, fp32, rp32
        sel = torch.nonzero(mine_p[rd_pair_all]).squeeze(1)          # my reads, in stream order
        loc = torch.cumsum(mine_p.to(torch.int64), 0) - 1            # global pair id -> local pair id
        rd_pair, rd_rev = loc[rd_pair_all[sel]], is_rev_all[sel].to(torch.int64)
        del rd_pair_all, is_rev_all, loc
        global_index = sel
        stream_context = dict(tick=(sel + 1).contiguous(), ev_tid=ev_tid.to(torch.int32).cpu().numpy(), ev_pos=ev_pos.to(torch.int32).cpu().numpy())
        d_ = pair_level(torch.nonzero(mine_p).squeeze(1))
        del mine_p
        P = int(d_["pid"].numel()); N = int(sel.numel())
    else:
        d_ = pair_level(torch.arange(P, **i64))
    pid, mol, p_tid, p_start, p_ins, strand = d_["pid"], d_["mol"], d_["p_tid"], d_["p_start"], d_["p_ins"], d_["strand"]
    fk, fklen, fa, rk, rklen, ra = d_["fk"], d_["fklen"], d_["fa"], d_["rk"], d_["rklen"], d_["ra"]
    f_pos, r_pos, r0, f_isize, r_isize = d_["f_pos"], d_["r_pos"], d_["r0"], d_["f_isize"], d_["r_isize"]
    del d_

    # ---------------------------------------------------------------- reads = 2 per pair, sorted by (tid, pos)
    if not lean:
        N = 2 * P
        rd_pair = torch.cat([pid, pid])
        rd_rev = torch.cat([torch.zeros(P, **i64), torch.ones(P, **i64)])
        rd_pos = torch.cat([f_pos, r_pos])
        rd_tid = torch.cat([p_tid, p_tid])
        # tie order inside one (tid,pos): a scrambled pair id, like an aligner's arbitrary order
        key = (rd_tid << 40) | rd_pos
        tie = rnd_u(seed, 5, rd_pair * 2 + rd_rev, 1 << 20)
        order = torch.argsort(tie, stable=True)
        order = order[torch.argsort(key[order], stable=True)]
        rd_pair, rd_rev = rd_pair[order], rd_rev[order]
        del order, tie, rd_pos, rd_tid
        if shard is not None:
            rank, world = shard
            # every read of this generator reaches the cluster map, so a read's global tick is its place in the stream (gencore.cpp:319);
            # flush events = the reads on which tick % period == 0 (gencore.cpp:321-322)
            evi = torch.arange(flush_period - 1, N, flush_period, **i64) if N >= flush_period else torch.zeros(0, **i64)
            ev_pair, ev_rev = rd_pair[evi], rd_rev[evi]
            ev_pos = torch.where(ev_rev == 1, r_pos[ev_pair], f_pos[ev_pair])
            ev_tid = p_tid[ev_pair]
            # cluster key of a pair: (tid, left) with left = the leftmost of the two mates (the read with isize < 0 follows its mate,
            # gencore.cpp:301-303); key ranges with equal pair counts
            kp = (p_tid << 40) | torch.minimum(f_pos, r_pos)
            if cfg.get("shard_mode", "range") == "lpt":
                # ultra-deep hotspots: whole clusters dealt to the least loaded rank, heaviest first, weight = depth^2
                uk, inv, cnt = torch.unique(kp, return_inverse=True, return_counts=True)
                cn = cnt.cpu().numpy().astype(np.float64)
                load, owner = np.zeros(world), np.zeros(len(cn), np.int64)
                for c in np.argsort(-cn ** 2, kind="stable"):
                    r = int(np.argmin(load)); owner[c] = r; load[r] += cn[c] ** 2
                mine_p = torch.from_numpy(owner).to(dev)[inv] == rank
                del uk, inv, cnt
            else:
                srt = torch.sort(kp).values
                cuts = srt[torch.tensor([min(P - 1, (P * r) // world) for r in range(1, world)], **i64)] if world > 1 else srt[:0]
                mine_p = torch.searchsorted(cuts, kp, right=True) == rank
                del srt
            del kp
            sel = torch.nonzero(mine_p[rd_pair]).squeeze(1)              # my reads, in stream order
            loc = torch.cumsum(mine_p.to(torch.int64), 0) - 1            # global pair id -> local pair id
            rd_pair, rd_rev = loc[rd_pair[sel]], rd_rev[sel]
            global_index = sel
            stream_context = dict(tick=(sel + 1).contiguous(), ev_tid=ev_tid.to(torch.int32).cpu().numpy(), ev_pos=ev_pos.to(torch.int32).cpu().numpy())
            # pair-level arrays restricted to my pairs; `pid` keeps the GLOBAL ids: they key the per-base / per-name random streams
            pid, mol, p_tid, p_start, p_ins, strand = pid[mine_p], mol[mine_p], p_tid[mine_p], p_start[mine_p], p_ins[mine_p], strand[mine_p]
            fk, fklen, fa, rk, rklen, ra = fk[mine_p], fklen[mine_p], fa[mine_p], rk[mine_p], rklen[mine_p], ra[mine_p]
            f_pos, r_pos, r0, f_isize, r_isize = f_pos[mine_p], r_pos[mine_p], r0[mine_p], f_isize[mine_p], r_isize[mine_p]
            P = int(pid.numel())
            N = int(sel.numel())
            del mine_p, loc, sel
    isrev = rd_rev == 1
    pos = torch.where(isrev, r_pos[rd_pair], f_pos[rd_pair])
    mpos = torch.where(isrev, f_pos[rd_pair], r_pos[rd_pair])
    tid = p_tid[rd_pair]
    isize = torch.where(isrev, r_isize[rd_pair], f_isize[rd_pair])
    kind = torch.where(isrev, rk[rd_pair], fk[rd_pair])
    klen = torch.where(isrev, rklen[rd_pair], fklen[rd_pair])
    asplit = torch.where(isrev, ra[rd_pair], fa[rd_pair])
    st = strand[rd_pair]
    # flags: top strand -> read1 forward (99) / read2 reverse (147); bottom strand -> read1 reverse (83) / read2 forward (163)
    flag = torch.where(st == 0, torch.where(isrev, torch.full_like(st, 147), torch.full_like(st, 99)),
                       torch.where(isrev, torch.full_like(st, 83), torch.full_like(st, 163)))
    # query start on the reference for column 0 of the M block(s)
    ref0 = torch.where(isrev, r0[rd_pair], p_start[rd_pair])

    # CIGAR: up to 3 ops
    M_, I_, D_, S_ = 0, 1, 2, 4
    ncig = torch.where(kind == 0, torch.ones_like(kind), torch.where(kind <= 2, torch.full_like(kind, 2), torch.full_like(kind, 3)))
    w0 = torch.where(kind == 0, torch.full_like(kind, (L << 4) | M_),
                     torch.where(kind == 1, (klen << 4) | S_, torch.where(kind == 2, ((L - klen) << 4) | M_, (asplit << 4) | M_)))
    w1 = torch.where(kind == 1, ((L - klen) << 4) | M_, torch.where(kind == 2, (klen << 4) | S_,
                     torch.where(kind == 3, (klen << 4) | I_, (klen << 4) | D_)))
    w2 = torch.where(kind == 3, ((L - asplit - klen) << 4) | M_, ((L - asplit) << 4) | M_)
    cigar_off = torch.cumsum(ncig, 0) - ncig
    cigar = torch.zeros(int(ncig.sum().item()), **i64)
    cigar[cigar_off] = w0
    m1 = ncig >= 2
    cigar[cigar_off[m1] + 1] = w1[m1]
    m2 = ncig >= 3
    cigar[cigar_off[m2] + 2] = w2[m2]

    # ---------------------------------------------------------------- reference contigs (random ACGT), FASTA 4-bit code
    # (the genome is a pure function of (seed, contig lengths): a test session that asks for the same workload twenty times -- 300 Mb at the default scale, seconds
    #  of CPU each -- gets it from a small cache; host tensors only, never the GPU-resident genomes of bench.py.  Nobody writes to these tensors.)
    ref_key = (seed, tuple(contigs))
    cached = _REF_CACHE.get(ref_key) if dev.type == "cpu" else None
    if cached is not None:
        reference, ref_all = cached
    else:
        reference, ref_codes = [], []
        fcode = torch.tensor([1, 3, 4, 2], dtype=torch.uint8, device=dev)     # our base index 0..3 = A,C,G,T -> FastaReader code
        for ci, ln in enumerate(contigs):
            b = (rnd(seed, 100 + ci, torch.arange(ln, **i64)) & 3).to(torch.uint8)   # 0..3 = A,C,G,T
            ref_codes.append(b)
            fc = fcode[b.long()]
            if ln % 2:
                fc = torch.cat([fc, torch.zeros(1, dtype=torch.uint8, device=dev)])
            reference.append(((fc[0::2] | (fc[1::2] << 4)).contiguous(), ln))
        ref_all = torch.cat(ref_codes)
        del ref_codes
        if dev.type == "cpu" and sum(contigs) <= 400_000_000:
            while len(_REF_CACHE) >= 2:
                _REF_CACHE.pop(next(iter(_REF_CACHE)))
            _REF_CACHE[ref_key] = (reference, ref_all)
    ref_base_off = torch.tensor([0] + list(np.cumsum(contigs)[:-1]), **i64)

    # ---------------------------------------------------------------- per-base data, chunked over reads
    SB = (L + 1) // 2
    SBs = (SB + align - 1) // align * align          # strides of the seq / qual slices
    Ls = (L + align - 1) // align * align
    seq = torch.zeros(N * SBs, dtype=torch.uint8, device=dev)
    qual = torch.zeros(N * Ls, dtype=torch.uint8, device=dev)
    nm = torch.empty(N, dtype=torch.int32, device=dev)
    bam_nib = torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=dev)
    j = torch.arange(L, **i64).unsqueeze(0)
    err_thr = int(0.005 * (1 << 20))
    for s in range(0, N, chunk_reads):
        e = min(N, s + chunk_reads)
        kd, kl, a = kind[s:e, None], klen[s:e, None], asplit[s:e, None]
        # reference offset of query column j (or -1 for inserted / clipped bases)
        roff = torch.where(kd == 0, j.expand(e - s, L),
               torch.where(kd == 1, torch.where(j >= kl, j, torch.full_like(j, -1)),    # ref0 is the unclipped start
               torch.where(kd == 2, torch.where(j < L - kl, j, torch.full_like(j, -1)),
               torch.where(kd == 3, torch.where(j < a, j, torch.where(j < a + kl, torch.full_like(j, -1), j - kl)),
                           torch.where(j < a, j, j + kl)))))
        rid = torch.arange(s, e, **i64)
        # per-base randomness must belong to the MOLECULE's duplicate, not the sorted position: key on (pair, mate)
        bkey = ((pid[rd_pair[s:e]] * 2 + rd_rev[s:e]) * 1024)[:, None] + j
        h = rnd(seed, 6, bkey)
        gpos = (ref_base_off[tid[s:e]] + ref0[s:e])[:, None] + roff.clamp(min=0)
        rb = ref_all[gpos.clamp(max=ref_all.numel() - 1)].long()
        is_err = (h & 0xFFFFF) < err_thr
        sub = (rb + 1 + (_lsr(h, 20) & 0xFF) % 3) & 3
        rand_b = _lsr(h, 28) & 3
        base = torch.where(roff < 0, rand_b, torch.where(is_err, sub, rb))
        qsel = _lsr(h, 32) % 100
        q = torch.where(qsel < 80, torch.full_like(qsel, 37), torch.where(qsel < 92, torch.full_like(qsel, 25), torch.full_like(qsel, 11)))
        mism = ((roff >= 0) & (base != rb)).sum(1)
        nm[s:e] = (mism + torch.where((kind[s:e] == 3) | (kind[s:e] == 4), klen[s:e], torch.zeros_like(mism))).to(torch.int32)
        nibs = bam_nib[base]
        if L % 2:
            nibs = torch.cat([nibs, torch.zeros(e - s, 1, dtype=torch.uint8, device=dev)], 1)
        seq.view(N, SBs)[s:e, :SB] = (nibs[:, 0::2] << 4) | nibs[:, 1::2]
        qual.view(N, Ls)[s:e, :L] = q.to(torch.uint8)
        del roff, h, gpos, rb, base, nibs, q, bkey, rid

    # ---------------------------------------------------------------- qnames: SIM:<lane>:<tile>:<x>:<y>[:UMI_<umi>]
    W = 4 + 2 + 5 + 6 + 6 + (5 + Utot + (1 if cfg["duplex"] else 0) if U else 0) + 1
    scr = (pid * 0x9E3779B1 + 12345) & 0xFFFFFFFF            # bijection on 32 bits => unique (x, y)
    lane = 1 + (scr % 8)
    tile = 1101 + ((scr >> 3) % 96)
    x = (scr >> 16) & 0xFFFF
    y = scr & 0xFFFF
    cols = []
    SKIP = 255

    def lit(sx):
        return [torch.full((P,), ord(ch), dtype=torch.uint8, device=dev) for ch in sx]

    def digits(v, width):
        out, started = [], torch.zeros(P, dtype=torch.bool, device=dev)
        for k in range(width - 1, -1, -1):
            d = (v // (10 ** k)) % 10
            started = started | (d > 0) | (k == 0)
            out.append(torch.where(started, (48 + d).to(torch.uint8), torch.full((P,), SKIP, dtype=torch.uint8, device=dev)))
        return out
    cols += lit("SIM:") + digits(lane, 1) + lit(":") + digits(tile, 4) + lit(":") + digits(x, 5) + lit(":") + digits(y, 5)
    if U:
        ub = torch.tensor([ord(ch) for ch in "ACGT"], dtype=torch.uint8, device=dev)
        cols += lit(":UMI_")
        um = (rnd(seed, 7, mol[:, None] * 64 + torch.arange(Utot, **i64)[None, :]) & 3)
        ue = rnd(seed, 8, pid[:, None] * 64 + torch.arange(Utot, **i64)[None, :])
        uerr = (ue & 0xFFFF) < int(0.01 * 65536)
        um = torch.where(uerr, (um + 1 + (_lsr(ue, 16) & 0xFF) % 3) & 3, um)
        if cfg["duplex"]:
            # top strand carries A_B, bottom strand B_A
            A, B = um[:, :U], um[:, U:]
            first = torch.where((strand == 0)[:, None], A, B)
            second = torch.where((strand == 0)[:, None], B, A)
            cols += [ub[first[:, k]] for k in range(U)] + lit("_") + [ub[second[:, k]] for k in range(U)]
        else:
            cols += [ub[um[:, k]] for k in range(U)]
    cols += [torch.zeros(P, dtype=torch.uint8, device=dev)]
    mat = torch.stack(cols, 1)
    assert mat.shape[1] <= W + 8
    keepm = mat != SKIP
    plen = keepm.sum(1)
    # per READ names (both mates carry the same name), laid out in sorted read order
    rlen = plen[rd_pair]
    qname_off = torch.cumsum(rlen, 0) - rlen
    rmat = mat[rd_pair]
    qname = rmat[rmat != SKIP].contiguous()
    del rmat, mat

    core = torch.zeros(N, 8, dtype=torch.int32, device=dev)
    core[:, 0] = tid.to(torch.int32)
    core[:, 1] = pos.to(torch.int32)
    # word 2: l_qname u8 | mapq u8 << 8 | bin u16 << 16 ; word 3: n_cigar u16 | flag u16 << 16
    core[:, 2] = (rlen | (60 << 8)).to(torch.int32)
    core[:, 3] = (ncig | (flag << 16)).to(torch.int32)
    core[:, 4] = L
    core[:, 5] = tid.to(torch.int32)
    core[:, 6] = mpos.to(torch.int32)
    core[:, 7] = isize.to(torch.int32)

    rid = torch.arange(N, **i64)
    tensors = dict(core=core, qname_off=qname_off, qname=qname, cigar_off=cigar_off, cigar=cigar.to(torch.int32),
                   seq_off=rid * SBs, seq=seq, qual_off=rid * Ls, qual=qual, nm=nm,
                   nm_type=torch.full((N,), ord("C"), dtype=torch.uint8, device=dev))
    info = dict(name=name, n_pairs=P, n_reads=N, n_pairs_stream=n_pairs_stream, n_molecules=M, read_len=L, umi_len=Utot,
                umi_prefix="UMI" if U else "", supporting_reads=cfg["supporting_reads"], genome_bases=total,
                bed_targets=(int(bed[0].numel()) if bed else 0))
    if bed:                                                  # BED regions as (tid, start, end), sorted
        b_tid = torch.searchsorted(cum, bed[0], right=True)
        b_start = bed[0] - (cum[b_tid] - clen[b_tid])
        b_start = torch.minimum(b_start, clen[b_tid] - bed[1]).clamp_(min=0)
        order_b = torch.argsort(b_tid * (1 << 40) + b_start)
        info["bed"] = torch.stack([b_tid[order_b], b_start[order_b], b_start[order_b] + bed[1]], 1).cpu().numpy()
    out = SynthData(tensors, reference, contigs, cfg, info)
    out.stream_context, out.global_index = stream_context, global_index
    return out
