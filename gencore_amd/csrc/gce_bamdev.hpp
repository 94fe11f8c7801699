// gce_bamdev.hpp — the GPU side of the BAM codec ("multi-threaded or GPU-assisted"): the host only inflates and deflates
// BGZF blocks; everything that touches RECORDS happens in HBM.
//   gce_raw_begin / gce_raw_push     the inflated BAM stream arrives window by window (host pinned buffer -> HBM, asynchronously, while the host
//                                    inflates the next window); the host keeps nothing of it
//   gce_raw_finish                   record index on the GPU (the chain of block_size fields, walked by one lane per 16 KB segment from a
//                                    guessed record start and joined where the guesses meet the chain -- the scheme of the host index in
//                                    bamio.cpp --), then the gce_batch of the stream WITHOUT copying names, bases or qualities: the blobs of
//                                    the struct-of-arrays ARE the raw stream (offsets point into it); only the 32-byte key records, the CIGAR
//                                    words (alignment) and NM / MI (aux walk) are extracted.  Replaces sam_read1's record parsing
//                                    (src/gencore.cpp:205-274 via htslib) for the whole file at once.
//   gce_raw_build_output             after gce_process: the emitted records as BAM records, assembled in HBM from the raw stream (the template's
//                                    record, mutated in place by the vote, with the name of its copyQName source, NM patched, FR / RR
//                                    appended: src/gencore.cpp:83-111 writeBam, src/pair.cpp:43-68) -- the host gets a byte stream that only
//                                    needs BGZF blocks around it
//   gce_raw_read_output[_async]      that stream, piece by piece, into host (pinned) memory
#pragma once

namespace {

typedef uint32_t rb_u32u __attribute__((aligned(1)));
typedef uint16_t rb_u16u __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t rb32(const uint8_t *p) { return *(const rb_u32u *)p; }
__device__ __forceinline__ uint32_t rb16(const uint8_t *p) { return *(const rb_u16u *)p; }
#define RAW_SEG (16u << 10)

// does a record start at o?  (bamio.cpp's test: sane block_size, contig ids inside the header's, a NUL-terminated name, the fixed fields fit)
__device__ __forceinline__ bool raw_plausible(const uint8_t *u, uint64_t o, uint64_t n, int32_t nref) {
    if (o + 36 > n) return false;
    const uint32_t bs = rb32(u + o);
    if (bs < 32 || bs > (1u << 28) || o + 4 + bs > n) return false;
    const uint8_t *r = u + o + 4;
    const int32_t tid = (int32_t)rb32(r), mtid = (int32_t)rb32(r + 20), ls = (int32_t)rb32(r + 16); const uint32_t lq = r[8], nc = rb16(r + 12);
    if (tid < -1 || tid >= nref || mtid < -1 || mtid >= nref || lq == 0 || ls < 0) return false;
    if (32ull + lq + 4ull * nc + (uint64_t)(ls + 1) / 2 + (uint64_t)ls > bs) return false;
    return r[32 + lq - 1] == 0;
}
// records starting in [o, hi): count, optionally their offsets; returns where the chain leaves the range (~0 = broken chain)
__device__ __forceinline__ uint64_t raw_walk(const uint8_t *u, uint64_t o, uint64_t hi, uint64_t n, uint32_t &cnt, uint64_t *out) {
    while (o < hi && o + 4 <= n) {
        const uint32_t bs = rb32(u + o);
        if (bs < 32 || o + 4 + bs > n) return ~0ull;
        if (out) out[cnt] = o;
        cnt++;
        o += 4ull + bs;
    }
    return o;
}
__global__ __launch_bounds__(256) void k_raw_seg(const uint8_t *u, uint64_t first, uint64_t n, int32_t nref, uint64_t nseg, uint64_t *guess, uint64_t *leave, uint32_t *cnt) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const uint64_t lo = first + s * RAW_SEG, hi = min(n, lo + RAW_SEG);
    uint64_t o = lo;
    if (s > 0) {                                                   // a guessed start: two sane records in a row (the first segment starts on the first record)
        while (o < hi && !(raw_plausible(u, o, n, nref) && (o + 4 + rb32(u + o) + 3 >= n || raw_plausible(u, o + 4 + rb32(u + o), n, nref)))) o++;
        if (o >= hi) { guess[s] = ~0ull; leave[s] = ~0ull; cnt[s] = 0; return; }
    }
    uint32_t c = 0;
    guess[s] = o;
    leave[s] = raw_walk(u, o, hi, n, c, nullptr);
    cnt[s] = c;
}
// every segment's guess must be where the chain of the segment in front of it leaves: flag = number of segments for which it is not
__global__ __launch_bounds__(256) void k_raw_check(const uint64_t *guess, const uint64_t *leave, uint64_t nseg, uint64_t n, unsigned int *bad, uint8_t *bad_of) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const bool b = leave[s] == ~0ull || (s > 0 && guess[s] != leave[s - 1]) || (s == nseg - 1 && leave[s] != n);
    bad_of[s] = b;
    if (b) atomicAdd(bad, 1u);
}
// repair, in parallel.  A guess can be a coincidence: one byte in front of a record of contig 0 the shifted fields pass the test about once
// in 4000 segments (block_size x 256 + the last NM byte, tid x 256 = 0 ...), and the chain walked from there leaves far behind the
// segment, which also puts the NEXT segment off the chain although its own guess is right.  Every round re-walks the flagged segments
// whose predecessor is NOT flagged (that one's chain is final, nothing it reads changes in the round) from where that chain leaves; the
// check that follows clears the neighbours.  Rounds = the longest run of truly wrong segments (records longer than a segment).
__global__ __launch_bounds__(256) void k_raw_fix(const uint8_t *u, uint64_t first, uint64_t n, uint64_t nseg, uint64_t *guess, uint64_t *leave, uint32_t *cnt, const uint8_t *bad_of, unsigned int *changed, unsigned int *broken) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg || s == 0 || !bad_of[s] || bad_of[s - 1]) return;
    const uint64_t at = leave[s - 1];
    if (at == ~0ull) return;
    const uint64_t hi = min(n, first + s * RAW_SEG + RAW_SEG);
    uint32_t c = 0; uint64_t x = at;
    if (at < hi) { x = raw_walk(u, at, hi, n, c, nullptr); if (x == ~0ull) { *broken = 1u; return; } }
    guess[s] = at; leave[s] = x; cnt[s] = c;
    *changed = 1u;
}
// the last resort (after several parallel rounds): ONE thread follows the chain from segment to segment
// and re-walks only the segments whose guess does not lie on it
__global__ void k_raw_repair(const uint8_t *u, uint64_t first, uint64_t n, uint64_t nseg, uint64_t *guess, uint64_t *leave, uint32_t *cnt, unsigned int *broken) {
    if (blockIdx.x || threadIdx.x) return;
    uint64_t at = first;
    for (uint64_t s = 0; s < nseg; s++) {
        const uint64_t lo = first + s * RAW_SEG, hi = min(n, lo + RAW_SEG);
        if (at >= hi) { guess[s] = at; leave[s] = at; cnt[s] = 0; continue; }       // a record spans the whole segment
        if (guess[s] != at || leave[s] == ~0ull) {
            uint32_t c = 0;
            const uint64_t x = raw_walk(u, at, hi, n, c, nullptr);
            if (x == ~0ull) { *broken = 1u; return; }
            guess[s] = at; leave[s] = x; cnt[s] = c;
        }
        at = leave[s];
    }
    if (at != n) *broken = 1u;
}
__global__ __launch_bounds__(256) void k_raw_offsets(const uint8_t *u, uint64_t first, uint64_t n, uint64_t nseg, const uint64_t *guess, const uint64_t *base, uint64_t *rec_off) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const uint64_t hi = min(n, first + s * RAW_SEG + RAW_SEG);
    uint32_t c = 0;
    (void)raw_walk(u, guess[s], hi, n, c, rec_off + base[s]);
}
// bytes of an aux value behind its type byte, or ~0 (bamio.cpp aux_size)
__device__ __forceinline__ uint64_t raw_aux_size(uint8_t type, const uint8_t *p, const uint8_t *end) {
    switch (type) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    case 'Z': case 'H': { const uint8_t *q = p; while (q < end && *q) q++; return q < end ? (uint64_t)(q - p) + 1 : ~0ull; }
    case 'B': {
        if (p + 5 > end) return ~0ull;
        const uint8_t t = p[0];
        const uint64_t es = (t == 'c' || t == 'C') ? 1 : (t == 's' || t == 'S') ? 2 : (t == 'i' || t == 'I' || t == 'f') ? 4 : ~0ull;
        return es == ~0ull ? ~0ull : 5 + es * (uint64_t)rb32(p + 1);
    }
    default: return ~0ull;
    }
}
struct RawSoA {
    gce_core *core; uint64_t *qoff, *soff, *loff, *mioff; uint32_t *ncig, *nm_pos; int32_t *nm; uint8_t *nmt; unsigned int *have_mi;
    unsigned int *bad_rec; int32_t n_ref;      // first record whose fields do not fit its block_size (atomicMin; NONE32 = none)
};
// one thread per record: the 32-byte key record, the offsets of name / bases / qualities INSIDE the raw stream, NM (first match, value as
// bam_aux2i gives it) and MI:Z from the aux area (bamio.cpp scan_aux), the place of NM's value byte for the writer
__global__ __launch_bounds__(256) void k_raw_fill(const uint8_t *u, const uint64_t *rec_off, uint64_t n_rec, RawSoA o) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const uint64_t ro = rec_off[i];
    const uint8_t *r = u + ro + 4; const uint32_t bs = rb32(u + ro);
    union { gce_core c; uint32_t w[8]; uint4 q[2]; } t;
#pragma unroll
    for (int k = 0; k < 8; k++) t.w[k] = rb32(r + 4 * k);
    reinterpret_cast<uint4 *>(o.core + i)[0] = t.q[0]; reinterpret_cast<uint4 *>(o.core + i)[1] = t.q[1];
    const uint32_t lq = t.c.l_qname, nc = t.c.n_cigar; const int32_t ls = t.c.l_qseq;
    // The chain of block_size fields says where records start, not that their fields fit: the test gce_bam_open applies on the host
    // (bamio.cpp, "inconsistent record lengths") -- a name, l_seq >= 0, name + CIGAR + bases + qualities inside block_size, contig ids
    // inside the header's.  A record that fails it gets empty fields here (nothing downstream follows its lengths) and fails the stream.
    if (lq == 0 || ls < 0 || 32ull + lq + 4ull * nc + (uint64_t)(ls + 1) / 2 + (uint64_t)ls > bs || t.c.tid < -1 || t.c.tid >= o.n_ref || t.c.mtid < -1 || t.c.mtid >= o.n_ref) {
        atomicMin(o.bad_rec, (unsigned int)i);
        o.qoff[i] = ro + 4; o.soff[i] = ro + 4; o.loff[i] = ro + 4; o.ncig[i] = 0; o.nm[i] = 0; o.nmt[i] = 0; o.nm_pos[i] = 0; o.mioff[i] = ~0ull;
        return;
    }
    const uint64_t q0 = ro + 36, c0 = q0 + lq, s0 = c0 + 4ull * nc, l0 = s0 + (uint64_t)(ls + 1) / 2, a0 = l0 + (uint64_t)ls;
    o.qoff[i] = q0; o.soff[i] = s0; o.loff[i] = l0; o.ncig[i] = nc;
    uint8_t nmt = 0; int32_t nm = 0; uint32_t nm_pos = 0; uint64_t mi = ~0ull;
    const uint8_t *p = u + a0, *end = r + bs;
    while (p + 3 <= end) {
        const uint8_t t0 = p[0], t1 = p[1], ty = p[2]; const uint8_t *v = p + 3;
        const uint64_t sz = raw_aux_size(ty, v, end);
        if (sz == ~0ull || v + sz > end) break;
        if (t0 == 'N' && t1 == 'M' && nmt == 0) {
            nmt = ty; nm_pos = (uint32_t)(v - (u + ro));
            switch (ty) {
            case 'c': nm = (int8_t)v[0]; break;            case 'C': nm = v[0]; break;
            case 's': nm = (int16_t)rb16(v); break;        case 'S': nm = (int32_t)rb16(v); break;
            case 'i': nm = (int32_t)rb32(v); break;        case 'I': nm = (int32_t)rb32(v); break;
            default: nm = 0; break;
            }
        } else if (t0 == 'M' && t1 == 'I' && ty == 'Z' && mi == ~0ull) mi = (uint64_t)(v - u);
        p = v + sz;
    }
    o.nm[i] = nm; o.nmt[i] = nmt; o.nm_pos[i] = nm_pos; o.mioff[i] = mi;
    if (mi != ~0ull && *(volatile unsigned int *)o.have_mi == 0u) atomicOr(o.have_mi, 1u);
}
__global__ __launch_bounds__(256) void k_raw_cigar(const uint8_t *u, const uint64_t *rec_off, uint64_t n_rec, const uint64_t *coff, uint32_t *cigar) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const uint8_t *r = u + rec_off[i] + 4;
    const uint32_t lq = r[8], nc = (uint32_t)(coff[i + 1] - coff[i]);           // (the count k_raw_fill accepted, not the record's own field)
    const uint8_t *c = r + 32 + lq;
    uint32_t *dst = cigar + coff[i];
    for (uint32_t k = 0; k < nc; k++) dst[k] = rb32(c + 4 * k);
}

// ---- output records
struct RawOut { const uint32_t *src, *qname_src; const int32_t *nm_new; const int16_t *fr, *rr; };
__global__ __launch_bounds__(256) void k_rec_size(const uint8_t *u, const uint64_t *rec_off, RawOut r, uint64_t n_out, uint64_t *size) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    const uint64_t so = rec_off[r.src[k]], qo = rec_off[r.qname_src[k]];
    const uint32_t bs = rb32(u + so), lq_old = u[so + 12], lq_new = u[qo + 12];
    size[k] = 4ull + bs - lq_old + lq_new + (r.fr[k] >= 0 ? 4 : 0) + (r.rr[k] >= 0 ? 4 : 0);
}
// 16 lanes per record: [block_size][core, l_qname of the name's source][that name][everything behind the name of the template's own record:
// CIGAR, bases and qualities as the vote left them, aux][FR][RR]; NM's value byte patched (type 'C' only, checked by the vote)
__global__ __launch_bounds__(256) void k_rec_build(const uint8_t *u, const uint64_t *rec_off, const uint32_t *nm_pos, RawOut r, uint64_t n_out, const uint64_t *roff, uint8_t *body) {
    const int sub = threadIdx.x & 15;
    for (uint64_t k = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; k < n_out; k += ((uint64_t)gridDim.x * blockDim.x) >> 4) {
        const uint32_t src = r.src[k];
        const uint64_t so = rec_off[src], qo = rec_off[r.qname_src[k]];
        const uint32_t bs = rb32(u + so), lq_old = u[so + 12], lq_new = u[qo + 12];
        const int fr = r.fr[k], rr = r.rr[k];
        const uint32_t tail = bs - 32 - lq_old;                                    // CIGAR .. aux
        const uint32_t nbs = 32 + lq_new + tail + (fr >= 0 ? 4 : 0) + (rr >= 0 ? 4 : 0);
        uint8_t *d = body + roff[k];
        if (sub < 9) {                                                             // block_size + the eight words of the core
            uint32_t w = sub == 0 ? nbs : rb32(u + so + 4 * sub);
            if (sub == 3) w = (w & ~0xFFu) | lq_new;                               // l_read_name is the low byte of the third core word
            *(rb_u32u *)(d + 4 * sub) = w;
        }
        for (uint32_t j = sub; j < lq_new; j += 16) d[36 + j] = u[qo + 36 + j];
        const uint8_t *ts = u + so + 36 + lq_old; uint8_t *td = d + 36 + lq_new;
        for (uint32_t j = 4 * sub; j < tail; j += 64) {                            // four bytes per lane and trip
            if (j + 4 <= tail) *(rb_u32u *)(td + j) = rb32(ts + j);
            else for (uint32_t q = j; q < tail; q++) td[q] = ts[q];
        }
        if (sub == 15) {
            uint8_t *e2 = td + tail;
            if (fr >= 0) { e2[0] = 'F'; e2[1] = 'R'; e2[2] = 'C'; e2[3] = (uint8_t)fr; e2 += 4; }
            if (rr >= 0) { e2[0] = 'R'; e2[1] = 'R'; e2[2] = 'C'; e2[3] = (uint8_t)rr; }
        }
    }
}
// (after k_rec_build: one more pass so that the NM patch cannot race with the copy of the same byte by another lane)
__global__ __launch_bounds__(256) void k_rec_nm(const uint8_t *u, const uint64_t *rec_off, const uint32_t *nm_pos, RawOut r, uint64_t n_out, const uint64_t *roff, uint8_t *body) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_out || r.nm_new[k] < 0) return;
    const uint32_t src = r.src[k], np = nm_pos[src];
    if (!np) return;
    const uint64_t so = rec_off[src], qo = rec_off[r.qname_src[k]];
    const uint32_t lq_old = u[so + 12], lq_new = u[qo + 12];
    body[roff[k] + np - lq_old + lq_new] = (uint8_t)r.nm_new[k];                   // dataNM[1] = newValNM (group.cpp:570)
}

// ---- the sharded file runner: one engine per shard, every engine holds the whole stream and takes its share of it
struct ShardSrc { const gce_core *core; const uint64_t *qoff, *coff, *soff, *loff, *mioff, *tick, *roff; const int32_t *nm; const uint8_t *nmt; const uint32_t *nmpos; };
struct ShardDst { gce_core *core; uint64_t *qoff, *coff, *soff, *loff, *mioff, *tick, *roff; int32_t *nm; uint8_t *nmt; uint32_t *nmpos; };
__global__ __launch_bounds__(256) void k_shard_flag(const int32_t *shard, int64_t n, int32_t rank, uint8_t *flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = shard[i] == rank;
}
// thread per selected read: its key record and the per-read words of the batch, gathered (names, CIGAR words, bases and qualities stay where they are)
__global__ __launch_bounds__(256) void k_shard_gather(const uint32_t *sel, int64_t m, ShardSrc a, ShardDst d) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t k = sel[j];
    const uint4 *src = reinterpret_cast<const uint4 *>(a.core + k); uint4 *dst = reinterpret_cast<uint4 *>(d.core + j);
    dst[0] = src[0]; dst[1] = src[1];
    d.qoff[j] = a.qoff[k]; d.coff[j] = a.coff[k]; d.soff[j] = a.soff[k]; d.loff[j] = a.loff[k]; d.nm[j] = a.nm[k]; d.nmt[j] = a.nmt[k];
    if (a.mioff) d.mioff[j] = a.mioff[k];
    d.tick[j] = a.tick[k]; d.roff[j] = a.roff[k]; d.nmpos[j] = a.nmpos[k];
}
// what the merge of the shards' output tables compares: bamComp's fields (gencore.h:19-47) + the read's place in the WHOLE stream (quirk Q3) + the record's bytes
struct __attribute__((aligned(16))) MergeKey { int32_t tid, pos, mtid, mpos, isize; uint32_t gidx; uint32_t size, pad; };
__global__ __launch_bounds__(256) void k_merge_keys(const gce_core *core, const uint32_t *src, const uint32_t *sel, const uint64_t *rsize, uint64_t n_out, MergeKey *out) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    const uint32_t r = src[k]; const gce_core c = core[r];
    MergeKey m; m.tid = c.tid; m.pos = c.pos; m.mtid = c.mtid; m.mpos = c.mpos; m.isize = c.isize; m.gidx = sel ? sel[r] : r; m.size = (uint32_t)rsize[k]; m.pad = 0;
    out[k] = m;
}
// 16 lanes per record: record k of the merged stream = bytes [from[k], from[k] + size[k]) of the staged shard streams
__global__ __launch_bounds__(256) void k_merge_copy(const uint8_t *stage, const uint64_t *from, const uint64_t *to, const uint32_t *size, uint64_t n, uint8_t *body) {
    const int sub = threadIdx.x & 15;
    for (uint64_t k = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; k < n; k += ((uint64_t)gridDim.x * blockDim.x) >> 4) {
        const uint8_t *s = stage + from[k]; uint8_t *d = body + to[k]; const uint32_t sz = size[k];
        for (uint32_t j = 4 * sub; j < sz; j += 64) {
            if (j + 4 <= sz) *(rb_u32u *)(d + j) = rb32(s + j);
            else for (uint32_t q = j; q < sz; q++) d[q] = s[q];
        }
    }
}
__global__ void k_add_i64(long long *acc, const long long *x, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) acc[i] += x[i]; }

}  // namespace


// ---- prefix sums and compaction of the file layer on the engine's own scan kernels (gce_cluster.hpp: tiles of 2048, one block over the tile
//      totals): exclusive sums out[0 .. n] (out[n] = the total) of n 32- or 64-bit values, three launches; flagged indices, three launches
namespace {
template <class T> __global__ __launch_bounds__(256) void k_xs_reduce(const T *in, uint64_t n, uint64_t *part) {
    __shared__ uint64_t s4[4];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE; uint64_t v = 0;
    for (int k = 0; k < SCAN_TILE / 256; k++) { const uint64_t i = base + k * 256 + threadIdx.x; v += i < n ? (uint64_t)in[i] : 0ull; }
    v = (uint64_t)wave_sum64((long long)v);
    if (lane_id() == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = s4[0] + s4[1] + s4[2] + s4[3];
}
template <class T> __global__ __launch_bounds__(256) void k_xs_apply(const T *in, uint64_t n, const uint64_t *part, uint64_t *out) {
    __shared__ uint64_t s_w[4]; __shared__ uint64_t s_carry;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = part[blockIdx.x];
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    for (int k = 0; k < SCAN_TILE / 256; k++) {
        const uint64_t i = base + k * 256 + threadIdx.x;
        const uint64_t v = i < n ? (uint64_t)in[i] : 0ull; uint64_t x = v;
        for (int q = 1; q < 64; q <<= 1) { const uint64_t t = (uint64_t)__shfl_up((long long)x, q); if (lane >= q) x += t; }
        if (lane == 63) s_w[wv] = x;
        __syncthreads();
        uint64_t woff = 0;
        for (int q = 0; q < wv; q++) woff += s_w[q];
        const uint64_t carry = s_carry, ex = carry + woff + x - v;
        if (i < n) out[i] = ex;
        if (i + 1 == n) out[n] = ex + v;                                              // the total behind the last element
        __syncthreads();
        if (threadIdx.x == 255) s_carry = carry + woff + x;
        __syncthreads();
    }
}
}  // namespace
template <class T> static hipError_t dev_exclusive_sum(const T *in, uint64_t n, uint64_t *out, DevBuf &tmp, hipStream_t s) {
    if (n == 0) return hipMemsetAsync(out, 0, 8, s);
    const unsigned nb = (unsigned)((n + SCAN_TILE - 1) / SCAN_TILE);
    hipError_t e = tmp.ensure((size_t)nb * 8 + 64);
    if (e != hipSuccess) return e;
    uint64_t *part = tmp.as<uint64_t>();
    hipLaunchKernelGGL(k_xs_reduce<T>, dim3(nb), dim3(256), 0, s, in, n, part);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, part, (uint64_t)nb, (unsigned long long *)(part + nb), (unsigned long long *)(part + nb + 1));
    hipLaunchKernelGGL(k_xs_apply<T>, dim3(nb), dim3(256), 0, s, in, n, (const uint64_t *)part, out);
    return hipGetLastError();
}
// indices (ascending) of the set flags -> out, their number -> *count (device memory)
static hipError_t dev_select_flagged(const uint8_t *flag, uint64_t n, uint32_t *out, unsigned long long *count, DevBuf &tmp, hipStream_t s) {
    if (n == 0) return hipMemsetAsync(count, 0, 8, s);
    const unsigned nb = (unsigned)((n + SCAN_TILE - 1) / SCAN_TILE);
    hipError_t e = tmp.ensure((size_t)nb * 8 + 64);
    if (e != hipSuccess) return e;
    uint64_t *part = tmp.as<uint64_t>();
    hipLaunchKernelGGL(k_flag_reduce, dim3(nb), dim3(256), 0, s, flag, n, part);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, part, (uint64_t)nb, count, (unsigned long long *)nullptr);
    hipLaunchKernelGGL(k_flag_apply, dim3(nb), dim3(256), 0, s, flag, n, (const uint64_t *)part, out);
    return hipGetLastError();
}

extern "C" {

int gce_raw_begin(gce_engine *e, size_t capacity_hint) {
    if (!e) return GCE_ERR_INVALID;
    (void)hipSetDevice(e->prm.device);
    if (e->processed || e->host_mode || e->device_mode) gce_reset(e);
    if (!e->up_stream) HIPCHK(hipStreamCreate(&e->up_stream));
    HIPCHK(e->raw.ensure(capacity_hint + 256));
    {   // the per-record arrays, sized for the most records the stream can hold (a record with bases is > 96 bytes): allocated NOW, beside
        // the host's first inflate, instead of in gce_raw_finish on the critical path (hipMalloc of gigabytes takes tens of milliseconds)
        const size_t n1 = capacity_hint / 96 + 1024, nseg = capacity_hint / RAW_SEG + 16;
        HIPCHK(e->rw_guess.ensure(nseg * 8)); HIPCHK(e->rw_leave.ensure(nseg * 8)); HIPCHK(e->rw_cnt.ensure(nseg * 4 + 8)); HIPCHK(e->rw_base.ensure(nseg * 8 + 8)); HIPCHK(e->rw_misc.ensure(64));
        HIPCHK(e->rw_off.ensure((n1 + 1) * 8));
        HIPCHK(e->b_core.ensure(n1 * sizeof(gce_core) + 64)); HIPCHK(e->b_qoff.ensure(n1 * 8 + 64)); HIPCHK(e->b_coff.ensure((n1 + 1) * 8 + 64)); HIPCHK(e->b_soff.ensure(n1 * 8 + 64)); HIPCHK(e->b_loff.ensure(n1 * 8 + 64));
        HIPCHK(e->b_nm.ensure(n1 * 4 + 64)); HIPCHK(e->b_nmt.ensure(n1 + 64)); HIPCHK(e->b_mioff.ensure(n1 * 8 + 64)); HIPCHK(e->rw_ncig.ensure(n1 * 4 + 64)); HIPCHK(e->rw_nmpos.ensure(n1 * 4 + 64));
        HIPCHK(e->b_cigar.ensure(n1 * 8 + 64));
    }
    e->raw_n = 0; e->raw_mode = true; e->raw_records = 0; e->z_n = 0; e->z_members.clear();
    return GCE_OK;
}

// bytes of the inflated stream, in order.  Asynchronous: `host` (pinned memory makes it a real DMA) must stay untouched until gce_raw_wait(ticket).
int gce_raw_push(gce_engine *e, const void *host, size_t bytes, int32_t *ticket) {
    if (!e || !e->raw_mode || (!host && bytes)) return GCE_ERR_INVALID;
    (void)hipSetDevice(e->prm.device);
    if (e->raw_n + bytes + 256 > e->raw.cap) {                                     // grow: the copies so far are in flight on the same stream, the move queues behind them
        DevBuf nb;
        HIPCHK(nb.ensure((e->raw_n + bytes) * 2 + 256));
        const size_t keep = std::min(e->raw_n, e->raw.cap);                        // (members waiting for the GPU inflate have places, not bytes yet: they may lie past the old buffer)
        if (keep) HIPCHK(hipMemcpyAsync(nb.p, e->raw.p, keep, hipMemcpyDeviceToDevice, e->up_stream));
        HIPCHK(hipStreamSynchronize(e->up_stream));
        e->raw.release(); e->raw = nb; nb.p = nullptr; nb.cap = 0;
    }
    if (bytes) HIPCHK(hipMemcpyAsync((char *)e->raw.p + e->raw_n, host, bytes, hipMemcpyHostToDevice, e->up_stream));
    e->raw_n += bytes;
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev, e->up_stream));
    e->up_events.push_back(ev);
    if (ticket) *ticket = (int32_t)e->up_events.size() - 1;
    for (gce_engine *m : e->mirrors) { const int rm = gce_raw_push(m, host, bytes, nullptr); if (rm != GCE_OK) return fail(e, rm, gce_last_error(m)); }
    return GCE_OK;
}

// BGZF members as they lie in the file (`comp`: host memory, comp_bytes; member k at coff[k], csize[k] bytes, ISIZE usize[k]) -> they are
// copied to HBM compressed and inflated BY THE GPU (gce_inflate.hpp), in one launch, when gce_raw_finish is called; their bytes follow what
// has been pushed so far.  Members with usize 0 (the EOF marker) may be passed or left out.  Asynchronous like gce_raw_push.
int gce_raw_push_bgzf(gce_engine *e, const void *comp, size_t comp_bytes, int32_t n_members, const uint64_t *coff, const uint32_t *csize, const uint32_t *usize, int32_t *ticket) {
    if (!e || !e->raw_mode || n_members < 0 || (n_members && (!comp || !coff || !csize || !usize))) return GCE_ERR_INVALID;
    for (int32_t k = 0; k < n_members; k++)                                             // every member is looked at before anything is queued or counted
        if (coff[k] > comp_bytes || csize[k] > comp_bytes - coff[k] || usize[k] > 0x10000u) return fail(e, GCE_ERR_INVALID, "BGZF member outside its buffer");
    (void)hipSetDevice(e->prm.device);
    if (e->z_n + comp_bytes + 64 > e->z_comp.cap) {
        DevBuf nb;
        HIPCHK(nb.ensure(std::max<size_t>((e->z_n + comp_bytes) * 2, e->raw.cap / 4) + 64));     // (gce_raw_begin sized the raw stream for ~5 x the file: a quarter of it holds the file, no second growth)
        if (e->z_n) HIPCHK(hipMemcpyAsync(nb.p, e->z_comp.p, e->z_n, hipMemcpyDeviceToDevice, e->up_stream));
        HIPCHK(hipStreamSynchronize(e->up_stream));
        e->z_comp.release(); e->z_comp = nb; nb.p = nullptr; nb.cap = 0;
    }
    if (comp_bytes) HIPCHK(hipMemcpyAsync((char *)e->z_comp.p + e->z_n, comp, comp_bytes, hipMemcpyHostToDevice, e->up_stream));
    for (int32_t k = 0; k < n_members; k++) {
        if (usize[k] == 0) continue;
        InfDir d; d.coff = e->z_n + coff[k]; d.uoff = e->raw_n; d.csize = csize[k]; d.usize = usize[k];
        e->z_members.push_back(d); e->raw_n += usize[k];
    }
    e->z_n += comp_bytes;
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev, e->up_stream));
    e->up_events.push_back(ev);
    if (ticket) *ticket = (int32_t)e->up_events.size() - 1;
    for (gce_engine *m : e->mirrors) { const int rm = gce_raw_push_bgzf(m, comp, comp_bytes, n_members, coff, csize, usize, nullptr); if (rm != GCE_OK) return fail(e, rm, gce_last_error(m)); }
    return GCE_OK;
}

// the launch behind gce_raw_push_bgzf: every member waiting, one lane each
static int raw_inflate_pending(gce_engine *e) {
    if (e->z_members.empty()) return GCE_OK;
    hipStream_t s = e->stream;
    if (e->raw_n + 256 > e->raw.cap) {
        DevBuf nb;
        HIPCHK(nb.ensure(e->raw_n + 256));
        const size_t keep = std::min(e->raw_n, e->raw.cap);                        // (everything so far: host windows may lie between the members' places)
        if (keep) HIPCHK(hipMemcpyAsync(nb.p, e->raw.p, keep, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        e->raw.release(); e->raw = nb; nb.p = nullptr; nb.cap = 0;
    }
    const size_t n = e->z_members.size(), LAUNCH = (size_t)1 << 18;                // members per launch: bounds the code-length scratch (320 bytes per member of a launch)
    HIPCHK(e->z_dir.ensure(n * sizeof(InfDir) + std::min(n, LAUNCH) * INF_NSYM)); HIPCHK(e->z_err.ensure(16));
    HIPCHK(hipMemcpyAsync(e->z_dir.p, e->z_members.data(), n * sizeof(InfDir), hipMemcpyHostToDevice, s));
    const unsigned int init[2] = {0u, 0xFFFFFFFFu};
    HIPCHK(hipMemcpyAsync(e->z_err.p, init, 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync((char *)e->z_comp.p + e->z_n, 0, 64, s));                 // (the bit reader looks up to 32 bytes ahead)
    unsigned int got[2] = {0, 0}; uint32_t bad_member = 0xFFFFFFFFu;
    for (size_t base = 0; base < n; base += LAUNCH) {
        const size_t m = std::min(LAUNCH, n - base);
        hipLaunchKernelGGL(k_bgzf_inflate, dim3((unsigned)((m + INF_T - 1) / INF_T)), dim3(INF_T), 0, s, e->z_comp.as<uint8_t>(), (const InfDir *)e->z_dir.p + base, (uint32_t)m, e->raw.as<uint8_t>(), e->z_err.as<unsigned int>(),
                           e->z_dir.as<uint8_t>() + n * sizeof(InfDir));
        if (base + LAUNCH < n) {                                                   // (the member number of a failure is relative to its launch: fetch it per launch when there are several)
            HIPCHK(hipMemcpyAsync(got, e->z_err.p, 8, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
            if (got[0]) { bad_member = (uint32_t)base + got[1]; break; }
        }
    }
    if (bad_member == 0xFFFFFFFFu) {
        HIPCHK(hipMemcpyAsync(got, e->z_err.p, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (got[0]) bad_member = (uint32_t)((n - 1) / LAUNCH * LAUNCH) + got[1];
    }
    HIPCHK(hipGetLastError());
    e->z_comp.release();                                                           // (the compressed copy of the file: not needed while the stream is processed)
    e->z_members.clear(); e->z_n = 0;
    if (bad_member != 0xFFFFFFFFu) { char m[96]; snprintf(m, sizeof m, "inflate / CRC failure in BGZF member %u of the GPU batch", bad_member); return fail(e, GCE_ERR_INVALID, m); }
    return GCE_OK;
}

// BGZF members -> bytes, nothing else (tests, tools): the decoder of gce_raw_push_bgzf on buffers of the caller.  first_bad: -1 or the first
// member that failed a check (its bytes and those of later failures are undefined).
int gce_bgzf_inflate(int32_t device, const void *comp, size_t comp_bytes, int32_t n_members, const uint64_t *coff, const uint32_t *csize, const uint32_t *usize, void *out, int32_t *first_bad) {
    if (n_members < 0 || (n_members && (!comp || !coff || !csize || !usize || !out))) return GCE_ERR_INVALID;
    if (first_bad) *first_bad = -1;
    if (hipSetDevice(device) != hipSuccess) return GCE_ERR_NO_DEVICE;
    std::vector<InfDir> dir; uint64_t total = 0;
    for (int32_t k = 0; k < n_members; k++) {
        if (coff[k] > comp_bytes || csize[k] > comp_bytes - coff[k] || usize[k] > 0x10000u) return GCE_ERR_INVALID;
        InfDir d; d.coff = coff[k]; d.uoff = total; d.csize = csize[k]; d.usize = usize[k]; dir.push_back(d); total += usize[k];
    }
    if (dir.empty()) return GCE_OK;
    DevBuf zc, zd, ze, zo;
    int rc = GCE_OK;
    auto chk = [&](hipError_t x) { if (x != hipSuccess && rc == GCE_OK) rc = GCE_ERR_HIP; };
    if (zc.ensure(comp_bytes + 64) != hipSuccess || zd.ensure(dir.size() * (sizeof(InfDir) + INF_NSYM)) != hipSuccess || ze.ensure(16) != hipSuccess || zo.ensure(total + 64) != hipSuccess) rc = GCE_ERR_OOM;
    if (rc == GCE_OK) {
        chk(hipMemcpy(zc.p, comp, comp_bytes, hipMemcpyHostToDevice)); chk(hipMemset((char *)zc.p + comp_bytes, 0, 64));
        chk(hipMemcpy(zd.p, dir.data(), dir.size() * sizeof(InfDir), hipMemcpyHostToDevice));
        const unsigned int init[2] = {0u, 0xFFFFFFFFu};
        chk(hipMemcpy(ze.p, init, 8, hipMemcpyHostToDevice));
        if (rc == GCE_OK) {
            hipLaunchKernelGGL(k_bgzf_inflate, dim3((unsigned)((dir.size() + INF_T - 1) / INF_T)), dim3(INF_T), 0, 0, zc.as<uint8_t>(), (const InfDir *)zd.p, (uint32_t)dir.size(), zo.as<uint8_t>(), ze.as<unsigned int>(), zd.as<uint8_t>() + dir.size() * sizeof(InfDir));
            chk(hipDeviceSynchronize()); chk(hipGetLastError());
            unsigned int got[2] = {0, 0};
            chk(hipMemcpy(got, ze.p, 8, hipMemcpyDeviceToHost));
            if (total) chk(hipMemcpy(out, zo.p, total, hipMemcpyDeviceToHost));
            if (rc == GCE_OK && got[0]) { if (first_bad) *first_bad = (int32_t)got[1]; rc = GCE_ERR_INVALID; }
        }
    }
    zc.release(); zd.release(); ze.release(); zo.release();
    return rc;
}

// The encoder of gce_raw_deflate_output on the caller's buffer (tests, tools): `n` bytes -> BGZF blocks of `block_bytes` input bytes each (<= 65 280),
// back to back in out[0 .. *out_bytes); out_cap >= n + n / 8 + 64 x blocks is always enough.  Mirror of gce_bgzf_inflate.
int gce_bgzf_deflate(int32_t device, const void *in, size_t n, uint32_t block_bytes, void *out, size_t out_cap, size_t *out_bytes) {
    if ((n && !in) || !out_bytes || block_bytes < 1 || block_bytes > 0xff00u || (n && !out)) return GCE_ERR_INVALID;
    *out_bytes = 0;
    if (hipSetDevice(device) != hipSuccess) return GCE_ERR_NO_DEVICE;
    if (!n) return GCE_OK;
    const uint64_t nb64 = (n + block_bytes - 1) / block_bytes;
    if (nb64 >= 0x7FFFFFF0ull) return GCE_ERR_INVALID;
    const uint32_t nb = (uint32_t)nb64, slot = block_bytes + block_bytes / 8 + 64;
    DevBuf zi, zs, zz, zo, zf, zt;
    int rc = GCE_OK;
    auto chk = [&](hipError_t x) { if (x != hipSuccess && rc == GCE_OK) rc = GCE_ERR_HIP; };
    if (zi.ensure(n + 64) != hipSuccess || zs.ensure((size_t)nb * slot + 64) != hipSuccess || zz.ensure(((size_t)nb + 1) * 4) != hipSuccess || zf.ensure(((size_t)nb + 1) * 8) != hipSuccess) rc = GCE_ERR_OOM;
    if (rc == GCE_OK) {
        chk(hipMemcpy(zi.p, in, n, hipMemcpyHostToDevice)); chk(hipMemset((char *)zi.p + n, 0, 64));
        hipLaunchKernelGGL(k_bgzf_deflate, dim3((nb + DEF_T - 1) / DEF_T), dim3(DEF_T), 0, 0, (const uint8_t *)zi.p, (uint64_t)n, block_bytes, nb, zs.as<uint8_t>(), slot, zz.as<uint32_t>());
        chk(hipMemset((char *)zz.p + (size_t)nb * 4, 0, 4));
        chk(dev_exclusive_sum(zz.as<uint32_t>(), (uint64_t)nb, zf.as<uint64_t>(), zt, 0));
        uint64_t csz = 0;
        if (rc == GCE_OK) chk(hipMemcpy(&csz, zf.as<uint64_t>() + nb, 8, hipMemcpyDeviceToHost));
        if (rc == GCE_OK && csz > out_cap) rc = GCE_ERR_INVALID;
        if (rc == GCE_OK && zo.ensure(csz + 64) != hipSuccess) rc = GCE_ERR_OOM;
        if (rc == GCE_OK) {
            hipLaunchKernelGGL(k_deflate_pack, dim3(std::min<uint32_t>((nb + 3) / 4, 16384u)), dim3(256), 0, 0, (const uint8_t *)zs.p, slot, (const uint32_t *)zz.p, (const uint64_t *)zf.p, nb, zo.as<uint8_t>());
            chk(hipDeviceSynchronize()); chk(hipGetLastError());
            chk(hipMemcpy(out, zo.p, csz, hipMemcpyDeviceToHost));
            if (rc == GCE_OK) *out_bytes = (size_t)csz;
        }
    }
    zi.release(); zs.release(); zz.release(); zo.release(); zf.release(); zt.release();
    return rc;
}

// The whole stream is in HBM: index the records behind `records_begin` (the end of the BAM header), build the batch.  Afterwards the engine is
// in the state gce_submit_device leaves it in: gce_process, then gce_drain / gce_result_device or gce_raw_build_output.
int gce_raw_finish(gce_engine *e, uint64_t records_begin, int32_t n_ref, int64_t *n_records) {
    if (!e || !e->raw_mode || records_begin > e->raw_n) return GCE_ERR_INVALID;
    (void)hipSetDevice(e->prm.device);
    const bool tprint = getenv("GCE_RAW_TIMING") != nullptr; const auto tnow = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }; double tq = tnow();
    auto lap = [&](const char *what) { if (tprint) { (void)hipStreamSynchronize(e->stream); const double x = tnow(); fprintf(stderr, "gce_raw_finish %s %.4f s\n", what, x - tq); tq = x; } };
    HIPCHK(hipStreamSynchronize(e->up_stream));
    lap("wait for the last copies");
    for (auto ev : e->up_events) (void)hipEventDestroy(ev);
    e->up_events.clear();
    { const int zr = raw_inflate_pending(e); if (zr != GCE_OK) return zr; }
    lap("BGZF members inflated by the GPU");
    hipStream_t s = e->stream;
    const uint8_t *u = e->raw.as<uint8_t>();
    const uint64_t total = e->raw_n;
    HIPCHK(hipMemsetAsync((char *)e->raw.p + total, 0, 64, s));                    // the blobs are readable 16 bytes past their end
    uint64_t n_rec = 0;
    if (total > records_begin) {
        const uint64_t nseg = (total - records_begin + RAW_SEG - 1) / RAW_SEG;
        HIPCHK(e->rw_guess.ensure(nseg * 8)); HIPCHK(e->rw_leave.ensure(nseg * 8)); HIPCHK(e->rw_cnt.ensure(nseg * 4 + 8)); HIPCHK(e->rw_base.ensure(nseg * 8 + 8)); HIPCHK(e->rw_misc.ensure(64)); HIPCHK(e->rw_bad.ensure(nseg + 8));
        HIPCHK(hipMemsetAsync(e->rw_misc.p, 0, 64, s));
        const unsigned nbs = (unsigned)((nseg + 255) / 256);
        hipLaunchKernelGGL(k_raw_seg, dim3(nbs), dim3(256), 0, s, u, records_begin, total, n_ref, nseg, e->rw_guess.as<uint64_t>(), e->rw_leave.as<uint64_t>(), e->rw_cnt.as<uint32_t>());
        hipLaunchKernelGGL(k_raw_check, dim3(nbs), dim3(256), 0, s, (const uint64_t *)e->rw_guess.p, (const uint64_t *)e->rw_leave.p, nseg, total, e->rw_misc.as<unsigned int>(), e->rw_bad.as<uint8_t>());
        unsigned int flags[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(flags, e->rw_misc.p, 8, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
        lap("segment walks + check");
        if (tprint) fprintf(stderr, "gce_raw_finish: %u of %llu segments off the chain\n", flags[0], (unsigned long long)nseg);
        if (tprint && flags[0]) {                                                       // which ones, and why
            std::vector<uint64_t> g(nseg), l(nseg);
            (void)hipMemcpy(g.data(), e->rw_guess.p, nseg * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(l.data(), e->rw_leave.p, nseg * 8, hipMemcpyDeviceToHost);
            int shown = 0;
            for (uint64_t q = 1; q < nseg && shown < 6; q++) if (l[q] == ~0ull || g[q] != l[q - 1]) { fprintf(stderr, "  segment %llu [%llu, +16K): guess %lld, chain enters at %lld, leaves %lld\n", (unsigned long long)q, (unsigned long long)(records_begin + q * RAW_SEG), (long long)g[q], (long long)l[q - 1], (long long)l[q]); shown++; }
        }
        for (int round = 0; flags[0] && round < 64; round++) {                           // parallel repair rounds
            HIPCHK(hipMemsetAsync(e->rw_misc.p, 0, 16, s));
            hipLaunchKernelGGL(k_raw_fix, dim3(nbs), dim3(256), 0, s, u, records_begin, total, nseg, e->rw_guess.as<uint64_t>(), e->rw_leave.as<uint64_t>(), e->rw_cnt.as<uint32_t>(), (const uint8_t *)e->rw_bad.p, e->rw_misc.as<unsigned int>() + 3, e->rw_misc.as<unsigned int>() + 1);
            hipLaunchKernelGGL(k_raw_check, dim3(nbs), dim3(256), 0, s, (const uint64_t *)e->rw_guess.p, (const uint64_t *)e->rw_leave.p, nseg, total, e->rw_misc.as<unsigned int>(), e->rw_bad.as<uint8_t>());
            HIPCHK(hipMemcpyAsync(flags, e->rw_misc.p, 8, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
            if (flags[1]) return fail(e, GCE_ERR_INVALID, "truncated or damaged BAM record stream");
        }
        lap("parallel repair");
        if (flags[0]) {
            HIPCHK(hipMemsetAsync(e->rw_misc.p, 0, 16, s));
            hipLaunchKernelGGL(k_raw_repair, dim3(1), dim3(64), 0, s, u, records_begin, total, nseg, e->rw_guess.as<uint64_t>(), e->rw_leave.as<uint64_t>(), e->rw_cnt.as<uint32_t>(), e->rw_misc.as<unsigned int>() + 1);
            HIPCHK(hipMemcpyAsync(flags, e->rw_misc.p, 8, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
            if (flags[1]) return fail(e, GCE_ERR_INVALID, "truncated or damaged BAM record stream");
            lap("repair");
        }
        HIPCHK(dev_exclusive_sum(e->rw_cnt.as<uint32_t>(), nseg, e->rw_base.as<uint64_t>(), e->rw_tmp, s));      // exclusive scan of the segments' record counts: the total comes out as base[nseg]
        HIPCHK(hipMemcpyAsync(&n_rec, e->rw_base.as<uint64_t>() + nseg, 8, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
        if (n_rec >= 0x7FFFFFF0ull) return fail(e, GCE_ERR_INVALID, "more than 2^31 records in one stream");
        lap("segments + scan");
        HIPCHK(e->rw_off.ensure((size_t)(n_rec + 1) * 8));
        hipLaunchKernelGGL(k_raw_offsets, dim3(nbs), dim3(256), 0, s, u, records_begin, total, nseg, (const uint64_t *)e->rw_guess.p, (const uint64_t *)e->rw_base.p, e->rw_off.as<uint64_t>());
    }
    const size_t n1 = (size_t)(n_rec ? n_rec : 1);
    HIPCHK(e->b_core.ensure(n1 * sizeof(gce_core) + 64)); HIPCHK(e->b_qoff.ensure(n1 * 8 + 64)); HIPCHK(e->b_coff.ensure(n1 * 8 + 64)); HIPCHK(e->b_soff.ensure(n1 * 8 + 64)); HIPCHK(e->b_loff.ensure(n1 * 8 + 64));
    HIPCHK(e->b_nm.ensure(n1 * 4 + 64)); HIPCHK(e->b_nmt.ensure(n1 + 64)); HIPCHK(e->b_mioff.ensure(n1 * 8 + 64)); HIPCHK(e->rw_ncig.ensure(n1 * 4 + 64)); HIPCHK(e->rw_nmpos.ensure(n1 * 4 + 64));
    lap("offsets + allocations");
    uint64_t cig_words = 0;
    unsigned int have_mi = 0;
    if (n_rec) {
        RawSoA o{e->b_core.as<gce_core>(), e->b_qoff.as<uint64_t>(), e->b_soff.as<uint64_t>(), e->b_loff.as<uint64_t>(), e->b_mioff.as<uint64_t>(), e->rw_ncig.as<uint32_t>(), e->rw_nmpos.as<uint32_t>(),
                 e->b_nm.as<int32_t>(), e->b_nmt.as<uint8_t>(), e->rw_misc.as<unsigned int>() + 2, e->rw_misc.as<unsigned int>() + 4, n_ref};
        HIPCHK(hipMemsetAsync(e->rw_misc.as<unsigned int>() + 4, 0xFF, 4, s));
        const unsigned nbr = (unsigned)((n_rec + 255) / 256);
        hipLaunchKernelGGL(k_raw_fill, dim3(nbr), dim3(256), 0, s, u, (const uint64_t *)e->rw_off.p, n_rec, o);
        HIPCHK(e->b_coff.ensure((n1 + 1) * 8 + 64));
        HIPCHK(dev_exclusive_sum(e->rw_ncig.as<uint32_t>(), n_rec, e->b_coff.as<uint64_t>(), e->rw_tmp, s));
        HIPCHK(hipMemcpyAsync(&cig_words, e->b_coff.as<uint64_t>() + n_rec, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(&have_mi, e->rw_misc.as<unsigned int>() + 2, 4, hipMemcpyDeviceToHost, s));
        unsigned int bad_rec = NONE32;
        HIPCHK(hipMemcpyAsync(&bad_rec, e->rw_misc.as<unsigned int>() + 4, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (bad_rec != NONE32) { char m[96]; snprintf(m, sizeof m, "inconsistent record lengths (record %u)", bad_rec); return fail(e, GCE_ERR_INVALID, m); }
        HIPCHK(e->b_cigar.ensure((size_t)cig_words * 4 + 64));
        hipLaunchKernelGGL(k_raw_cigar, dim3(nbr), dim3(256), 0, s, u, (const uint64_t *)e->rw_off.p, n_rec, (const uint64_t *)e->b_coff.p, e->b_cigar.as<uint32_t>());
    } else HIPCHK(e->b_cigar.ensure(64));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    lap("fill + cigar");
    gce_batch &d = e->dev_batch; memset(&d, 0, sizeof d);
    d.n_reads = (int64_t)n_rec; d.core = e->b_core.as<gce_core>();
    d.qname_off = e->b_qoff.as<uint64_t>(); d.qname = e->raw.as<char>(); d.cigar_off = e->b_coff.as<uint64_t>(); d.cigar = e->b_cigar.as<uint32_t>();
    d.seq_off = e->b_soff.as<uint64_t>(); d.seq = e->raw.as<uint8_t>(); d.qual_off = e->b_loff.as<uint64_t>(); d.qual = e->raw.as<uint8_t>();
    d.nm = e->b_nm.as<int32_t>(); d.nm_type = e->b_nmt.as<uint8_t>();
    if (have_mi) { d.mi_off = e->b_mioff.as<uint64_t>(); d.mi = e->raw.as<char>(); d.mi_bytes = total; }
    d.qname_bytes = total; d.cigar_words = cig_words; d.seq_bytes = total; d.qual_bytes = total;
    e->device_mode = true; e->have_tick = false; e->raw_records = (int64_t)n_rec;
    if (n_records) *n_records = (int64_t)n_rec;
    return GCE_OK;
}

// after gce_process: the emitted records as a stream of BAM records in HBM, in the order of the output table
int gce_raw_build_output(gce_engine *e, uint64_t *body_bytes, int64_t *n_out) {
    if (!e || !e->raw_mode || !e->processed || e->dev_error || !body_bytes) return GCE_ERR_INVALID;
    (void)hipSetDevice(e->prm.device);
    hipStream_t s = e->stream;
    const uint64_t no = (uint64_t)e->n_out;
    *body_bytes = 0; if (n_out) *n_out = e->n_out;
    if (!no) return GCE_OK;
    HIPCHK(e->rw_rsize.ensure((no + 1) * 8)); HIPCHK(e->rw_roff.ensure((no + 1) * 8));
    const uint8_t *u = e->raw.as<uint8_t>();
    RawOut r{e->o_src.as<uint32_t>(), e->o_qsrc.as<uint32_t>(), e->o_nm.as<int32_t>(), e->o_fr.as<int16_t>(), e->o_rr.as<int16_t>()};
    const unsigned nb = (unsigned)((no + 255) / 256);
    hipLaunchKernelGGL(k_rec_size, dim3(nb), dim3(256), 0, s, u, (const uint64_t *)e->rw_off.p, r, no, e->rw_rsize.as<uint64_t>());
    HIPCHK(dev_exclusive_sum(e->rw_rsize.as<uint64_t>(), no, e->rw_roff.as<uint64_t>(), e->rw_tmp, s));
    uint64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, e->rw_roff.as<uint64_t>() + no, 8, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
    HIPCHK(e->rw_body.ensure(total + 64));
    hipLaunchKernelGGL(k_rec_build, dim3((unsigned)std::min<uint64_t>((no + 15) / 16, 65535u)), dim3(256), 0, s, u, (const uint64_t *)e->rw_off.p, (const uint32_t *)e->rw_nmpos.p, r, no, (const uint64_t *)e->rw_roff.p, e->rw_body.as<uint8_t>());
    hipLaunchKernelGGL(k_rec_nm, dim3(nb), dim3(256), 0, s, u, (const uint64_t *)e->rw_off.p, (const uint32_t *)e->rw_nmpos.p, r, no, (const uint64_t *)e->rw_roff.p, e->rw_body.as<uint8_t>());
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    e->raw_body_bytes = total;
    *body_bytes = total;
    return GCE_OK;
}

int gce_raw_read_output_async(gce_engine *e, uint64_t offset, void *host, size_t bytes, int32_t *ticket) {
    if (!e || !e->raw_mode || offset + bytes > e->raw_body_bytes || (!host && bytes)) return GCE_ERR_INVALID;
    (void)hipSetDevice(e->prm.device);
    if (bytes) HIPCHK(hipMemcpyAsync(host, (const char *)e->rw_body.p + offset, bytes, hipMemcpyDeviceToHost, e->up_stream));
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev, e->up_stream));
    e->up_events.push_back(ev);
    if (ticket) *ticket = (int32_t)e->up_events.size() - 1;
    return GCE_OK;
}


// ---- several engines over ONE file (gce_run_bam_sharded): the mirrors of an engine receive every gce_raw_push / gce_raw_push_bgzf made to it
int gce_raw_attach_mirror(gce_engine *e, gce_engine *mirror) {
    if (!e || !mirror || e == mirror || !e->raw_mode || !mirror->raw_mode || e->raw_n != mirror->raw_n) return GCE_ERR_INVALID;
    // gce_submit_wait waits on the SAME ticket number of every mirror: the two engines must have handed out the same tickets so far, and neither
    // may hold BGZF members the other never saw (an attach behind an earlier push made the wait land on the wrong event)
    if (e->up_events.size() != mirror->up_events.size() || !e->z_members.empty() || !mirror->z_members.empty()) return fail(e, GCE_ERR_INVALID, "gce_raw_attach_mirror: attach before the first push to either engine");
    e->mirrors.push_back(mirror);
    return GCE_OK;
}
// After gce_raw_finish: this engine keeps shard `rank` of `world` of the stream it holds -- the planner runs on its own device over the key
// records in HBM (gce_stream_context: every read's global tick, the flush events of the whole stream; gce_plan_shards: key ranges or clusters
// dealt by weight), the reads of the shard are gathered (key records and per-read words; names, CIGARs, bases and qualities stay in the raw
// stream), the engine gets their ticks and the events.  Every engine of a sharded run computes the same plan from the same stream: nothing is
// exchanged.  Replaces the host-side cut of round 2's runner (gencore.cpp:164-205 over several GPUs).
int gce_raw_select_shard(gce_engine *e, int32_t world, int32_t rank, int32_t plan_mode) {
    if (!e || !e->raw_mode || !e->device_mode || e->processed || world < 1 || rank < 0 || rank >= world) return GCE_ERR_INVALID;
    (void)hipSetDevice(e->prm.device);
    int64_t n = e->raw_records;
    hipStream_t s = e->stream;
    e->shard_n = 0;
    if (n == 0) { e->have_tick = false; return GCE_OK; }
    const int64_t n_all = n;
    // --quit_after_contig (gencore.cpp:243-246) ends the loop ONCE, on the first read of the whole stream with tid >= maxContig: that read is
    // counted by the pre-Stats and nothing behind it exists.  The cut is made here, on the whole stream, in front of the plan: shard 0 receives
    // the cut read behind its own reads and counts it (its gce_process finds the cut there), the other engines do not look for a cut at all --
    // every shard cutting its own part counted one extra read per shard that held a read of the later contigs.
    int64_t cut = -1;
    e->shard_cut_done = false;
    if (e->prm.max_contig > 0) {
        unsigned int first = NONE32;
        HIPCHK(hipMemcpyAsync(e->rw_misc.p, &first, 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_first_contig_ge, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const gce_core *)e->b_core.as<gce_core>(), n, e->prm.max_contig, (unsigned int *)e->rw_misc.p);
        HIPCHK(hipMemcpyAsync(&first, e->rw_misc.p, 4, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
        if (first != NONE32) { cut = (int64_t)first; n = cut; }
        e->shard_cut_done = rank != 0 || cut < 0;
    }
    HIPCHK(e->sh_tickall.ensure((size_t)n_all * 8)); HIPCHK(e->sh_shard.ensure((size_t)n_all * 4)); HIPCHK(e->sh_flag.ensure((size_t)n_all + 64)); HIPCHK(e->sh_sel.ensure((size_t)n_all * 4 + 64));
    int rc;
    if (n > 0) {
        int32_t n_ev = 0, *ev_tid = nullptr, *ev_pos = nullptr;
        const int period = e->prm.flush_period > 0 ? e->prm.flush_period : 10000;
        rc = gce_stream_context(e->prm.device, e->b_core.as<gce_core>(), n, period, e->sh_tickall.as<uint64_t>(), &n_ev, &ev_tid, &ev_pos);
        if (rc != GCE_OK) return fail(e, rc, rc == GCE_ERR_INVALID ? "not shardable by cluster key: a mapped read follows the first unmapped read" : gce_status_message(rc));
        rc = gce_set_flush_events(e, n_ev, ev_tid, ev_pos);
        gce_free(ev_tid); gce_free(ev_pos);
        if (rc != GCE_OK) return rc;
        if ((rc = gce_plan_shards(e->prm.device, e->b_core.as<gce_core>(), n, world, plan_mode, e->sh_shard.as<int32_t>())) != GCE_OK) return fail(e, rc, gce_status_message(rc));
        const unsigned nb = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_shard_flag, dim3(nb), dim3(256), 0, s, (const int32_t *)e->sh_shard.p, n, rank, e->sh_flag.as<uint8_t>());
    } else if ((rc = gce_set_flush_events(e, 0, nullptr, nullptr)) != GCE_OK) return rc;
    if (cut >= 0) {                                                                  // the cut read: shard 0's last read (its tick is never looked at: gce_process drops it)
        HIPCHK(hipMemsetAsync(e->sh_flag.as<uint8_t>() + cut, rank == 0 ? 1 : 0, 1, s));
        HIPCHK(hipMemsetAsync(e->sh_tickall.as<uint64_t>() + cut, 0, 8, s));
    }
    HIPCHK(dev_select_flagged(e->sh_flag.as<uint8_t>(), (uint64_t)(n + (cut >= 0 ? 1 : 0)), e->sh_sel.as<uint32_t>(), (unsigned long long *)e->rw_misc.p, e->rw_tmp, s));
    int64_t m = 0;
    HIPCHK(hipMemcpyAsync(&m, e->rw_misc.p, 8, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
    const size_t m1 = (size_t)(m > 0 ? m : 1);
    HIPCHK(e->sh_core.ensure(m1 * sizeof(gce_core) + 64)); HIPCHK(e->sh_qoff.ensure(m1 * 8 + 64)); HIPCHK(e->sh_coff.ensure(m1 * 8 + 64)); HIPCHK(e->sh_soff.ensure(m1 * 8 + 64)); HIPCHK(e->sh_loff.ensure(m1 * 8 + 64));
    HIPCHK(e->sh_nm.ensure(m1 * 4 + 64)); HIPCHK(e->sh_nmt.ensure(m1 + 64)); HIPCHK(e->sh_mioff.ensure(m1 * 8 + 64)); HIPCHK(e->sh_tick.ensure(m1 * 8 + 64)); HIPCHK(e->sh_roff.ensure(m1 * 8 + 64)); HIPCHK(e->sh_nmpos.ensure(m1 * 4 + 64));
    gce_batch &d = e->dev_batch;
    if (m > 0) {
        ShardSrc a{d.core, d.qname_off, d.cigar_off, d.seq_off, d.qual_off, d.mi_off, (const uint64_t *)e->sh_tickall.p, (const uint64_t *)e->rw_off.p, d.nm, d.nm_type, (const uint32_t *)e->rw_nmpos.p};
        ShardDst o{e->sh_core.as<gce_core>(), e->sh_qoff.as<uint64_t>(), e->sh_coff.as<uint64_t>(), e->sh_soff.as<uint64_t>(), e->sh_loff.as<uint64_t>(), e->sh_mioff.as<uint64_t>(), e->sh_tick.as<uint64_t>(),
                   e->sh_roff.as<uint64_t>(), e->sh_nm.as<int32_t>(), e->sh_nmt.as<uint8_t>(), e->sh_nmpos.as<uint32_t>()};
        hipLaunchKernelGGL(k_shard_gather, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (const uint32_t *)e->sh_sel.p, m, a, o);
    }
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    d.n_reads = m; d.core = e->sh_core.as<gce_core>(); d.qname_off = e->sh_qoff.as<uint64_t>(); d.cigar_off = e->sh_coff.as<uint64_t>(); d.seq_off = e->sh_soff.as<uint64_t>(); d.qual_off = e->sh_loff.as<uint64_t>();
    d.nm = e->sh_nm.as<int32_t>(); d.nm_type = e->sh_nmt.as<uint8_t>(); if (d.mi_off) d.mi_off = e->sh_mioff.as<uint64_t>();
    d.tick = e->sh_tick.as<uint64_t>();
    std::swap(e->rw_off, e->sh_roff); std::swap(e->rw_nmpos, e->sh_nmpos);          // gce_raw_build_output looks records up by the batch's (now: the shard's) read index
    e->have_tick = true; e->raw_records = m; e->shard_n = m;
    e->sh_tickall.release(); e->sh_shard.release(); e->sh_flag.release();
    return GCE_OK;
}
// The Stats payloads of several engines (gce_stats_payload_device, same step and regions on every one) added into engs[0]'s, device to device.
int gce_stats_payload_sum(gce_engine **engs, int32_t n_engs, const int64_t **payload, gce_payload_layout *layout) {
    if (!engs || n_engs < 1 || !payload || !layout || !engs[0]) return GCE_ERR_INVALID;
    gce_engine *e = engs[0];
    const int nt = (int)e->target_len.size();
    if (e->h_binoff.size() != (size_t)nt + 1 || !e->dp_depth.p) return fail(e, GCE_ERR_INVALID, "gce_stats_payload_sum before gce_stats_payload_device");
    const int64_t nbins = e->h_binoff[nt];
    // (the number of regions is what is left of the buffer behind the Stats blocks and the bins: every engine was given the same list)
    for (int r = 1; r < n_engs; r++) if (!engs[r] || engs[r]->h_binoff != e->h_binoff || !engs[r]->dp_depth.p || engs[r]->payload_words != e->payload_words) return fail(e, GCE_ERR_INVALID, "gce_stats_payload_sum: the engines' payloads differ in shape");
    const int64_t words = e->payload_words;
    (void)hipSetDevice(e->prm.device);
    hipStream_t s = e->stream;
    if (n_engs > 1) {
        DevBuf tmp; HIPCHK(tmp.ensure((size_t)words * 8 + 64));
        for (int r = 1; r < n_engs; r++) {
            gce_engine *x = engs[r];
            (void)hipSetDevice(x->prm.device); HIPCHK(hipStreamSynchronize(x->stream)); (void)hipSetDevice(e->prm.device);      // x's payload kernels are through
            const hipError_t ce = x->prm.device == e->prm.device ? hipMemcpyAsync(tmp.p, x->dp_depth.p, (size_t)words * 8, hipMemcpyDeviceToDevice, s)
                                                                  : hipMemcpyPeerAsync(tmp.p, e->prm.device, x->dp_depth.p, x->prm.device, (size_t)words * 8, s);
            if (ce != hipSuccess) return fail(e, GCE_ERR_HIP, "payload sum: device-to-device copy");
            hipLaunchKernelGGL(k_add_i64, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, e->dp_depth.as<long long>(), (const long long *)tmp.p, (int)words);
        }
        HIPCHK(hipStreamSynchronize(s));
        tmp.release();
    }
    layout->stats_words = 2 * GCE_STATS_WORDS; layout->n_targets = nt; layout->n_bins = nbins; layout->n_regions = (int32_t)((words - 2 * GCE_STATS_WORDS - 2 * nbins) / 2); layout->total_words = words;
    layout->bin_off = e->h_binoff.data();
    *payload = e->dp_depth.as<int64_t>();
    return GCE_OK;
}
// After every engine's gce_raw_build_output: the shards' record streams (each in bamComp order) merged into ONE stream in bamComp order over the
// whole file -- (tid, pos, mtid, mpos, isize, place in the input stream), gencore.h:19-47 -- in engs[0]'s output buffer, so that the writer
// streams it like a single engine's; the Stats blocks summed device to device into engs[0]'s (no host bounce of the blocks, no collective).
// The order is worked out on the host from 32 bytes per emitted record (runs of one shard are found by bisection: key ranges overlap only at
// their borders), the bytes are moved by the GPU.
int gce_raw_merge_outputs(gce_engine **engs, int32_t n_engs, uint64_t *body_bytes, int64_t *n_out_total, gce_stats *pre, gce_stats *post, int64_t *n_reads_total) {
    if (!engs || n_engs < 1 || !body_bytes) return GCE_ERR_INVALID;
    gce_engine *e = engs[0];
    for (int r = 0; r < n_engs; r++) if (!engs[r] || !engs[r]->raw_mode || !engs[r]->processed || engs[r]->dev_error) return GCE_ERR_INVALID;
    std::vector<std::vector<MergeKey>> keys((size_t)n_engs);
    std::vector<uint64_t> base((size_t)n_engs + 1, 0);                              // where shard r's stream lies in the staging buffer
    int64_t total_out = 0, total_reads = 0;
    for (int r = 0; r < n_engs; r++) {
        gce_engine *x = engs[r];
        (void)hipSetDevice(x->prm.device);
        const uint64_t no = (uint64_t)x->n_out;
        base[r + 1] = base[r] + x->raw_body_bytes; total_out += x->n_out; total_reads += x->n;
        keys[r].resize((size_t)no);
        if (!no) continue;
        if (x->sh_keys.ensure(no * sizeof(MergeKey)) != hipSuccess) return fail(e, GCE_ERR_OOM, "out of device memory");
        hipLaunchKernelGGL(k_merge_keys, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, x->stream, x->dev_batch.core, (const uint32_t *)x->o_src.p, x->shard_n >= 0 ? (const uint32_t *)x->sh_sel.p : (const uint32_t *)nullptr,
                           (const uint64_t *)x->rw_rsize.p, no, x->sh_keys.as<MergeKey>());
        if (hipMemcpyAsync(keys[r].data(), x->sh_keys.p, no * sizeof(MergeKey), hipMemcpyDeviceToHost, x->stream) != hipSuccess || hipStreamSynchronize(x->stream) != hipSuccess) return fail(e, GCE_ERR_HIP, "merge keys");
    }
    const uint64_t total = base[n_engs];
    // ---- the order: smallest head first, then as many records of that shard as stay in front of every other head
    std::vector<uint64_t> from((size_t)std::max<int64_t>(total_out, 1)), to((size_t)std::max<int64_t>(total_out, 1)); std::vector<uint32_t> size((size_t)std::max<int64_t>(total_out, 1));
    auto less = [](const MergeKey &a, const MergeKey &b) {
        if (a.tid != b.tid) return a.tid < b.tid;
        if (a.pos != b.pos) return a.pos < b.pos;
        if (a.mtid != b.mtid) return a.mtid < b.mtid;
        if (a.mpos != b.mpos) return a.mpos < b.mpos;
        if (a.isize != b.isize) return a.isize < b.isize;
        return a.gidx < b.gidx;
    };
    {
        std::vector<size_t> head((size_t)n_engs, 0); std::vector<uint64_t> soff((size_t)n_engs, 0);
        int64_t row = 0; uint64_t at = 0;
        while (row < total_out) {
            int best = -1, second = -1;
            for (int r = 0; r < n_engs; r++) if (head[r] < keys[r].size()) {
                if (best < 0 || less(keys[r][head[r]], keys[best][head[best]])) { second = best; best = r; }
                else if (second < 0 || less(keys[r][head[r]], keys[second][head[second]])) second = r;
            }
            size_t stop = keys[best].size();
            if (second >= 0) {                                                      // the first record of `best` that is not in front of `second`'s head (the shard's table is sorted)
                const MergeKey &lim = keys[second][head[second]];
                size_t a = head[best] + 1, z = keys[best].size();
                while (a < z) { const size_t mid = (a + z) >> 1; if (less(keys[best][mid], lim)) a = mid + 1; else z = mid; }
                stop = a;
            }
            for (size_t k = head[best]; k < stop; k++, row++) { const uint32_t sz = keys[best][k].size; from[row] = base[best] + soff[best]; to[row] = at; size[row] = sz; soff[best] += sz; at += sz; }
            head[best] = stop;
        }
        if (at != total) return fail(e, GCE_ERR_INVALID, "merge: record sizes do not add up");
    }
    // ---- the bytes: every shard's stream staged on engs[0]'s device, one gather into the merged stream
    (void)hipSetDevice(e->prm.device);
    hipStream_t s = e->stream;
    if (e->sh_stage.ensure(total + 64) != hipSuccess) return fail(e, GCE_ERR_OOM, "out of device memory");
    for (int r = 0; r < n_engs; r++) {
        gce_engine *x = engs[r];
        if (!x->raw_body_bytes) continue;
        const hipError_t ce = x->prm.device == e->prm.device ? hipMemcpyAsync((char *)e->sh_stage.p + base[r], x->rw_body.p, x->raw_body_bytes, hipMemcpyDeviceToDevice, s)
                                                              : hipMemcpyPeerAsync((char *)e->sh_stage.p + base[r], e->prm.device, x->rw_body.p, x->prm.device, x->raw_body_bytes, s);
        if (ce != hipSuccess) return fail(e, GCE_ERR_HIP, "merge: device-to-device copy");
    }
    HIPCHK(hipStreamSynchronize(s));                                                 // (engs[0]'s own stream was the source of one of the copies: it may be overwritten from here on)
    if (total_out > 0) {
        DevBuf d_from, d_to, d_size;
        HIPCHK(d_from.ensure((size_t)total_out * 8)); HIPCHK(d_to.ensure((size_t)total_out * 8)); HIPCHK(d_size.ensure((size_t)total_out * 4));
        HIPCHK(hipMemcpyAsync(d_from.p, from.data(), (size_t)total_out * 8, hipMemcpyHostToDevice, s)); HIPCHK(hipMemcpyAsync(d_to.p, to.data(), (size_t)total_out * 8, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_size.p, size.data(), (size_t)total_out * 4, hipMemcpyHostToDevice, s));
        HIPCHK(e->rw_body.ensure(total + 64));
        hipLaunchKernelGGL(k_merge_copy, dim3((unsigned)std::min<uint64_t>(((uint64_t)total_out + 15) / 16, 65535u)), dim3(256), 0, s, (const uint8_t *)e->sh_stage.p, (const uint64_t *)d_from.p, (const uint64_t *)d_to.p,
                           (const uint32_t *)d_size.p, (uint64_t)total_out, e->rw_body.as<uint8_t>());
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipGetLastError());
        d_from.release(); d_to.release(); d_size.release();
    }
    e->sh_stage.release();
    e->raw_body_bytes = total;
    // ---- Stats: the shards' blocks added into engs[0]'s, device to device
    const int W2 = 2 * GCE_STATS_WORDS;
    if (n_engs > 1) {
        DevBuf tmp; HIPCHK(tmp.ensure((size_t)W2 * 8));
        long long *acc = (long long *)e->si.as<StreamInfo>()->pre;
        for (int r = 1; r < n_engs; r++) {
            gce_engine *x = engs[r];
            const void *src = (const void *)x->si.as<StreamInfo>()->pre;
            const hipError_t ce = x->prm.device == e->prm.device ? hipMemcpyAsync(tmp.p, src, (size_t)W2 * 8, hipMemcpyDeviceToDevice, s) : hipMemcpyPeerAsync(tmp.p, e->prm.device, src, x->prm.device, (size_t)W2 * 8, s);
            if (ce != hipSuccess) return fail(e, GCE_ERR_HIP, "merge: Stats copy");
            hipLaunchKernelGGL(k_add_i64, dim3((W2 + 255) / 256), dim3(256), 0, s, acc, (const long long *)tmp.p, W2);
        }
        HIPCHK(hipStreamSynchronize(s));
        tmp.release();
    }
    static_assert(offsetof(StreamInfo, post) == offsetof(StreamInfo, pre) + GCE_STATS_WORDS * 8, "pre and post lie back to back");
    long long both[2 * GCE_STATS_WORDS];
    HIPCHK(hipMemcpy(both, e->si.as<StreamInfo>()->pre, sizeof both, hipMemcpyDeviceToHost));
    if (pre) memcpy(pre, both, GCE_STATS_WORDS * 8);
    if (post) memcpy(post, both + GCE_STATS_WORDS, GCE_STATS_WORDS * 8);
    *body_bytes = total;
    if (n_out_total) *n_out_total = total_out;
    if (n_reads_total) *n_reads_total = total_reads;
    return GCE_OK;
}


// After gce_raw_build_output (or gce_raw_merge_outputs): the record stream compressed into BGZF blocks by the GPU (gce_deflate.hpp: greedy LZ77,
// fixed Huffman codes, one lane per block) -- replaces bgzf_write's deflate under sam_write1 (src/gencore.cpp:104).  *comp_bytes = size of the
// file image of the records (BGZF blocks back to back; the caller writes the BAM header's blocks in front and the EOF marker behind);
// gce_raw_read_deflated_async copies a piece of it to the host.
int gce_raw_deflate_output(gce_engine *e, uint64_t *comp_bytes) {
    if (!e || !e->raw_mode || !comp_bytes) return GCE_ERR_INVALID;
    (void)hipSetDevice(e->prm.device);
    hipStream_t s = e->stream;
    const uint64_t total = e->raw_body_bytes;
    *comp_bytes = 0; e->zo_bytes = 0;
    if (!total) return GCE_OK;
    // input bytes per block: enough blocks for every CU's 32 lanes (one workgroup per CU: the hash tables fill its LDS), never more than BGZF's 65 280
    uint64_t blk = (total + 8191) / 8192; blk = (blk + 255) & ~(uint64_t)255;
    if (blk < 4096) blk = 4096; if (blk > 0xff00) blk = 0xff00;
    const uint64_t nb64 = (total + blk - 1) / blk;
    if (nb64 >= 0x7FFFFFF0ull) return fail(e, GCE_ERR_INVALID, "output stream too large for one deflate pass");
    const uint32_t nb = (uint32_t)nb64, slot = (uint32_t)(blk + blk / 8 + 64);
    HIPCHK(e->zo_slots.ensure((size_t)nb * slot + 64)); HIPCHK(e->zo_sizes.ensure(((size_t)nb + 1) * 4)); HIPCHK(e->zo_off.ensure(((size_t)nb + 1) * 8));
    hipLaunchKernelGGL(k_bgzf_deflate, dim3((nb + DEF_T - 1) / DEF_T), dim3(DEF_T), 0, s, (const uint8_t *)e->rw_body.p, total, (uint32_t)blk, nb, e->zo_slots.as<uint8_t>(), slot, e->zo_sizes.as<uint32_t>());
    HIPCHK(hipMemsetAsync((char *)e->zo_sizes.p + (size_t)nb * 4, 0, 4, s));
    HIPCHK(dev_exclusive_sum(e->zo_sizes.as<uint32_t>(), (uint64_t)nb, e->zo_off.as<uint64_t>(), e->rw_tmp, s));
    uint64_t csz = 0;
    HIPCHK(hipMemcpyAsync(&csz, e->zo_off.as<uint64_t>() + nb, 8, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
    HIPCHK(e->zo_out.ensure(csz + 64));
    hipLaunchKernelGGL(k_deflate_pack, dim3(std::min<uint32_t>((nb + 3) / 4, 16384u)), dim3(256), 0, s, (const uint8_t *)e->zo_slots.p, slot, (const uint32_t *)e->zo_sizes.p, (const uint64_t *)e->zo_off.p, nb, e->zo_out.as<uint8_t>());
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    e->zo_slots.release();
    e->zo_bytes = csz; *comp_bytes = csz;
    return GCE_OK;
}
int gce_raw_read_deflated_async(gce_engine *e, uint64_t offset, void *host, size_t bytes, int32_t *ticket) {
    if (!e || !e->raw_mode || offset + bytes > e->zo_bytes || (!host && bytes)) return GCE_ERR_INVALID;
    (void)hipSetDevice(e->prm.device);
    if (bytes) HIPCHK(hipMemcpyAsync(host, (const char *)e->zo_out.p + offset, bytes, hipMemcpyDeviceToHost, e->up_stream));
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev, e->up_stream));
    e->up_events.push_back(ev);
    if (ticket) *ticket = (int32_t)e->up_events.size() - 1;
    return GCE_OK;
}

// pinned host memory for the windows of the file path (the DMA engines copy from / to it without a staging copy)
int gce_host_alloc(size_t bytes, void **out) { if (!out) return GCE_ERR_INVALID; return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable) == hipSuccess ? GCE_OK : GCE_ERR_OOM; }      // (portable: the sharded runner's devices all copy from the same windows)
void gce_host_free(void *p) { if (p) (void)hipHostFree(p); }

}  // extern "C"
