// gce_output.hpp — the output side of the path: the order of the reference's output set and the compact table of emitted records.
//
// Gencore::outputPair hands every record to a std::set ordered by bamComp (gencore.h:19-47: tid, pos, mtid, mpos, isize, then the
// heap address of the record — quirk Q3) and writeBam drains that set (gencore.cpp:72-143).  Here the emitted reads are flagged per
// input read (out_flag); listed in input order they are sorted by (tid, pos) already, and what is left of bamComp is the order INSIDE
// a run of equal (tid, pos): (mtid, mpos, isize), ties by input index.  Runs are short (the reads of a few clusters that start on one
// position), so every record ranks itself by walking its run.
//   k_out_reduce   per tile of 4096 reads: emitted records, 16-byte units of their bases / qualities
//   k_out_partials exclusive scan of those triples (one block) -> n_out, blob sizes
//   k_out_meta     THE gather of per-record facts, once: for every emitted read k (input order) its key record, what outputPair noted
//                  (OutRec), where its bases / qualities go in the compact blobs, writeBam's Stats::addRead (stats.cpp:101-121) --
//                  compact arrays indexed by k, so that the passes below touch no per-read array again (round 2 gathered the 32-byte key
//                  record of every emitted read in four kernels: order, rows, Stats, gather: a 64-byte sector each time)
//   k_out_rows     bamComp rank inside the (tid, pos) run from the compact keys (neighbours are neighbours in memory) -> the table row
//                  by row; row_of[read]
//   k_out_gather   bases and qualities of the emitted records, 16 lanes per record, blobs laid out in INPUT order (sources and
//                  destinations both walk forward)
//   k_out_mate     mate read -> mate row
// plus k_pack_reference: FastaReader::to4bits (fastareader.cpp:139-152) for a whole contig.
#pragma once

struct __attribute__((aligned(16))) OutKey { int32_t tid, pos, mtid, mpos, isize; uint32_t read; int32_t lq; uint32_t kind; };
static_assert(sizeof(OutKey) == 32, "OutKey must stay 32 bytes");
struct OutTable {
    uint32_t *src, *qname_src, *mate; uint8_t *kind; int32_t *nm_new; int16_t *fr, *rr;
    uint64_t *seq_off, *qual_off; uint8_t *seq, *qual;
    // indexed by k (emitted reads in input order)
    OutKey *key; OutRec *rec; uint64_t *ksoff, *kqoff; uint32_t *krow;
    uint32_t *rank64;                    // [reads / 64]: emitted reads in front of read 64 q (k_out_meta -> k_out_mate)
    uint64_t *part3;                     // [tiles][3]: records, base units, quality units
};
#define GCE_POST_SLOTS 64

// bamComp below (tid, pos): is a < b ?  (gencore.h:27-36; `ia < ib` stands in for the pointer comparison)
__device__ __forceinline__ bool out_less(const OutKey &a, const OutKey &b) {
    if (a.mtid != b.mtid) return a.mtid < b.mtid;
    if (a.mpos != b.mpos) return a.mpos < b.mpos;
    if (a.isize != b.isize) return a.isize < b.isize;
    return a.read < b.read;
}
__device__ __forceinline__ uint64_t out_units_of(uint32_t lq) { return ((uint64_t)(((lq + 1) / 2 + 15) / 16) << 32) | (uint64_t)((lq + 15) / 16); }

// Tiles of OUT_TILE reads, eight CONSECUTIVE reads per thread: their flags are one 8-byte load, the thread's emitted reads (one in
// eight at the benchmark's depth) are dealt with one after the other, and a tile needs two barriers (a loop over 256-read slices with
// a block scan each was 24 barriers per 2048 reads: 261 us for the pass that is now k_out_meta).
#define OUT_T 512
#define OUT_TILE (OUT_T * 8)
__device__ __forceinline__ uint64_t out_flags8(const Work &w, uint64_t i0, uint64_t n) {
    uint64_t f = 0;
    if (i0 + 8 <= n) f = *reinterpret_cast<const uint64_t *>(w.out_flag + i0);             // (i0 is a multiple of 8; the flag array is 8-byte aligned)
    else for (int j = 0; j < 8; j++) if (i0 + j < n) f |= (uint64_t)w.out_flag[i0 + j] << (8 * j);
    return f;
}
__global__ __launch_bounds__(OUT_T) void k_out_reduce(DevBatch b, Work w, OutTable o) {
    __shared__ uint64_t s[OUT_T / 64][2];
    const uint64_t i0 = (uint64_t)blockIdx.x * OUT_TILE + 8 * (uint64_t)threadIdx.x;
    const uint64_t f8 = out_flags8(w, i0, (uint64_t)b.n);
    uint64_t cnt = 0, un = 0;
    const int lq_u = w.si->lq_min == w.si->lq_max ? w.si->lq_max : -1;                   // every read that can be emitted has this length (k_describe): no gather of lengths
#pragma unroll
    for (int j = 0; j < 8; j++) if ((f8 >> (8 * j)) & 0xFF) { cnt++; un += out_units_of(lq_u >= 0 ? (uint32_t)lq_u : (uint32_t)b.core[i0 + j].l_qseq); }
    cnt = (uint64_t)wave_sum64((long long)cnt); un = (uint64_t)wave_sum64((long long)un);       // (halves of `un` < 2^32 each: no carry)
    if (lane_id() == 0) { s[threadIdx.x >> 6][0] = cnt; s[threadIdx.x >> 6][1] = un; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t c = 0, u = 0;
        for (int q = 0; q < OUT_T / 64; q++) { c += s[q][0]; u += s[q][1]; }
        o.part3[3 * (uint64_t)blockIdx.x] = c; o.part3[3 * (uint64_t)blockIdx.x + 1] = u >> 32; o.part3[3 * (uint64_t)blockIdx.x + 2] = u & 0xFFFFFFFFull;
    }
}
// exclusive scan of the tile triples, one block; totals -> n_out, out_units
__global__ __launch_bounds__(1024) void k_out_partials(OutTable o, uint64_t nparts, Work w) {
    __shared__ uint64_t s_w[16][3];
    __shared__ uint64_t s_carry[3];
    if (threadIdx.x < 3) s_carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    for (uint64_t base = 0; base < nparts; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        uint64_t v[3], x[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            v[c] = i < nparts ? o.part3[3 * i + c] : 0; x[c] = v[c];
            for (int q = 1; q < 64; q <<= 1) { const uint64_t t = (uint64_t)__shfl_up((long long)x[c], q); if (lane >= q) x[c] += t; }
            if (lane == 63) s_w[wv][c] = x[c];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; c++) {
            uint64_t woff = 0;
            for (int k = 0; k < wv; k++) woff += s_w[k][c];
            if (i < nparts) o.part3[3 * i + c] = s_carry[c] + woff + x[c] - v[c];
            x[c] += woff;
        }
        __syncthreads();
        if (threadIdx.x == 1023) { s_carry[0] += x[0]; s_carry[1] += x[1]; s_carry[2] += x[2]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { w.si->n_out = s_carry[0]; w.si->out_units = (s_carry[1] << 32) | (s_carry[2] & 0xFFFFFFFFull); }
}

// Round 5: the gather runs over the tile's COMPACTED list.  A thread used to deal with the emitted reads among its eight one after the other -- eight rounds of
// (scattered loads -> stores) per wave, each waiting for its loads: 239 us for 0.65 GB.  Now the tile's emitted reads are listed in LDS behind the scan (their place
// in the tile, their kind, where their bytes go when the lengths differ), thread t takes the t-th of them: one round of loads in flight per 512 records, stores to
// consecutive k.  UNIFORM (every emitted read has the same length: the host's look at k_describe's range): no offset lists, 12 KB of LDS instead of 44.
// rank64[q] = emitted reads in front of read 64 q: k_out_mate finds a mate's place in the list with it (two loads instead of a bisection of 21 dependent probes).
template <bool UNIFORM>
__global__ __launch_bounds__(OUT_T) void k_out_meta(DevBatch b, Work w, OutTable o, int lq_uniform) {
    __shared__ uint64_t s_w[OUT_T / 64][3];
    __shared__ long long s_stat[OUT_T / 64][6];
    __shared__ uint16_t s_idx[OUT_TILE];
    __shared__ uint8_t s_kind[OUT_TILE];
    __shared__ uint32_t s_so[UNIFORM ? 1 : OUT_TILE], s_qo[UNIFORM ? 1 : OUT_TILE];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint64_t tile0 = (uint64_t)blockIdx.x * OUT_TILE, i0 = tile0 + 8 * (uint64_t)threadIdx.x;
    const uint64_t f8 = out_flags8(w, i0, (uint64_t)b.n);
    uint32_t lqs[8]; uint64_t cnt = 0, us = 0, uq = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        lqs[j] = 0;
        if ((f8 >> (8 * j)) & 0xFF) { lqs[j] = UNIFORM ? (uint32_t)lq_uniform : (uint32_t)b.core[i0 + j].l_qseq; cnt++; const uint64_t un = out_units_of(lqs[j]); us += un >> 32; uq += un & 0xFFFFFFFFull; }
    }
    uint64_t xc = cnt, xs = us, xq = uq;                                                  // inclusive wave scans
    for (int q = 1; q < 64; q <<= 1) {
        const uint64_t t0 = (uint64_t)__shfl_up((long long)xc, q), t1 = (uint64_t)__shfl_up((long long)xs, q), t2 = (uint64_t)__shfl_up((long long)xq, q);
        if (lane >= q) { xc += t0; xs += t1; xq += t2; }
    }
    if (lane == 63) { s_w[wv][0] = xc; s_w[wv][1] = xs; s_w[wv][2] = xq; }
    __syncthreads();
    const uint64_t kbase = o.part3[3 * (uint64_t)blockIdx.x], sbase = o.part3[3 * (uint64_t)blockIdx.x + 1], qbase = o.part3[3 * (uint64_t)blockIdx.x + 2];
    uint32_t kr = (uint32_t)(xc - cnt), sr = (uint32_t)(xs - us), qr = (uint32_t)(xq - uq), total = 0;      // places inside the tile
    for (int q = 0; q < OUT_T / 64; q++) { const uint32_t c0 = (uint32_t)s_w[q][0]; if (q < wv) { kr += c0; sr += (uint32_t)s_w[q][1]; qr += (uint32_t)s_w[q][2]; } total += c0; }
    if ((threadIdx.x & 7) == 0 && i0 < (uint64_t)b.n) o.rank64[i0 >> 6] = (uint32_t)(kbase + kr);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t kind = (uint32_t)(f8 >> (8 * j)) & 0xFFu;
        if (!kind) continue;
        s_idx[kr] = (uint16_t)(8 * threadIdx.x + j); s_kind[kr] = (uint8_t)kind;
        if (!UNIFORM) { s_so[kr] = sr; s_qo[kr] = qr; const uint64_t un = out_units_of(lqs[j]); sr += (uint32_t)(un >> 32); qr += (uint32_t)un; }
        kr++;
    }
    __syncthreads();
    long long st[6] = {0, 0, 0, 0, 0, 0};                     // writeBam -> mPostStats->addRead: reads, bases, unmapped reads / bases, mismatches, reads with mismatches
    const uint64_t un_u = out_units_of((uint32_t)lq_uniform);
    for (uint32_t t2 = threadIdx.x; t2 < total; t2 += OUT_T) {
        const uint64_t i = tile0 + s_idx[t2];
        const uint32_t kind = s_kind[t2];
        const uint64_t kk = kbase + t2;
        union { gce_core c; uint4 q[2]; } t; const uint4 *src = reinterpret_cast<const uint4 *>(b.core + i); t.q[0] = src[0]; t.q[1] = src[1];
        const uint32_t nx = w.nmx[i];                                                      // k_describe: NM and "NM present" in one word
        union { OutRec r; uint4 q; } rc;
        rc.r.qname_src = (uint32_t)i; rc.r.mate = NONE32; rc.r.nm_new = -1; rc.r.fr = -1; rc.r.rr = -1; rc.r.pad = 0;                  // pass-through: written as it came
        if (kind == 1) rc.q = *reinterpret_cast<const uint4 *>(w.orec + i);
        union { OutKey k; uint4 q[2]; } key;
        key.k.tid = t.c.tid; key.k.pos = t.c.pos; key.k.mtid = t.c.mtid; key.k.mpos = t.c.mpos; key.k.isize = t.c.isize; key.k.read = (uint32_t)i; key.k.lq = t.c.l_qseq; key.k.kind = kind;
        const bool mapped = t.c.tid >= 0;
        const int nm = rc.r.nm_new >= 0 ? (int)rc.r.nm_new : (int)(nx >> 1);
        const int mism = (mapped && (nx & 1u)) ? nm : 0;
        st[0] += 1; st[1] += t.c.l_qseq; st[4] += mism;
        if (!mapped) { st[2] += 1; st[3] += t.c.l_qseq; }
        if (mism > 0) st[5] += 1;
        w.out_index[kk] = (uint32_t)i;
        reinterpret_cast<uint4 *>(o.key + kk)[0] = key.q[0]; reinterpret_cast<uint4 *>(o.key + kk)[1] = key.q[1];
        *reinterpret_cast<uint4 *>(o.rec + kk) = rc.q;
        const uint64_t so = UNIFORM ? sbase + (uint64_t)t2 * (un_u >> 32) : sbase + s_so[t2], qo = UNIFORM ? qbase + (uint64_t)t2 * (un_u & 0xFFFFFFFFull) : qbase + s_qo[t2];
        o.ksoff[kk] = so * 16; o.kqoff[kk] = qo * 16;
    }
    for (int k = 0; k < 6; k++) { const long long v = wave_sum64(st[k]); if (lane == 0) s_stat[wv][k] = v; }
    __syncthreads();
    if (threadIdx.x < 6) {
        long long v = 0;
        for (int q = 0; q < OUT_T / 64; q++) v += s_stat[q][threadIdx.x];
        if (v) atomicAdd((unsigned long long *)&w.si->post_slot[blockIdx.x & (GCE_POST_SLOTS - 1)][threadIdx.x], (unsigned long long)v);
    }
}

__global__ __launch_bounds__(256) void k_out_rows(Work w, OutTable o) {
    const uint32_t n_out = (uint32_t)w.si->n_out;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_out; k += gridDim.x * blockDim.x) {
        union { OutKey k; uint4 q[2]; } me; me.q[0] = reinterpret_cast<const uint4 *>(o.key + k)[0]; me.q[1] = reinterpret_cast<const uint4 *>(o.key + k)[1];
        uint32_t back = 0, less = 0;
        for (uint32_t j = k; j-- > 0;) {
            union { OutKey k; uint4 q[2]; } ot; ot.q[0] = reinterpret_cast<const uint4 *>(o.key + j)[0]; ot.q[1] = reinterpret_cast<const uint4 *>(o.key + j)[1];
            if (ot.k.tid != me.k.tid || ot.k.pos != me.k.pos) break;
            back++; less += out_less(ot.k, me.k);
        }
        for (uint32_t j = k + 1; j < n_out; j++) {
            union { OutKey k; uint4 q[2]; } ot; ot.q[0] = reinterpret_cast<const uint4 *>(o.key + j)[0]; ot.q[1] = reinterpret_cast<const uint4 *>(o.key + j)[1];
            if (ot.k.tid != me.k.tid || ot.k.pos != me.k.pos) break;
            less += out_less(ot.k, me.k);
        }
        const uint32_t row = k - back + less;
        union { OutRec r; uint4 q; } rc; rc.q = *reinterpret_cast<const uint4 *>(o.rec + k);
        o.src[row] = me.k.read; o.krow[k] = row;
        o.kind[row] = (uint8_t)me.k.kind; o.qname_src[row] = rc.r.qname_src; o.nm_new[row] = (int32_t)rc.r.nm_new; o.fr[row] = rc.r.fr; o.rr[row] = rc.r.rr;
        o.mate[row] = rc.r.mate;                               // still a READ (or NONE): k_out_mate turns it into the mate's row
        o.seq_off[row] = o.ksoff[k]; o.qual_off[row] = o.kqoff[k];
    }
}

// exclusive scan of packed (hi, lo) 32-bit counters held in uint64 (no carry between the halves while each total < 2^32);
// the element count lives on the device
__global__ __launch_bounds__(256) void k_u64_reduce(const uint64_t *v, const unsigned long long *n_ptr, uint64_t *part) {
    __shared__ uint64_t s[4];
    const uint64_t n = *n_ptr, base = (uint64_t)blockIdx.x * SCAN_TILE;
    if (base >= n) return;
    uint64_t x = 0;
    for (int k = 0; k < SCAN_TILE / 256; k++) { const uint64_t i = base + k * 256 + threadIdx.x; if (i < n) x += v[i]; }
    x = (uint64_t)wave_sum64((long long)x);
    if (lane_id() == 0) s[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(1024) void k_u64_partials(uint64_t *part, const unsigned long long *n_ptr, unsigned long long *total) {
    __shared__ uint64_t s_w[16];
    __shared__ uint64_t s_carry;
    const uint64_t nparts = (*n_ptr + SCAN_TILE - 1) / SCAN_TILE;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    for (uint64_t base = 0; base < nparts; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        uint64_t v = i < nparts ? part[i] : 0, x = v;
        for (int o = 1; o < 64; o <<= 1) { uint64_t t = (uint64_t)__shfl_up((long long)x, o); if (lane >= o) x += t; }
        if (lane == 63) s_w[wv] = x;
        __syncthreads();
        uint64_t woff = 0;
        for (int k = 0; k < wv; k++) woff += s_w[k];
        const uint64_t carry = s_carry;
        if (i < nparts) part[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}
// 16 bytes from an arbitrary byte address
__device__ __forceinline__ uint4 ld16_unaligned(const uint8_t *p_) {
    typedef uint64_t u64u __attribute__((aligned(1)));
    const uint64_t a = *(const u64u *)p_, c = *(const u64u *)(p_ + 8);
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)c, (uint32_t)(c >> 32));
}
// 16 lanes per record: 16-byte pieces of its bases, then of its qualities (sources are unaligned, destinations 16-byte aligned;
// the pad bytes behind a record's last base come from the bytes that follow it in the source blob, readable by contract)
__global__ __launch_bounds__(256) void k_out_gather(DevBatch b, Work w, OutTable o) {
    const uint32_t n_out = (uint32_t)w.si->n_out;
    const int sub = threadIdx.x & 15;
    for (uint32_t k = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; k < n_out; k += (gridDim.x * blockDim.x) >> 4) {
        const OutKey *ky = o.key + k;
        const uint32_t i = ky->read, lq = (uint32_t)ky->lq, su = ((lq + 1) / 2 + 15) / 16, qu = (lq + 15) / 16;
        uint64_t so, qo;
        if (ky->kind == 1) { const ReadDesc d = load_desc(w.rdesc, i); so = d.so; qo = d.qo; }     // (an emitted record is a template: its descriptor holds both offsets)
        else { so = b.seq_off[i]; qo = b.qual_off[i]; }
        const uint8_t *ss = b.seq + so, *qs = b.qual + qo;
        uint8_t *sd = o.seq + o.ksoff[k], *qd = o.qual + o.kqoff[k];
        for (uint32_t u = sub; u < su + qu; u += 16) {
            if (u < su) *(uint4 *)(sd + 16 * u) = ld16_unaligned(ss + 16 * u);
            else *(uint4 *)(qd + 16 * (u - su)) = ld16_unaligned(qs + 16 * (u - su));
        }
    }
}

// mate read -> mate row: the mate's place in the ascending list of emitted reads = emitted reads in front of its block of 64 (rank64, k_out_meta) + the
// non-zero flags of that block in front of it -- one 64-byte sector of flags and one word, both addressed by the read index itself (rounds 2-4: a bisection
// of out_index, 21 dependent probes per record: 42 us) --, then that record's row
__global__ __launch_bounds__(256) void k_out_mate(Work w, OutTable o) {
    const uint32_t n_out = (uint32_t)w.si->n_out;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_out; k += gridDim.x * blockDim.x) {
        const uint32_t mr = o.rec[k].mate;
        if (mr == NONE32) continue;                            // (k_out_rows wrote NONE into the row)
        const uint32_t blk = mr >> 6, within = mr & 63u;
        uint32_t km = o.rank64[blk];
        const uint64_t *f = reinterpret_cast<const uint64_t *>(w.out_flag + ((uint64_t)blk << 6));       // (the flag array is padded to a multiple of 64)
        uint64_t v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = f[q];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int nb = (int)within - 8 * q;                // bytes of this word in front of the mate
            if (nb <= 0) continue;
            uint64_t x = v[q];
            if (nb < 8) x &= (1ull << (8 * nb)) - 1ull;
            const uint64_t nz = (((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x) & 0x8080808080808080ull;
            km += (uint32_t)__popcll(nz);
        }
        o.mate[o.krow[k]] = o.krow[km];
    }
}

// FastaReader::to4bits (fastareader.cpp:139-152) + base2bits (:106-113): A=1, T=2, C=3, G=4, anything else 0; low nibble = even position
__global__ __launch_bounds__(256) void k_pack_reference(const char *bases, int64_t n, uint8_t *out) {
    const int64_t nb = (n + 1) / 2;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += (int64_t)gridDim.x * blockDim.x) {
        auto code = [](char c) -> uint32_t { return c == 'A' ? 1u : c == 'T' ? 2u : c == 'C' ? 3u : c == 'G' ? 4u : 0u; };
        const uint32_t lo = code(bases[2 * k]), hi = 2 * k + 1 < n ? code(bases[2 * k + 1]) : 0u;
        out[k] = (uint8_t)(lo | (hi << 4));
    }
}
