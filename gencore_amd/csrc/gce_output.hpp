// gce_output.hpp — the output side of the path: the order of the reference's output set and the compact table of emitted records.
//
// Gencore::outputPair hands every record to a std::set ordered by bamComp (gencore.h:19-47: tid, pos, mtid, mpos, isize, then the
// heap address of the record — quirk Q3) and writeBam drains that set (gencore.cpp:72-143).  Here the emitted reads are already
// flagged per input read (out_flag) and compacted in input order (out_index), i.e. sorted by (tid, pos); what is left of bamComp
// is the order INSIDE a run of equal (tid, pos): (mtid, mpos, isize), ties by input index.  Runs are short (the reads of a few
// clusters that start on one position), so every record ranks itself by walking its run.
//   k_out_order   row of every emitted read; o_src[row], row_of[read]
//   k_out_rows    the table row by row (kind, qname source, NM, FR, RR, mate ROW) + the record's size in 16-byte units
//   k_u64_*       exclusive scan of the sizes -> offsets into the compact blobs
//   k_out_gather  bases and qualities of the emitted records, 16 lanes per record
// plus k_pack_reference: FastaReader::to4bits (fastareader.cpp:139-152) for a whole contig.
#pragma once

struct OutTable {
    uint32_t *src, *qname_src, *mate, *row_of; uint8_t *kind; int32_t *nm_new; int16_t *fr, *rr;
    uint64_t *units, *seq_off, *qual_off; uint8_t *seq, *qual;
};

// bamComp below (tid, pos): is a < b ?  (gencore.h:27-36; `ia < ib` stands in for the pointer comparison)
__device__ __forceinline__ bool out_less(const gce_core &a, uint32_t ia, const gce_core &b, uint32_t ib) {
    if (a.mtid != b.mtid) return a.mtid < b.mtid;
    if (a.mpos != b.mpos) return a.mpos < b.mpos;
    if (a.isize != b.isize) return a.isize < b.isize;
    return ia < ib;
}

__global__ __launch_bounds__(256) void k_out_order(DevBatch b, Work w, OutTable o) {
    const uint32_t n_out = (uint32_t)w.si->n_out;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_out; k += gridDim.x * blockDim.x) {
        const uint32_t i = w.out_index[k];
        const gce_core ci = b.core[i];
        uint32_t back = 0, less = 0;
        for (uint32_t j = k; j-- > 0;) {
            const uint32_t r = w.out_index[j];
            const gce_core cj = b.core[r];
            if (cj.tid != ci.tid || cj.pos != ci.pos) break;
            back++; less += out_less(cj, r, ci, i);
        }
        for (uint32_t j = k + 1; j < n_out; j++) {
            const uint32_t r = w.out_index[j];
            const gce_core cj = b.core[r];
            if (cj.tid != ci.tid || cj.pos != ci.pos) break;
            less += out_less(cj, r, ci, i);
        }
        const uint32_t row = k - back + less;
        o.src[row] = i; o.row_of[i] = row;
    }
}

__global__ __launch_bounds__(256) void k_out_rows(DevBatch b, Work w, OutTable o) {
    const uint32_t n_out = (uint32_t)w.si->n_out;
    for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < n_out; row += gridDim.x * blockDim.x) {
        const uint32_t i = o.src[row];
        const uint8_t kind = w.out_flag[i];
        OutRec r; r.qname_src = i; r.mate = NONE32; r.nm_new = -1; r.fr = -1; r.rr = -1; r.pad = 0;     // pass-through: written as it came
        if (kind == 1) r = w.orec[i];
        o.kind[row] = kind; o.qname_src[row] = r.qname_src; o.nm_new[row] = (int32_t)r.nm_new; o.fr[row] = r.fr; o.rr[row] = r.rr;
        o.mate[row] = r.mate == NONE32 ? NONE32 : o.row_of[r.mate];
        const uint32_t lq = (uint32_t)b.core[i].l_qseq;
        o.units[row] = ((uint64_t)(((lq + 1) / 2 + 15) / 16) << 32) | (uint64_t)((lq + 15) / 16);
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
// applies the scan of the record sizes: byte offsets of every row in the two compact blobs
__global__ __launch_bounds__(256) void k_out_offsets(OutTable o, const unsigned long long *n_ptr, const uint64_t *part) {
    __shared__ uint64_t s_w[4];
    __shared__ uint64_t s_carry;
    const uint64_t n = *n_ptr, base = (uint64_t)blockIdx.x * SCAN_TILE;
    if (base >= n) return;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = part[blockIdx.x];
    __syncthreads();
    for (int k = 0; k < SCAN_TILE / 256; k++) {
        const uint64_t i = base + k * 256 + threadIdx.x;
        uint64_t v = i < n ? o.units[i] : 0, x = v;
        for (int q = 1; q < 64; q <<= 1) { uint64_t t = (uint64_t)__shfl_up((long long)x, q); if (lane >= q) x += t; }
        if (lane == 63) s_w[wv] = x;
        __syncthreads();
        uint64_t woff = 0;
        for (int q = 0; q < wv; q++) woff += s_w[q];
        const uint64_t carry = s_carry, ex = carry + woff + x - v;
        if (i < n) { o.seq_off[i] = (ex >> 32) * 16; o.qual_off[i] = (ex & 0xFFFFFFFFull) * 16; }
        __syncthreads();
        if (threadIdx.x == 255) s_carry = carry + woff + x;
        __syncthreads();
    }
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
    for (uint32_t row = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; row < n_out; row += (gridDim.x * blockDim.x) >> 4) {
        const uint32_t i = o.src[row];
        const uint32_t lq = (uint32_t)b.core[i].l_qseq, su = ((lq + 1) / 2 + 15) / 16, qu = (lq + 15) / 16;
        const uint8_t *ss = b.seq + b.seq_off[i], *qs = b.qual + b.qual_off[i];
        uint8_t *sd = o.seq + o.seq_off[row], *qd = o.qual + o.qual_off[row];
        for (uint32_t u = sub; u < su + qu; u += 16) {
            if (u < su) *(uint4 *)(sd + 16 * u) = ld16_unaligned(ss + 16 * u);
            else *(uint4 *)(qd + 16 * (u - su)) = ld16_unaligned(qs + 16 * (u - su));
        }
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
