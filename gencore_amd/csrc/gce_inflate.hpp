// gce_inflate.hpp — BGZF members inflated on the GPU (replaces htslib's bgzf_read -> inflate under sam_read1,
// src/gencore.cpp:205-274): the file goes over PCIe compressed (a quarter of the bytes) and the host inflates nothing but the header.
//
// A BGZF member is an independent raw-deflate stream of <= 64 KB (RFC 1951 inside the gzip framing of SAM spec 4.1), so the file is tens
// of thousands of independent serial problems: ONE LANE PER MEMBER, a workgroup = 32 members (half a wave).  Per lane:
//   * a 64-bit bit buffer refilled with one unaligned 8-byte load per symbol (a literal / length code, its extra bits, a distance code and
//     its extra bits are at most 48 bits);
//   * canonical Huffman decoding without lookup tables (the count of codes per length, 15 x 10 bits and 15 x 6 bits, lives in registers;
//     only the symbol permutation -- 288 + 32 entries of 16 bits -- lies in LDS, one column per lane): a table of 2^9 entries per lane
//     would be 64 KB per wave;
//   * stored, fixed and dynamic blocks, any number of them per member;
//   * output straight into the raw stream in HBM: literals byte by byte, matches eight bytes at a time when the distance allows it (a lane
//     reads back what it wrote itself: ordinary program order);
//   * CRC-32 of the member (slicing-by-8, tables in LDS, shared by the wave) and its ISIZE are checked; a member that fails any check raises
//     the launch's error flag and the run fails (GCE_ERR_INVALID from gce_raw_finish; gce_run_bam with GCE_BAM_HOST_INFLATE=1 or
//     gce_run_bam_hostcodec decode the same file on the host, with zlib as arbiter).
// Every input read is bounded by the member (the staging buffer is padded by 16 bytes), every output write by ISIZE.
#pragma once

#ifndef INF_T
#define INF_T 32            // members per workgroup (half a wave: less divergence than 64, measured 49.7 vs 53.7 ms on the cfg3 file; 16: 79.7)
#endif
#define INF_NSYM 320                         // 288 literal / length symbols, then 32 distance symbols
struct InfDir { uint64_t coff, uoff; uint32_t csize, usize; };

namespace {

__device__ __constant__ uint16_t INF_LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ __constant__ uint8_t INF_LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ __constant__ uint16_t INF_DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ __constant__ uint8_t INF_DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ __constant__ uint8_t INF_CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

typedef uint64_t inf_u64u __attribute__((aligned(1)));

struct InfBits {
    const uint8_t *p, *end;                  // next byte to move into the buffer; end of the member's deflate data
    uint64_t buf; int cnt;
    // the stream is consumed strictly forward, but where a refill reads depends on the bits consumed so far -- one load per symbol on the
    // critical path, a microsecond each.  Three aligned words ahead are kept in registers (q0 holds *p): the word a refill needs was asked
    // for two words (five to ten symbols) earlier.
    uint64_t q0, q1, q2; const uint64_t *wp;
    __device__ __forceinline__ void start(const uint8_t *at) {
        const uint64_t *a = reinterpret_cast<const uint64_t *>(reinterpret_cast<uintptr_t>(at) & ~(uintptr_t)7);
        q0 = a[0]; q1 = a[1]; q2 = a[2]; wp = a + 3; p = at; buf = 0; cnt = 0;
    }
    __device__ __forceinline__ void refill() {                                       // afterwards cnt >= 56 (bytes past `end` are padding / the next member: never consumed, see over())
        const int sh = 8 * (int)(reinterpret_cast<uintptr_t>(p) & 7);
        const uint64_t w = sh ? (q0 >> sh) | (q1 << (64 - sh)) : q0;
        buf |= w << cnt;
        const int adv = (63 - cnt) >> 3;
        cnt += adv * 8;
        const uint8_t *np = p + adv;
        if ((reinterpret_cast<uintptr_t>(np) ^ reinterpret_cast<uintptr_t>(p)) & ~(uintptr_t)7) { q0 = q1; q1 = q2; q2 = *wp++; }      // (adv <= 7: at most one word boundary)
        p = np;
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    __device__ __forceinline__ void drop(int n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
    __device__ __forceinline__ bool over() const { return p - (cnt >> 3) > end; }     // more bits consumed than the member holds
};

// the counts of codes per length 1..15, ten (literal / length) or six (distance) bits each, three or five per register
struct InfCnt { uint32_t r[5]; };
template <int BITS, int PER> __device__ __forceinline__ uint32_t inf_cnt(const InfCnt &c, int len /* 1..15, a compile-time constant after unrolling */) {
    return (c.r[(len - 1) / PER] >> (BITS * ((len - 1) % PER))) & ((1u << BITS) - 1u);
}

// one symbol of a canonical code (the decode loop of Mark Adler's puff.c, over bits already in the buffer); -1: not a code
template <int BITS, int PER>
__device__ __forceinline__ int inf_decode(InfBits &in, const InfCnt &c, const uint16_t *sym /* this lane's column: stride INF_T */) {
    int code = 0, first = 0, index = 0;
    uint64_t b = in.buf;
#pragma unroll
    for (int len = 1; len <= 15; len++) {
        code |= (int)(b & 1u); b >>= 1;
        const int count = (int)inf_cnt<BITS, PER>(c, len);
        if (code - count < first) { in.drop(len); return sym[(index + (code - first)) * INF_T]; }
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -1;
}

// lengths[0..n) (this member's scratch in device memory) -> counts + symbol permutation; false: over-subscribed or (for more than one code) incomplete
template <int BITS, int PER>
__device__ bool inf_construct(const uint8_t *len, int n, InfCnt &c, uint16_t *sym, uint16_t *offs /* 16 entries, column */) {
    for (int l = 0; l <= 15; l++) offs[l * INF_T] = 0;
    for (int s = 0; s < n; s++) offs[len[s] * INF_T]++;                       // (counts, parked in offs)
    int left = 1; uint32_t cnt[16];
    cnt[0] = offs[0];
#pragma unroll
    for (int l = 1; l <= 15; l++) { cnt[l] = offs[l * INF_T]; left <<= 1; left -= (int)cnt[l]; if (left < 0) return false; }
    for (int k = 0; k < 5; k++) c.r[k] = 0;
#pragma unroll
    for (int l = 1; l <= 15; l++) c.r[(l - 1) / PER] |= cnt[l] << (BITS * ((l - 1) % PER));
    uint32_t o = 0;
#pragma unroll
    for (int l = 1; l <= 15; l++) { offs[l * INF_T] = (uint16_t)o; o += cnt[l]; }
    for (int s = 0; s < n; s++) { const int l = len[s]; if (l) { sym[offs[l * INF_T] * INF_T] = (uint16_t)s; offs[l * INF_T]++; } }
    return left == 0 || (int)cnt[0] + 1 >= n;                                         // complete, or a single code (RFC 1951 allows one distance code of one bit)
}

__device__ __forceinline__ uint32_t inf_crc_word(const uint32_t (*tab)[256], uint32_t crc, uint64_t w) {
    const uint32_t lo = (uint32_t)w ^ crc, hi = (uint32_t)(w >> 32);
    return tab[7][lo & 0xFF] ^ tab[6][(lo >> 8) & 0xFF] ^ tab[5][(lo >> 16) & 0xFF] ^ tab[4][lo >> 24] ^ tab[3][hi & 0xFF] ^ tab[2][(hi >> 8) & 0xFF] ^ tab[1][(hi >> 16) & 0xFF] ^ tab[0][hi >> 24];
}

}  // namespace

// One lane per member.  err[0] |= 1 if any member is damaged (its number goes to err[1] by atomicMin).
// lens: INF_NSYM bytes of scratch per member (the code lengths of a dynamic block while its tables are built: in LDS they were 20 KB of the 70 KB
// that allowed two workgroups per CU -- 580 workgroups of a 565 MB file then ran in two rounds)
__global__ __launch_bounds__(INF_T) void k_bgzf_inflate(const uint8_t *comp, const InfDir *dir, uint32_t n_members, uint8_t *out, unsigned int *err, uint8_t *lens) {
    __shared__ uint16_t s_sym[INF_NSYM][INF_T];
    __shared__ uint16_t s_off[16][INF_T];
    __shared__ uint32_t s_crc[8][256];
    const int lane = threadIdx.x;
    for (int k = lane; k < 256; k += INF_T) {                                         // CRC-32 (reflected 0xEDB88320), slicing-by-8 tables
        uint32_t c = (uint32_t)k;
        for (int j = 0; j < 8; j++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        s_crc[0][k] = c;
    }
    __syncthreads();
    for (int k = lane; k < 256; k += INF_T) { uint32_t c = s_crc[0][k]; for (int t = 1; t < 8; t++) { c = s_crc[0][c & 0xFF] ^ (c >> 8); s_crc[t][k] = c; } }
    __syncthreads();
    const uint32_t m = blockIdx.x * INF_T + lane;
    if (m >= n_members) return;
    const InfDir d = dir[m];
    const uint8_t *src = comp + d.coff;
    uint8_t *o = out + d.uoff;
    const uint32_t usize = d.usize;
    bool ok = d.csize >= 26;
    uint32_t pos = 0;
    if (ok) {
        const uint32_t xlen = (uint32_t)src[10] | (uint32_t)src[11] << 8;
        ok = 12u + xlen + 8u <= d.csize;
        if (ok) {
            InfBits in; in.end = src + d.csize - 8; in.start(src + 12 + xlen);
            uint16_t *sym_l = &s_sym[0][lane], *sym_d = &s_sym[288][lane], *offs = &s_off[0][lane];
            uint8_t *len = lens + (size_t)m * INF_NSYM;
            InfCnt cl, cd;
            int last = 0;
            while (ok && !last) {
                in.refill();
                last = (int)in.take(1);
                const int type = (int)in.take(2);
                if (type == 0) {                                                      // stored: to the byte boundary, LEN, ~LEN, bytes
                    in.drop(in.cnt & 7);
                    in.refill();
                    const uint32_t ln = in.take(16), nl = in.take(16);
                    if ((ln ^ nl) != 0xFFFFu || pos + ln > usize) { ok = false; break; }
                    const uint8_t *q = in.p - (in.cnt >> 3);                           // the buffer holds whole bytes here
                    if (q + ln > in.end) { ok = false; break; }
                    for (uint32_t k = 0; k < ln; k++) o[pos + k] = q[k];
                    pos += ln;
                    in.start(q + ln);
                    continue;
                }
                if (type == 3) { ok = false; break; }
                if (type == 1) {                                                      // fixed codes (RFC 1951 3.2.6)
                    for (int s = 0; s < 288; s++) len[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                    for (int s = 0; s < 30; s++) len[(288 + s)] = 5;
                    if (!inf_construct<10, 3>(len, 288, cl, sym_l, offs)) { ok = false; break; }
                    (void)inf_construct<6, 5>(len + 288, 30, cd, sym_d, offs);
                } else {                                                              // dynamic codes (3.2.7)
                    const int nlen = (int)in.take(5) + 257, ndist = (int)in.take(5) + 1, ncode = (int)in.take(4) + 4;
                    if (nlen > 286 || ndist > 30) { ok = false; break; }
                    for (int k = 0; k < 19; k++) len[k] = 0;
                    for (int k = 0; k < ncode; k++) { if (in.cnt < 3) in.refill(); len[INF_CLORD[k]] = (uint8_t)in.take(3); }
                    InfCnt cc;
                    uint16_t *sym_c = sym_d;                                          // the code length code's permutation: in the distance part, which is rebuilt below
                    if (!inf_construct<10, 3>(len, 19, cc, sym_c, offs)) { ok = false; break; }
                    int idx = 0;
                    while (idx < nlen + ndist) {
                        in.refill();
                        const int s = inf_decode<10, 3>(in, cc, sym_c);
                        if (s < 0) { ok = false; break; }
                        if (s < 16) len[idx++] = (uint8_t)s;
                        else {
                            int prev = 0, rep;
                            if (s == 16) { if (idx == 0) { ok = false; break; } prev = len[(idx - 1)]; rep = 3 + (int)in.take(2); }
                            else if (s == 17) rep = 3 + (int)in.take(3);
                            else rep = 11 + (int)in.take(7);
                            if (idx + rep > nlen + ndist) { ok = false; break; }
                            while (rep--) len[idx++] = (uint8_t)prev;
                        }
                    }
                    if (!ok) break;
                    if (len[256] == 0) { ok = false; break; }                 // no end-of-block code
                    // distance lengths lie behind the literal / length ones: move them to their own place before either table is built
                    for (int k = ndist - 1; k >= 0; k--) len[(288 + k)] = len[(nlen + k)];
                    for (int k = nlen; k < 288; k++) len[k] = 0;
                    for (int k = ndist; k < 30; k++) len[(288 + k)] = 0;
                    if (!inf_construct<10, 3>(len, 288, cl, sym_l, offs)) { ok = false; break; }
                    if (!inf_construct<6, 5>(len + 288, 30, cd, sym_d, offs)) { ok = false; break; }
                }
                // ---- the block's symbols
                for (;;) {
                    in.refill();
                    int s = inf_decode<10, 3>(in, cl, sym_l);
                    if (s < 0 || in.over()) { ok = false; break; }
                    if (s < 256) { if (pos >= usize) { ok = false; break; } o[pos++] = (uint8_t)s; continue; }
                    if (s == 256) break;
                    s -= 257;
                    if (s >= 29) { ok = false; break; }
                    const uint32_t mlen = (uint32_t)INF_LBASE[s] + in.take(INF_LEXT[s]);
                    const int ds = inf_decode<6, 5>(in, cd, sym_d);
                    if (ds < 0 || ds >= 30) { ok = false; break; }
                    const uint32_t dist = (uint32_t)INF_DBASE[ds] + in.take(INF_DEXT[ds]);
                    if (dist > pos || pos + mlen > usize) { ok = false; break; }
                    uint8_t *dst = o + pos; const uint8_t *from = dst - dist;
                    uint32_t k = 0;
                    if (dist >= 8) for (; k + 8 <= mlen; k += 8) *(inf_u64u *)(dst + k) = *(const inf_u64u *)(from + k);      // (the words do not overlap)
                    for (; k < mlen; k++) dst[k] = from[k];
                    pos += mlen;
                }
            }
            ok = ok && !in.over() && pos == usize;
#ifndef INF_NO_CRC
            if (ok) {                                                                 // CRC-32 over what was written
                uint32_t crc = 0xFFFFFFFFu, k = 0;
                for (; k + 8 <= usize; k += 8) crc = inf_crc_word(s_crc, crc, *(const inf_u64u *)(o + k));
                for (; k < usize; k++) crc = s_crc[0][(crc ^ o[k]) & 0xFF] ^ (crc >> 8);
                crc = ~crc;
                const uint8_t *t = src + d.csize - 8;
                const uint32_t want = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
                ok = crc == want;
            }
#endif
        }
    }
    if (!ok) { atomicOr(&err[0], 1u); atomicMin(&err[1], m); }
}
