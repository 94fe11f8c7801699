// gce_deflate.hpp — the output record stream compressed into BGZF blocks ON THE GPU (SURVEY.md 8(f)1 "multi-threaded or GPU-assisted"; replaces
// bgzf_write's deflate under sam_write1, src/gencore.cpp:104 via htslib): the mirror of gce_inflate.hpp.  The stream in HBM is cut into blocks of
// <= 65 280 input bytes (as many as keep every CU busy: a block is an independent deflate stream, RFC 1951 inside the gzip framing of SAM spec
// 4.1), ONE LANE PER BLOCK, 32 blocks per workgroup:
//   * greedy LZ77 with one hash-table probe per position (the last place the next four bytes were seen: 2048 entries of 16 bits per lane, one
//     column of a 128 KB table in LDS -- one workgroup per CU), matches extended eight bytes at a time, up to 258 bytes, distances up to 32 768;
//   * FIXED Huffman codes (BTYPE 01): no code construction, literal / length / distance codes from the bit patterns of RFC 1951 3.2.6 computed with
//     v_bfrev and a count-leading-zeros each -- no tables; the bit stream leaves through a 64-bit buffer, four bytes a store;
//   * a block that fixed codes would expand (incompressible bytes: 9 bits per literal) is written STORED (BTYPE 00) instead;
//   * CRC-32 of the block's input (slicing-by-8, tables in LDS) and ISIZE in the trailer, BSIZE in the BGZF extra field.
// The blocks come out in slots of worst-case size; a prefix sum over their sizes and one gather make the file image the host writes.
// What it is for: the host's deflate is the largest stage of the file path that scales with the host (16 threads: 0.08 s for 230 MB at level 1;
// 4 threads: 0.3 s); this encoder takes a few milliseconds and leaves the host the write() alone.  Ratio: that of a greedy fixed-Huffman encoder
// (the library's host level -1 is the same scheme).
#pragma once

#define DEF_T 32
#define DEF_HBITS 11
// the hash table is sized for gfx950's 160 KB of LDS per CU (one 32-lane workgroup per CU holds 136 KB of it): this library is built for gfx950 only
static_assert((size_t)(1u << DEF_HBITS) * DEF_T * 2 + 8 * 256 * 4 <= 160u * 1024u, "k_bgzf_deflate: hash table + CRC tables must fit gfx950's 160 KB of LDS");
namespace {
typedef uint32_t def_u32u __attribute__((aligned(1)));
typedef uint64_t def_u64u __attribute__((aligned(1)));
typedef uint16_t def_u16u __attribute__((aligned(1)));

__global__ __launch_bounds__(DEF_T) void k_bgzf_deflate(const uint8_t *in, uint64_t total, uint32_t blk, uint32_t n_blocks, uint8_t *slots, uint32_t slot_bytes, uint32_t *sizes) {
    __shared__ uint16_t s_tab[1 << DEF_HBITS][DEF_T];                                 // last position + 1 of a 4-byte hash, one column per lane
    __shared__ uint32_t s_crc[8][256];
    const int lane = threadIdx.x;
    for (int k = lane; k < 256; k += DEF_T) {                                         // CRC-32 (reflected 0xEDB88320), slicing-by-8 tables
        uint32_t c = (uint32_t)k;
        for (int j = 0; j < 8; j++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        s_crc[0][k] = c;
    }
    __syncthreads();
    for (int k = lane; k < 256; k += DEF_T) { uint32_t c = s_crc[0][k]; for (int t = 1; t < 8; t++) { c = s_crc[0][c & 0xFF] ^ (c >> 8); s_crc[t][k] = c; } }
    for (int k = 0; k < (1 << DEF_HBITS); k++) s_tab[k][lane] = 0;
    __syncthreads();
    const uint32_t bi = blockIdx.x * DEF_T + (uint32_t)lane;
    if (bi >= n_blocks) return;
    const uint8_t *src = in + (uint64_t)bi * blk;
    const uint32_t n = (uint32_t)min((uint64_t)blk, total - (uint64_t)bi * blk);
    uint8_t *dst = slots + (uint64_t)bi * slot_bytes, *o = dst + 18;
    uint64_t bb = 0; int bc = 0;
    auto put = [&](uint32_t v, int nb) {                                              // nb <= 13 bits, LSB first
        bb |= (uint64_t)v << bc; bc += nb;
        if (bc >= 32) { *(def_u32u *)o = (uint32_t)bb; o += 4; bb >>= 32; bc -= 32; }
    };
    put(3u, 3);                                                                       // BFINAL = 1, BTYPE = 01 (bits 1, then 1 0)
    uint32_t pos = 0;
    while (pos < n) {
        uint32_t len = 0, dist = 0;
        if (pos + 4 <= n) {
            const uint32_t w4 = *(const def_u32u *)(src + pos);
            const uint32_t h = (w4 * 2654435761u) >> (32 - DEF_HBITS);
            const uint32_t cand = s_tab[h][lane];
            s_tab[h][lane] = (uint16_t)(pos + 1);
            if (cand != 0u) {
                const uint32_t c = cand - 1u;
                if (pos - c <= 32768u && *(const def_u32u *)(src + c) == w4) {
                    const uint32_t lim = min(258u, n - pos);
                    len = 4; dist = pos - c;
                    bool open = true;
                    while (open && len + 8 <= lim) {
                        const uint64_t x = *(const def_u64u *)(src + c + len) ^ *(const def_u64u *)(src + pos + len);
                        if (x) { len += (uint32_t)(__ffsll((long long)x) - 1) >> 3; open = false; } else len += 8;
                    }
                    while (open && len < lim && src[c + len] == src[pos + len]) len++;
                }
            }
        }
        if (len >= 4) {
            // length symbol (RFC 1951 3.2.5): 3..10 -> 257..264; beyond, 4 codes per number of extra bits; 258 -> 285
            const uint32_t x = len - 3u;
            uint32_t sym, eb = 0, ev = 0;
            if (len == 258u) sym = 285;
            else if (x < 8u) sym = 257u + x;
            else { const uint32_t nb = 31u - (uint32_t)__clz((int)x); eb = nb - 2u; sym = 261u + 4u * eb + ((x >> eb) & 3u); ev = x & ((1u << eb) - 1u); }
            if (sym < 280u) put(__brev(sym - 256u) >> 25, 7); else put(__brev(0xC0u + (sym - 280u)) >> 24, 8);
            if (eb) put(ev, (int)eb);
            const uint32_t d = dist - 1u;
            uint32_t dc, deb = 0, dev = 0;
            if (d < 4u) dc = d;
            else { const uint32_t nb = 31u - (uint32_t)__clz((int)d); deb = nb - 1u; dc = 2u * nb + ((d >> deb) & 1u); dev = d & ((1u << deb) - 1u); }
            put(__brev(dc) >> 27, 5);
            if (deb) put(dev, (int)deb);
            pos += len;
        } else {
            const uint32_t lit = src[pos];
            if (lit < 144u) put(__brev(0x30u + lit) >> 24, 8); else put(__brev(0x190u + (lit - 144u)) >> 23, 9);
            pos++;
        }
    }
    put(0u, 7);                                                                       // end of block (symbol 256)
    while (bc > 0) { *o++ = (uint8_t)bb; bb >>= 8; bc -= 8; }
    uint32_t dbytes = (uint32_t)(o - (dst + 18));
    if (dbytes > n + 5u) {                                                            // fixed codes expanded it: one stored block (LEN, NLEN, the bytes)
        uint8_t *q = dst + 18;
        q[0] = 1; q[1] = (uint8_t)n; q[2] = (uint8_t)(n >> 8); q[3] = (uint8_t)~n; q[4] = (uint8_t)(~n >> 8);
        uint32_t k = 0;
        for (; k + 8 <= n; k += 8) *(def_u64u *)(q + 5 + k) = *(const def_u64u *)(src + k);
        for (; k < n; k++) q[5 + k] = src[k];
        dbytes = n + 5u;
    }
    uint32_t crc = 0xFFFFFFFFu, k = 0;
    for (; k + 8 <= n; k += 8) crc = inf_crc_word(s_crc, crc, *(const def_u64u *)(src + k));
    for (; k < n; k++) crc = s_crc[0][(crc ^ src[k]) & 0xFF] ^ (crc >> 8);
    crc = ~crc;
    const uint32_t bsize = 18u + dbytes + 8u;
    const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    for (int j = 0; j < 16; j++) dst[j] = hdr[j];
    dst[16] = (uint8_t)(bsize - 1u); dst[17] = (uint8_t)((bsize - 1u) >> 8);
    uint8_t *t = dst + 18 + dbytes;
    *(def_u32u *)t = crc; *(def_u32u *)(t + 4) = n;
    sizes[bi] = bsize;
}
// block bi of the file image = its slot's first sizes[bi] bytes, at off[bi]: a wave per block
__global__ __launch_bounds__(256) void k_deflate_pack(const uint8_t *slots, uint32_t slot_bytes, const uint32_t *sizes, const uint64_t *off, uint32_t n_blocks, uint8_t *out) {
    const int lane = threadIdx.x & 63;
    for (uint32_t bi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; bi < n_blocks; bi += (gridDim.x * blockDim.x) >> 6) {
        const uint8_t *s = slots + (uint64_t)bi * slot_bytes; uint8_t *d = out + off[bi]; const uint32_t sz = sizes[bi];
        for (uint32_t j = 8u * (uint32_t)lane; j < sz; j += 512u) {
            if (j + 8 <= sz) *(def_u64u *)(d + j) = *(const def_u64u *)(s + j);
            else for (uint32_t q = j; q < sz; q++) d[q] = s[q];
        }
    }
}
}  // namespace
