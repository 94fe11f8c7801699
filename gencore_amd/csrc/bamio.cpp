// bamio.cpp — BGZF/BAM reader and writer + FASTA loader behind the C-ABI (include/gencore_amd.h, "files" section).  BGZF members are
// decoded by the raw-deflate decoder below with zlib as fallback and arbiter, their CRC-32 by carry-less multiplication (zlib where the
// CPU has no PCLMULQDQ); the writer deflates with zlib (levels 0..9) or the fixed-Huffman encoder below (level -1).
//
// Replaces on the host what the reference does through htslib and FastaReader around the hot path:
//   sam_open / sam_hdr_read / sam_read1        src/gencore.cpp:164-205      -> gce_bam_open (+ gce_bam_chunk: records -> gce_batch)
//   sam_hdr_write / sam_write1 / sam_close     src/gencore.cpp:187-190,104  -> gce_bam_write (result rows -> records -> BGZF)
//   FastaReader::readAll / readNext / to4bits  src/fastareader.cpp:57-104,139-152,157-168 -> gce_fasta_load
// htslib itself is a pinned dependency that is absent from /root/reference; the formats are the published ones (SAMv1 section 4:
// BGZF = concatenated gzip members with a "BC" extra field, BAM record layout), restated here.
// Everything that scales with the file is spread over `threads` host threads: inflate per BGZF block, the struct-of-arrays fill
// per record range, record rebuild + deflate per output block.  No GPU code in this file.
#include <zlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../include/gencore_amd.h"
#include "gce_samtext.hpp"

static double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

namespace {

struct Block { uint64_t coff; uint32_t csize, usize; uint64_t uoff; };

// host threads when the caller does not say: the CPUs this process may run on (affinity mask / cgroup quota), at most 64 -- on the
// 256-thread box the measurements were made on, inflate and deflate stop scaling near 64 threads and lose 30 % at 256
int default_threads() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota> <period>" or "max <period>"
        long long q = 0, per = 0;
        if (fscanf(f, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) n = std::min<long long>(n, std::max<long long>(1, (q + per - 1) / per));
        fclose(f);
    }
    return std::min(n, 64);
}

inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | p[1] << 8); }
inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline int32_t rdi32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }

template <class F> void parallel_for(int threads, int64_t n, F f) {          // f(thread, begin, end) over contiguous ranges
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n));
    if (threads == 1) { f(0, (int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back([=] { f(t, n * t / threads, n * (t + 1) / threads); });
    for (auto &x : th) x.join();
}

// ---- CRC-32 (gzip polynomial, reflected) by carry-less multiplication: "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ"
// (Gopal et al., Intel 2009) -- fold four 128-bit lanes over 64 input bytes a step, fold the lanes together, reduce 128 -> 64 -> 32
// bits (Barrett).  zlib 1.2.11's table-driven crc32 runs at 1.0 GB/s per thread, which made the checksum 45 % of a BGZF block's
// inflate time (zlib inflates BAM data at ~0.8 GB/s).  Used only when the CPU has PCLMULQDQ and a self-check against zlib passes;
// tails and short buffers stay with zlib.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("pclmul,sse4.1"))) inline __m128i crc_fold(__m128i acc, __m128i k, __m128i next) {   // acc * x^distance mod P, plus the next 16 bytes
    return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc, k, 0x00), _mm_clmulepi64_si128(acc, k, 0x11)), next);
}
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul_state(const uint8_t *buf, size_t len /* >= 64, multiple of 16 */, uint32_t state) {
    // x^(n) mod P constants of the paper for the bit-reflected gzip polynomial: fold distances 4 x 128 (+-32), 128 (+-32), 64, and P / mu
    const __m128i k_fold4 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll), k_fold1 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k_64 = _mm_set_epi64x(0, 0x0163cd6124ll), k_poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    const __m128i *p = reinterpret_cast<const __m128i *>(buf);
    __m128i a0 = _mm_xor_si128(_mm_loadu_si128(p), _mm_cvtsi32_si128((int)state)), a1 = _mm_loadu_si128(p + 1), a2 = _mm_loadu_si128(p + 2), a3 = _mm_loadu_si128(p + 3);
    p += 4; len -= 64;
    for (; len >= 64; p += 4, len -= 64) {
        a0 = crc_fold(a0, k_fold4, _mm_loadu_si128(p)); a1 = crc_fold(a1, k_fold4, _mm_loadu_si128(p + 1));
        a2 = crc_fold(a2, k_fold4, _mm_loadu_si128(p + 2)); a3 = crc_fold(a3, k_fold4, _mm_loadu_si128(p + 3));
    }
    a0 = crc_fold(a0, k_fold1, a1); a0 = crc_fold(a0, k_fold1, a2); a0 = crc_fold(a0, k_fold1, a3);
    for (; len >= 16; p += 1, len -= 16) a0 = crc_fold(a0, k_fold1, _mm_loadu_si128(p));
    // 128 -> 64 bits
    const __m128i low32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_xor_si128(_mm_srli_si128(a0, 8), _mm_clmulepi64_si128(a0, k_fold1, 0x10));
    t = _mm_xor_si128(_mm_srli_si128(t, 4), _mm_clmulepi64_si128(_mm_and_si128(t, low32), k_64, 0x00));
    // Barrett reduction 64 -> 32 bits
    __m128i q = _mm_clmulepi64_si128(_mm_and_si128(t, low32), k_poly, 0x10);
    q = _mm_clmulepi64_si128(_mm_and_si128(q, low32), k_poly, 0x00);
    return (uint32_t)_mm_extract_epi32(_mm_xor_si128(t, q), 1);
}
bool crc32_clmul_usable() {
    static const bool ok = [] {
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
        uint8_t tmp[1024 + 16];
        uint32_t x = 0x9E3779B9u;
        for (size_t i = 0; i < sizeof tmp; i++) { x = x * 1664525u + 1013904223u; tmp[i] = (uint8_t)(x >> 24); }
        for (size_t off = 0; off < 3; off++)
            for (size_t n : {(size_t)64, (size_t)80, (size_t)128, (size_t)1008, (size_t)1024}) {
                const uint32_t want = (uint32_t)crc32(crc32(0L, Z_NULL, 0), tmp + off, (uInt)n);
                if ((uint32_t)~crc32_clmul_state(tmp + off, n, 0xFFFFFFFFu) != want) return false;
            }
        return true;
    }();
    return ok;
}
#else
bool crc32_clmul_usable() { return false; }
uint32_t crc32_clmul_state(const uint8_t *, size_t, uint32_t s) { return s; }
#endif
// CRC-32 of a whole buffer (what a gzip member stores)
uint32_t crc32_buf(const uint8_t *buf, size_t n) {
    uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
    size_t done = 0;
    if (n >= 64 && crc32_clmul_usable()) { done = n & ~(size_t)15; c = ~crc32_clmul_state(buf, done, 0xFFFFFFFFu); }
    while (done < n) { const size_t m = std::min<size_t>(n - done, 1u << 30); c = (uint32_t)crc32(c, buf + done, (uInt)m); done += m; }
    return c;
}

template <class V> bool read_file(const char *path, V &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize((size_t)std::max(0l, sz));
    const size_t got = sz > 0 ? fread(out.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    return got == (size_t)std::max(0l, sz);
}

// ---- raw-deflate decoder for BGZF members (RFC 1951), used in front of zlib: 64-bit bit buffer refilled eight bytes at a time, one
// table look-up per symbol (10-bit primary table + subtables for literals/lengths, 8-bit + subtables for distances), matches copied
// in 8-byte words.  A BGZF member is self-contained (empty window at its start, <= 64 KB out), every output byte is bounds-checked,
// and the caller verifies the member's CRC-32 -- whatever this decoder does not handle (incomplete Huffman codes, damaged streams)
// or gets wrong falls back to zlib's inflate, which stays the arbiter of what a valid stream is.
namespace fastinf {
enum : uint32_t { K_INVALID = 0, K_LIT = 1, K_LEN = 2, K_EOB = 4, K_SUB = 8, K_DIST = 6 };   // (K_LIT and K_SUB are single bits: tested with one AND)
constexpr int LIT_BITS = 10, DIST_BITS = 8, LIT_CAP = (1 << LIT_BITS) + 1024, DIST_CAP = (1 << DIST_BITS) + 512;
// entry: bits 0..7 code bits to consume | 8..11 kind | 12..15 extra bits (K_SUB: subtable bits) | 16..31 value (literal, base, subtable start)
inline uint32_t mk(uint32_t kind, uint32_t bits, uint32_t extra, uint32_t value) { return bits | kind << 8 | extra << 12 | value << 16; }
const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t sym_entry_litlen(int sym, uint32_t bits) {
    if (sym < 256) return mk(K_LIT, bits, 0, (uint32_t)sym);
    if (sym == 256) return mk(K_EOB, bits, 0, 0);
    if (sym <= 285) return mk(K_LEN, bits, LEN_EXTRA[sym - 257], LEN_BASE[sym - 257]);
    return mk(K_INVALID, bits, 0, 0);
}
inline uint32_t sym_entry_dist(int sym, uint32_t bits) {
    if (sym < 30) return mk(K_DIST, bits, DIST_EXTRA[sym], DIST_BASE[sym]);
    return mk(K_INVALID, bits, 0, 0);
}
inline uint32_t rev_bits(uint32_t code, int len) { uint32_t r = 0; for (int i = 0; i < len; i++) { r = r << 1 | (code & 1); code >>= 1; } return r; }

// canonical Huffman code of `lens` -> look-up table.  Only COMPLETE codes are taken (Kraft sum exactly 1); returns false otherwise.
template <class EntryOf> bool build_table(const uint8_t *lens, int nsym, int primary, uint32_t *table, int cap, EntryOf entry_of) {
    int count[16] = {0};
    for (int s = 0; s < nsym; s++) count[lens[s]]++;
    count[0] = 0;
    uint32_t kraft = 0;
    for (int l = 1; l <= 15; l++) kraft += (uint32_t)count[l] << (15 - l);
    if (kraft != (1u << 15)) return false;
    uint32_t next_code[16]; { uint32_t code = 0; for (int l = 1; l <= 15; l++) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; } }
    // reversed code of every coded symbol; the longest code behind every primary prefix
    uint16_t rcode[288]; uint8_t sub_bits[1 << LIT_BITS];
    const int np = 1 << primary;
    memset(sub_bits, 0, (size_t)np);
    for (int s = 0; s < nsym; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = rev_bits(next_code[l]++, l);
        rcode[s] = (uint16_t)r;
        if (l > primary) { uint8_t &b = sub_bits[r & (uint32_t)(np - 1)]; b = (uint8_t)std::max<int>(b, l - primary); }
    }
    int used = np;
    for (int i = 0; i < np; i++) {
        if (!sub_bits[i]) continue;
        if (used + (1 << sub_bits[i]) > cap) return false;
        table[i] = mk(K_SUB, (uint32_t)primary, sub_bits[i], (uint32_t)used);
        used += 1 << sub_bits[i];
    }
    for (int s = 0; s < nsym; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = rcode[s];
        if (l <= primary) { const uint32_t e = entry_of(s, (uint32_t)l); for (uint32_t i = r; i < (uint32_t)np; i += 1u << l) table[i] = e; }
        else {
            const uint32_t pi = r & (uint32_t)(np - 1), sb = sub_bits[pi], start = table[pi] >> 16, e = entry_of(s, (uint32_t)(l - primary));
            for (uint32_t i = r >> primary; i < (1u << sb); i += 1u << (l - primary)) table[start + i] = e;
        }
    }
    return true;
}

struct Tables { uint32_t lit[LIT_CAP], dist[DIST_CAP]; };
const Tables *fixed_tables() {
    static const Tables *t = [] {
        Tables *x = new Tables;
        uint8_t l[288]; for (int i = 0; i < 288; i++) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        uint8_t d[32]; for (int i = 0; i < 32; i++) d[i] = 5;
        build_table(l, 288, LIT_BITS, x->lit, LIT_CAP, sym_entry_litlen); build_table(d, 32, DIST_BITS, x->dist, DIST_CAP, sym_entry_dist);
        return x;
    }();
    return t;
}

// src[0, n): raw deflate; dst[0, want): exactly `want` bytes must come out.  src must be readable up to src + n + 8 (a BGZF member's
// CRC-32 and ISIZE follow its deflate data).  false = not handled (the caller runs zlib).
bool inflate_raw(const uint8_t *src, size_t n, uint8_t *dst, size_t want) {
    const uint8_t *in = src, *const in_end = src + n;
    uint8_t *out = dst, *const out_end = dst + want;
    uint64_t bb = 0; int nb = 0;                                              // bit buffer, valid bits
    const uint8_t *const lim = in_end + 8;                                    // readable up to here (the member's CRC-32 and ISIZE)
    auto refill = [&]() -> bool {                                             // >= 56 valid bits afterwards; bits past the deflate data are whatever follows
        uint64_t w = 0;                                                       // it (or zeros) -- a stream that needs them fails the position check at the end
        if (in + 8 <= lim) memcpy(&w, in, 8);
        else { if (in > lim) return false; for (int i = 0; in + i < lim; i++) w |= (uint64_t)in[i] << (8 * i); }
        bb |= w << nb; in += (63 - nb) >> 3; nb |= 56;
        return true;
    };
    Tables dyn;
    for (bool last = false; !last;) {
        if (!refill()) return false;
        last = bb & 1; const uint32_t type = (uint32_t)(bb >> 1) & 3; bb >>= 3; nb -= 3;
        if (type == 0) {                                                      // stored
            const int drop = nb & 7; bb >>= drop; nb -= drop;
            if (!refill()) return false;
            const uint32_t len = (uint32_t)bb & 0xFFFF, nlen = (uint32_t)(bb >> 16) & 0xFFFF; bb >>= 32; nb -= 32;
            if ((len ^ nlen) != 0xFFFF) return false;
            const uint8_t *p = in - (nb >> 3);                                // first byte not yet consumed (nb is a multiple of 8 here)
            if ((size_t)(in_end - p) < len || p > in_end || (size_t)(out_end - out) < len) return false;
            memcpy(out, p, len); out += len; in = p + len; bb = 0; nb = 0;
            continue;
        }
        const uint32_t *lit, *dist;
        if (type == 1) { const Tables *f = fixed_tables(); lit = f->lit; dist = f->dist; }
        else if (type == 2) {
            const int hlit = (int)(bb & 31) + 257, hdist = (int)(bb >> 5 & 31) + 1, hclen = (int)(bb >> 10 & 15) + 4; bb >>= 14; nb -= 14;
            if (hlit > 286 || hdist > 30) return false;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < hclen; i++) { if (nb < 3 && !refill()) return false; cl[order[i]] = (uint8_t)(bb & 7); bb >>= 3; nb -= 3; }
            uint32_t clt[1 << 7];
            if (!build_table(cl, 19, 7, clt, 1 << 7, [](int s, uint32_t bits) { return mk(K_LIT, bits, 0, (uint32_t)s); })) return false;
            uint8_t lens[288 + 32]; int k = 0;
            memset(lens, 0, sizeof lens);
            while (k < hlit + hdist) {
                if (!refill()) return false;
                const uint32_t e = clt[bb & 127]; const int s = (int)(e >> 16); bb >>= (e & 0xFF); nb -= (int)(e & 0xFF);
                if (s < 16) { lens[k++] = (uint8_t)s; continue; }
                int rep; uint8_t v = 0;
                if (s == 16) { if (k == 0) return false; v = lens[k - 1]; rep = 3 + (int)(bb & 3); bb >>= 2; nb -= 2; }
                else if (s == 17) { rep = 3 + (int)(bb & 7); bb >>= 3; nb -= 3; }
                else { rep = 11 + (int)(bb & 127); bb >>= 7; nb -= 7; }
                if (k + rep > hlit + hdist) return false;
                memset(lens + k, v, (size_t)rep); k += rep;
            }
            if (lens[256] == 0) return false;                                 // no end-of-block code
            uint8_t dl[32]; memset(dl, 0, sizeof dl); memcpy(dl, lens + hlit, (size_t)hdist);
            memset(lens + hlit, 0, (size_t)(288 - hlit));
            if (!build_table(lens, 288, LIT_BITS, dyn.lit, LIT_CAP, sym_entry_litlen)) return false;
            if (!build_table(dl, 32, DIST_BITS, dyn.dist, DIST_CAP, sym_entry_dist)) return false;      // (a lone distance code: zlib's business)
            lit = dyn.lit; dist = dyn.dist;
        } else return false;
        for (;;) {                                                            // symbols of the block
            if (!refill()) return false;                                      // >= 56 bits: a length/distance pair needs at most 15 + 5 + 15 + 13 = 48
            uint32_t e = lit[bb & ((1u << LIT_BITS) - 1)];
            if (e & (K_LIT << 8)) {                                           // literals first: up to four from one refill (<= 10 + 3 x 10 + ... bits of the 56)
                if (out_end - out < 4) { if (out >= out_end) return false; bb >>= (e & 0xFF); nb -= (int)(e & 0xFF); *out++ = (uint8_t)(e >> 16); continue; }
                bb >>= (e & 0xFF); nb -= (int)(e & 0xFF); *out++ = (uint8_t)(e >> 16);
                e = lit[bb & ((1u << LIT_BITS) - 1)];
                if (!(e & (K_LIT << 8))) continue;
                bb >>= (e & 0xFF); nb -= (int)(e & 0xFF); *out++ = (uint8_t)(e >> 16);
                e = lit[bb & ((1u << LIT_BITS) - 1)];
                if (!(e & (K_LIT << 8))) continue;
                bb >>= (e & 0xFF); nb -= (int)(e & 0xFF); *out++ = (uint8_t)(e >> 16);
                e = lit[bb & ((1u << LIT_BITS) - 1)];
                if (!(e & (K_LIT << 8))) continue;
                bb >>= (e & 0xFF); nb -= (int)(e & 0xFF); *out++ = (uint8_t)(e >> 16);
                continue;
            }
            if (e & (K_SUB << 8)) { bb >>= LIT_BITS; nb -= LIT_BITS; e = lit[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 12) & 15)) - 1))]; }
            bb >>= (e & 0xFF); nb -= (int)(e & 0xFF);
            const uint32_t kind = (e >> 8) & 15;
            if (kind == K_LIT) { if (out >= out_end) return false; *out++ = (uint8_t)(e >> 16); continue; }      // (a literal with a long code)
            if (kind == K_EOB) break;
            if (kind != K_LEN) return false;
            const uint32_t xl = (e >> 12) & 15, length = (e >> 16) + (uint32_t)(bb & ((1u << xl) - 1)); bb >>= xl; nb -= (int)xl;
            uint32_t d = dist[bb & ((1u << DIST_BITS) - 1)];
            if (d & (K_SUB << 8)) { bb >>= DIST_BITS; nb -= DIST_BITS; d = dist[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 12) & 15)) - 1))]; }
            bb >>= (d & 0xFF); nb -= (int)(d & 0xFF);
            if (((d >> 8) & 15) != K_DIST) return false;
            const uint32_t xd = (d >> 12) & 15, distance = (d >> 16) + (uint32_t)(bb & ((1u << xd) - 1)); bb >>= xd; nb -= (int)xd;
            if (nb < 0) return false;                                         // ran past what the refill provided
            if (distance > (size_t)(out - dst) || length > (size_t)(out_end - out)) return false;
            const uint8_t *from = out - distance;
            if (distance >= 8 && (size_t)(out_end - out) >= length + 8) {     // whole words (may run up to 7 bytes over the match, inside the member)
                uint8_t *o = out; const uint8_t *f = from;
                for (uint32_t c = 0; c < length; c += 8) { uint64_t w; memcpy(&w, f + c, 8); memcpy(o + c, &w, 8); }
            } else if (distance == 1) memset(out, from[0], length);           // a run of one byte (quality strings are full of them)
            else for (uint32_t c = 0; c < length; c++) out[c] = from[c];
            out += length;
        }
        if (nb < 0) return false;
    }
    if ((size_t)(in - src) * 8 - (size_t)nb > n * 8) return false;            // consumed bits past the end of the deflate data
    return out == out_end;
}
}  // namespace fastinf

// one raw-deflate BGZF member -> dst (usize bytes); checks the CRC
bool inflate_block(const uint8_t *src, const Block &b, uint8_t *dst) {
    const uint16_t xlen = rd16(src + 10);
    const uint8_t *cdata = src + 12 + xlen;
    const uint32_t clen = b.csize - 12 - xlen - 8;
    static const bool zlib_only = getenv("GCE_BAM_ZLIB_ONLY") != nullptr;     // (A/B and tests)
    if (!zlib_only && fastinf::inflate_raw(cdata, clen, dst, b.usize) && crc32_buf(dst, b.usize) == rd32(src + b.csize - 8)) return true;
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t *>(cdata); zs.avail_in = clen; zs.next_out = dst; zs.avail_out = b.usize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.total_out != b.usize) return false;
    return crc32_buf(dst, b.usize) == rd32(src + b.csize - 8);
}

// ---- "level 1" raw-deflate encoder for BGZF members: greedy LZ77 over a 2^13-entry hash of 4-byte strings (the member is its own
// window: positions fit 16 bits), ONE fixed-Huffman block (RFC 1951 3.2.6) -- no code construction, a 64-bit bit accumulator.  Gives
// up (returns 0) when the result would not fit a BGZF member; the caller then takes zlib as before.
namespace fastdef {
struct Codes {
    uint16_t lit[286]; uint8_t lit_bits[286];              // literal / end-of-block codes, bit-reversed for the LSB-first stream
    uint32_t len[259]; uint8_t len_bits[259];              // match length 3..258: code + extra bits in one word
    uint8_t dcode[512];                                    // distance - 1 -> distance code (zlib's two-level index)
};
inline uint32_t rev(uint32_t code, int len) { uint32_t r = 0; for (int i = 0; i < len; i++) { r = r << 1 | (code & 1); code >>= 1; } return r; }
const Codes &codes() {
    static const Codes c = [] {
        Codes x; memset(&x, 0, sizeof x);
        auto fixed = [](int s, uint32_t &code, int &bits) {
            if (s < 144) { code = 0x30 + (uint32_t)s; bits = 8; } else if (s < 256) { code = 0x190 + (uint32_t)(s - 144); bits = 9; }
            else if (s < 280) { code = (uint32_t)(s - 256); bits = 7; } else { code = 0xC0 + (uint32_t)(s - 280); bits = 8; }
        };
        for (int s = 0; s <= 256; s++) { uint32_t code; int bits; fixed(s, code, bits); x.lit[s] = (uint16_t)rev(code, bits); x.lit_bits[s] = (uint8_t)bits; }
        for (int L = 3; L <= 258; L++) {
            int idx = 28; while (fastinf::LEN_BASE[idx] > L) idx--;
            if (L == 258) idx = 28;
            uint32_t code; int bits; fixed(257 + idx, code, bits);
            x.len[L] = rev(code, bits) | (uint32_t)(L - fastinf::LEN_BASE[idx]) << bits; x.len_bits[L] = (uint8_t)(bits + fastinf::LEN_EXTRA[idx]);
        }
        for (int d = 1; d <= 32768; d++) {
            int dc = 29; while (fastinf::DIST_BASE[dc] > d) dc--;
            const int k = d - 1;
            x.dcode[k < 256 ? k : 256 + (k >> 7)] = (uint8_t)dc;           // (all distances that share an index share a code)
        }
        return x;
    }();
    return c;
}
// src[0, n), n <= 65535 -> dst[0, cap); returns the size or 0
size_t deflate_fixed(const uint8_t *src, uint32_t n, uint8_t *dst, size_t cap) {
    const Codes &c = codes();
    uint16_t head[1 << 13];
    memset(head, 0, sizeof head);
    uint8_t *out = dst, *const out_end = dst + cap;
    uint64_t acc = 0; int nacc = 0;
    auto put = [&](uint64_t v, int bits) -> bool {                            // bits <= 31 per call
        acc |= v << nacc; nacc += bits;
        if (nacc >= 32) { if (out + 4 > out_end) return false; const uint32_t w = (uint32_t)acc; memcpy(out, &w, 4); out += 4; acc >>= 32; nacc -= 32; }
        return true;
    };
    if (!put(1 | 1 << 1, 3)) return 0;                                        // BFINAL = 1, BTYPE = 01
    uint32_t i = 0;
    while (i + 4 <= n) {
        uint32_t cur; memcpy(&cur, src + i, 4);
        const uint32_t h = (cur * 2654435761u) >> 19;
        const uint32_t cand = head[h];
        head[h] = (uint16_t)(i + 1);
        uint32_t at;
        if (cand && (memcpy(&at, src + cand - 1, 4), at == cur) && i - (cand - 1) <= 32768u) {
            const uint8_t *a = src + i, *b = src + cand - 1;
            const uint32_t maxlen = std::min<uint32_t>(258u, n - i);
            uint32_t len = 4;
            while (len + 8 <= maxlen) { uint64_t x, y; memcpy(&x, a + len, 8); memcpy(&y, b + len, 8); if (x != y) { len += (uint32_t)__builtin_ctzll(x ^ y) >> 3; goto done; } len += 8; }
            while (len < maxlen && a[len] == b[len]) len++;
        done:
            const uint32_t d = i - (cand - 1), dc = c.dcode[d - 1 < 256 ? d - 1 : 256 + ((d - 1) >> 7)];
            if (!put(c.len[len], c.len_bits[len])) return 0;
            if (!put(rev(dc, 5) | (uint64_t)(d - fastinf::DIST_BASE[dc]) << 5, 5 + fastinf::DIST_EXTRA[dc])) return 0;
            i += len;
        } else {
            if (!put(c.lit[src[i]], c.lit_bits[src[i]])) return 0;
            i++;
        }
    }
    for (; i < n; i++) if (!put(c.lit[src[i]], c.lit_bits[src[i]])) return 0;
    if (!put(c.lit[256], c.lit_bits[256])) return 0;
    while (nacc > 0) { if (out >= out_end) return 0; *out++ = (uint8_t)acc; acc >>= 8; nacc -= 8; }
    return (size_t)(out - dst);
}
}  // namespace fastdef

// one BGZF member from `n` (<= 0xff00) bytes; returns its size
size_t deflate_block(const uint8_t *src, uint32_t n, int level, uint8_t *dst /* >= 0x10000 + 64 */) {
    static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(dst, head, 16);
    static const bool zlib_only = getenv("GCE_BAM_ZLIB_ONLY") != nullptr;
    if (level < 0 && !zlib_only) {                                            // level -1, "fastest": the fixed-Huffman encoder above
        const size_t clen = fastdef::deflate_fixed(src, n, dst + 18, 0x10000 - 18 - 8);
        if (clen) {
            const size_t total = 18 + clen + 8;
            const uint16_t bsize = (uint16_t)(total - 1);
            memcpy(dst + 16, &bsize, 2);
            const uint32_t crc = crc32_buf(src, n);
            memcpy(dst + 18 + clen, &crc, 4); memcpy(dst + 18 + clen + 4, &n, 4);
            return total;
        }
    }
    if (level < 0) level = 1;
    if (level > 9) level = 9;
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = const_cast<uint8_t *>(src); zs.avail_in = n; zs.next_out = dst + 18; zs.avail_out = 0x10000 - 18 - 8;
    int rc = deflate(&zs, Z_FINISH);
    if (rc != Z_STREAM_END) {                                                 // incompressible: store
        deflateEnd(&zs); memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, 0, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = const_cast<uint8_t *>(src); zs.avail_in = n; zs.next_out = dst + 18; zs.avail_out = 0x10000 - 18 - 8;
        rc = deflate(&zs, Z_FINISH);
    }
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return 0;                                         // (cannot happen for <= 0xff00 bytes stored; the caller reports it)
    const size_t total = 18 + clen + 8;
    const uint16_t bsize = (uint16_t)(total - 1);
    memcpy(dst + 16, &bsize, 2);
    const uint32_t crc = crc32_buf(src, n);
    memcpy(dst + 18 + clen, &crc, 4); memcpy(dst + 18 + clen + 4, &n, 4);
    return total;
}

// a growable buffer that is NOT value-initialised (std::vector::resize would write gigabytes of zeros on one thread)
template <class T> struct Raw {
    T *p = nullptr; size_t cap = 0, n = 0;
    Raw() = default;
    Raw(const Raw &) = delete; Raw &operator=(const Raw &) = delete;
    Raw(Raw &&o) noexcept : p(o.p), cap(o.cap), n(o.n) { o.p = nullptr; o.cap = o.n = 0; }
    Raw &operator=(Raw &&o) noexcept { if (this != &o) { free(p); p = o.p; cap = o.cap; n = o.n; o.p = nullptr; o.cap = o.n = 0; } return *this; }
    ~Raw() { free(p); }
    void release() { free(p); p = nullptr; cap = n = 0; }
    void resize(size_t k) {
        if (k > cap) {
            free(p);
            const size_t bytes = std::max<size_t>(k, 1) * sizeof(T);
            if (bytes >= (8u << 20)) {                                           // big buffers: 2 MB pages where the kernel offers them (first touch
                const size_t rounded = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);    // of 4 KB pages is what the host path mostly waits for)
                p = (T *)aligned_alloc(2u << 20, rounded);
                if (p) madvise(p, rounded, MADV_HUGEPAGE);
            } else p = (T *)malloc(bytes);
            cap = p ? k : 0;
        }
        n = p ? k : 0;
    }
    bool ok() const { return p != nullptr; }                                     // false after a failed allocation (callers return GCE_ERR_OOM)
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

// the whole file into an uninitialised buffer, every thread pulling its own range (one thread copies out of the page cache at ~10 GB/s:
// 60 ms for the 572 MB of the benchmark's file)
bool read_file_parallel(const char *path, Raw<uint8_t> &out, int threads) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 0) { close(fd); return false; }
    const int64_t sz = (int64_t)st.st_size;
    out.resize((size_t)sz);
    if (!out.ok()) { close(fd); return false; }
    std::atomic<int> bad{0};
    const int64_t piece = 8 << 20;
    parallel_for(threads, (sz + piece - 1) / piece, [&](int, int64_t a, int64_t e) {
        for (int64_t k = a; k < e; k++) {
            int64_t off = k * piece; const int64_t end = std::min(sz, off + piece);
            while (off < end) { const ssize_t got = pread(fd, out.data() + off, (size_t)(end - off), (off_t)off); if (got <= 0) { bad = 1; return; } off += got; }
        }
    });
    close(fd);
    return !bad;
}
struct Slot {                                     // struct-of-arrays buffers of one chunk
    Raw<gce_core> core; Raw<uint64_t> qoff, coff, soff, loff, mioff;
    Raw<char> qname, mi; Raw<uint32_t> cigar; Raw<uint8_t> seq, qual, nmt; Raw<int32_t> nm;
};

// aux walk of one record: NM (type + value as bam_aux2i gives it) and MI:Z
struct AuxInfo { uint8_t nm_type; int32_t nm; const char *mi; };
inline size_t aux_size(uint8_t type, const uint8_t *p, const uint8_t *end) {   // bytes of the value behind the type byte, or SIZE_MAX
    switch (type) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    case 'Z': case 'H': { const uint8_t *q = p; while (q < end && *q) q++; return q < end ? (size_t)(q - p) + 1 : SIZE_MAX; }
    case 'B': {
        if (p + 5 > end) return SIZE_MAX;
        const size_t es = aux_size(p[0], nullptr, nullptr);
        return es == SIZE_MAX ? SIZE_MAX : 5 + es * (size_t)rd32(p + 1);
    }
    default: return SIZE_MAX;
    }
}
inline AuxInfo scan_aux(const uint8_t *p, const uint8_t *end) {
    AuxInfo a{0, 0, nullptr};
    while (p + 3 <= end) {
        const uint8_t t0 = p[0], t1 = p[1], type = p[2];
        const uint8_t *v = p + 3;
        const size_t sz = aux_size(type, v, end);
        if (sz == SIZE_MAX || v + sz > end) break;
        if (t0 == 'N' && t1 == 'M' && a.nm_type == 0) {                      // bam_aux_get returns the first match
            a.nm_type = type;
            switch (type) {                                                   // bam_aux2i
            case 'c': a.nm = (int8_t)v[0]; break;   case 'C': a.nm = v[0]; break;
            case 's': a.nm = (int16_t)rd16(v); break; case 'S': a.nm = rd16(v); break;
            case 'i': a.nm = rdi32(v); break;        case 'I': a.nm = (int32_t)rd32(v); break;
            default: a.nm = 0; break;
            }
        } else if (t0 == 'M' && t1 == 'I' && type == 'Z' && !a.mi) a.mi = (const char *)v;
        p = v + sz;
    }
    return a;
}

// body -> BGZF blocks of 0xff00 bytes + the EOF marker block
int write_bgzf(const char *path, const Raw<uint8_t> &body, int T, int level) {
    const uint64_t BS = 0xff00;
    const int64_t nb = (int64_t)((body.size() + BS - 1) / BS);
    Raw<uint8_t> z; z.resize((size_t)nb * 0x10000 + 64);                      // (not a std::vector: its zero fill of these 400 MB, and of the body's, on one
                                                                              //  thread was most of the write stage)
    if (!z.ok()) return GCE_ERR_OOM;
    std::vector<uint32_t> zs((size_t)nb, 0);
    parallel_for(T, nb, [&](int, int64_t a, int64_t e) {
        for (int64_t k = a; k < e; k++) {
            const uint64_t o = (uint64_t)k * BS; const uint32_t len = (uint32_t)std::min<uint64_t>(BS, body.size() - o);
            zs[k] = (uint32_t)deflate_block(body.data() + o, len, level, z.data() + (size_t)k * 0x10000);
        }
    });
    for (int64_t k = 0; k < nb; k++) if (zs[k] == 0) return GCE_ERR_INVALID;       // a block that could not be deflated
    FILE *f = fopen(path, "wb");
    if (!f) return GCE_ERR_INVALID;
    for (int64_t k = 0; k < nb; k++) if (fwrite(z.data() + (size_t)k * 0x10000, 1, zs[k], f) != zs[k]) { fclose(f); return GCE_ERR_INVALID; }
    static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool eof_ok = fwrite(eof_block, 1, 28, f) == 28;                         // (a full disk must not pass for a finished BAM: the reference exits when sam_write1 / sam_close fail)
    if (fclose(f) != 0 || !eof_ok) return GCE_ERR_INVALID;
    return GCE_OK;
}

}  // namespace

struct gce_bam {
    Raw<uint8_t> u;                               // the inflated stream
    std::string text;
    std::vector<std::string> names; std::vector<const char *> name_ptr; std::vector<uint32_t> lens;
    std::vector<uint64_t> rec;                    // offset of every record's block_size
    uint64_t tot_q = 0, tot_c = 0, tot_s = 0, tot_l = 0, tot_mi = 0;
    int threads = 1;
    Slot slot[2];
    std::string err;
    double t_read = 0, t_inflate = 0, t_index = 0;
};


extern "C" {

int gce_bam_open(const char *path, int threads, gce_bam **out) {
    if (!path || !out) return GCE_ERR_INVALID;
    gce_bam *f = new gce_bam();
    f->threads = threads > 0 ? threads : default_threads();
    *out = f;
    double t0 = now_s();
    Raw<uint8_t> z;
    if (!read_file_parallel(path, z, f->threads)) { f->err = std::string("cannot read ") + path; return GCE_ERR_INVALID; }
    f->t_read = now_s() - t0; t0 = now_s();
    // ---- BGZF members
    std::vector<Block> blocks;
    uint64_t off = 0, uoff = 0;
    while (off + 18 <= z.size()) {
        const uint8_t *p = z.data() + off;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) { f->err = "not a BGZF file"; return GCE_ERR_INVALID; }
        const uint16_t xlen = rd16(p + 10);
        if (off + 12 + (uint64_t)xlen + 8 > z.size()) { f->err = "truncated BGZF block header"; return GCE_ERR_INVALID; }   // the extra field (and the trailer) must lie inside the file
        uint32_t bsize = 0; bool found = false;
        for (uint32_t x = 0; x + 4 <= xlen; ) {
            const uint8_t *s = p + 12 + x; const uint16_t sl = rd16(s + 2);
            if (x + 4 + (uint32_t)sl > xlen) break;                                 // a subfield that runs past the extra field
            if (s[0] == 'B' && s[1] == 'C' && sl == 2) { bsize = (uint32_t)rd16(s + 4) + 1; found = true; }
            x += 4 + sl;
        }
        if (!found || off + bsize > z.size() || bsize < 12u + xlen + 8u) { f->err = "bad BGZF block"; return GCE_ERR_INVALID; }
        Block b; b.coff = off; b.csize = bsize; b.usize = rd32(p + bsize - 4); b.uoff = uoff;
        if (b.usize > 0x10000u) { f->err = "bad BGZF block (ISIZE above 64 KB)"; return GCE_ERR_INVALID; }   // the format's limit: the sum of these sizes the buffer below
        blocks.push_back(b);
        off += bsize; uoff += b.usize;
    }
    if (off != z.size()) { f->err = "trailing bytes after the last BGZF block"; return GCE_ERR_INVALID; }
    f->u.resize(uoff + 64);
    if (!f->u.ok()) { f->err = "out of host memory for the inflated stream"; return GCE_ERR_OOM; }
    if (getenv("GCE_BAM_PRETOUCH")) {                                              // diagnostic: first-touch cost of the destination alone
        const double tp = now_s();
        parallel_for(f->threads, (int64_t)((uoff + 4095) / 4096), [&](int, int64_t a, int64_t e) { for (int64_t k = a; k < e; k++) f->u.data()[(size_t)k * 4096] = 0; });
        fprintf(stderr, "pretouch %.3f s\n", now_s() - tp);
    }
    std::atomic<int> bad{0};
    parallel_for(f->threads, (int64_t)blocks.size(), [&](int, int64_t a, int64_t e) {
        for (int64_t k = a; k < e; k++) if (blocks[k].usize && !inflate_block(z.data() + blocks[k].coff, blocks[k], f->u.data() + blocks[k].uoff)) bad = 1;
    });
    if (bad) { f->err = "inflate / CRC failure"; return GCE_ERR_INVALID; }
    f->t_inflate = now_s() - t0; t0 = now_s();
    z.release();
    // ---- header (SAMv1 4.2)
    const uint8_t *u = f->u.data(); const uint64_t n = uoff;
    if (n < 12 || memcmp(u, "BAM\1", 4) != 0) { f->err = "not a BAM stream"; return GCE_ERR_INVALID; }
    uint64_t p = 4;
    const uint32_t l_text = rd32(u + p); p += 4;
    if (p + l_text + 4 > n) { f->err = "truncated header"; return GCE_ERR_INVALID; }
    f->text.assign((const char *)u + p, l_text); p += l_text;
    const uint32_t n_ref = rd32(u + p); p += 4;
    for (uint32_t r = 0; r < n_ref; r++) {
        if (p + 4 > n) { f->err = "truncated header"; return GCE_ERR_INVALID; }
        const uint32_t ln = rd32(u + p); p += 4;
        if (p + ln + 4 > n || ln == 0) { f->err = "truncated header"; return GCE_ERR_INVALID; }
        f->names.emplace_back((const char *)u + p, ln - 1); p += ln;
        f->lens.push_back(rd32(u + p)); p += 4;
    }
    for (auto &s : f->names) f->name_ptr.push_back(s.c_str());
    // ---- record index.  The records form a chain (every block_size leads to the next record): one dependent cache miss per record
    //      when walked by one thread.  The stream is cut into segments instead; every segment is walked from a GUESSED record start
    //      (a position where two records in a row look sane), and the pieces are then joined: segment s is accepted from the exact
    //      offset at which the verified chain of segment s - 1 leaves that segment.  A guess that never meets the true chain costs a
    //      sequential re-walk of its segment, never a wrong index.
    {
        const uint64_t first = p;
        const int32_t nref = (int32_t)n_ref;
        auto plausible = [&](uint64_t o) -> bool {                                   // does a record start at o?
            if (o + 36 > n) return false;
            const uint32_t bs = rd32(u + o);
            if (bs < 32 || bs > (1u << 24) || o + 4 + bs > n) return false;
            const uint8_t *r = u + o + 4;
            const int32_t tid = rdi32(r), mtid = rdi32(r + 20); const uint32_t lq = r[8], nc = rd16(r + 12); const int32_t ls = rdi32(r + 16);
            if (tid < -1 || tid >= nref || mtid < -1 || mtid >= nref || lq == 0 || ls < 0) return false;
            if (32ull + lq + 4ull * nc + (uint64_t)(ls + 1) / 2 + (uint64_t)ls > bs) return false;
            return r[32 + lq - 1] == 0;
        };
        const int S = (int)std::max<int64_t>(1, std::min<int64_t>(f->threads, (int64_t)((n - first) >> 22)));       // >= 4 MB per segment
        std::vector<std::vector<uint64_t>> part(S);
        std::vector<uint64_t> seg_lo(S + 1);
        for (int sg = 0; sg <= S; sg++) seg_lo[sg] = first + (n - first) * (uint64_t)sg / (uint64_t)S;
        auto walk = [&](uint64_t o, uint64_t hi, std::vector<uint64_t> &out) -> uint64_t {    // records starting in [o, hi); returns where the chain leaves
            while (o < hi && o + 4 <= n) {
                const uint32_t bs = rd32(u + o);
                if (bs < 32 || o + 4 + bs > n) return UINT64_MAX;                        // broken chain
                out.push_back(o);
                o += 4 + bs;
            }
            return o;
        };
        std::vector<uint64_t> leave(S, 0);
        parallel_for(S, S, [&](int, int64_t a, int64_t e) {
            for (int64_t sg = a; sg < e; sg++) {
                uint64_t o = seg_lo[sg];
                if (sg > 0) {                                                            // guess: first position where two records in a row look sane
                    const uint64_t lim = std::min<uint64_t>(seg_lo[sg + 1], o + (1u << 20));
                    while (o < lim && !(plausible(o) && (o + 4 + rd32(u + o) >= n - 3 || plausible(o + 4 + rd32(u + o))))) o++;
                    if (o >= lim) { leave[sg] = UINT64_MAX; continue; }
                }
                part[sg].reserve((size_t)((seg_lo[sg + 1] - seg_lo[sg]) / 200));
                leave[sg] = walk(o, seg_lo[sg + 1], part[sg]);
            }
        });
        const double t_walk = now_s();
        int rewalks = 0;
        uint64_t at = first;                                                             // where the verified chain enters the next segment
        for (int sg = 0; sg < S; sg++) {
            std::vector<uint64_t> &v = part[sg];
            size_t from = 0; bool ok = leave[sg] != UINT64_MAX;
            if (ok && at < seg_lo[sg + 1]) {                                             // (a record longer than a segment skips it entirely)
                const auto it = std::lower_bound(v.begin(), v.end(), at);
                ok = it != v.end() && *it == at; from = (size_t)(it - v.begin());
            } else if (ok) from = v.size();
            if (!ok) {                                                                   // the guess never met the chain: walk the segment for real
                rewalks++;
                v.clear(); from = 0;
                leave[sg] = at < seg_lo[sg + 1] ? walk(at, seg_lo[sg + 1], v) : at;
                if (leave[sg] == UINT64_MAX) { f->err = "truncated record"; return GCE_ERR_INVALID; }
            }
            f->rec.insert(f->rec.end(), v.begin() + from, v.end());
            if (at < seg_lo[sg + 1]) at = leave[sg];
            std::vector<uint64_t>().swap(v);
        }
        p = at;
        if (getenv("GCE_BAM_VERBOSE")) fprintf(stderr, "index: %d segments, walk %.3f s, join %.3f s, %d re-walked\n", S, t_walk - t0, now_s() - t_walk, rewalks);
    }
    if (p != n) { f->err = "trailing bytes after the last record"; return GCE_ERR_INVALID; }
    const int64_t nr = (int64_t)f->rec.size();
    std::vector<uint64_t> tq(f->threads + 1, 0), tc(f->threads + 1, 0), ts(f->threads + 1, 0), tl(f->threads + 1, 0), tm(f->threads + 1, 0);
    std::atomic<int> badrec{0};
    parallel_for(f->threads, nr, [&](int t, int64_t a, int64_t e) {
        uint64_t q = 0, c = 0, sq = 0, l = 0, m = 0;                              // (thread-local: the shared arrays would bounce between cores)
        for (int64_t k = a; k < e; k++) {
            const uint8_t *r = u + f->rec[k] + 4; const uint32_t bs = rd32(u + f->rec[k]);
            const uint32_t lq = r[8], nc = rd16(r + 12); const int32_t ls = rdi32(r + 16);
            if (lq == 0 || ls < 0 || 32ull + lq + 4ull * nc + (uint64_t)(ls + 1) / 2 + (uint64_t)ls > bs) { badrec = 1; continue; }
            q += lq; c += nc; sq += (uint64_t)(ls + 1) / 2; l += (uint64_t)ls;
            const AuxInfo ai = scan_aux(r + 32 + lq + 4 * nc + (ls + 1) / 2 + ls, r + bs);
            if (ai.mi) m += strlen(ai.mi) + 1;
        }
        tq[t] = q; tc[t] = c; ts[t] = sq; tl[t] = l; tm[t] = m;
    });
    if (badrec) { f->err = "inconsistent record lengths"; return GCE_ERR_INVALID; }
    for (int t = 0; t < f->threads; t++) { f->tot_q += tq[t]; f->tot_c += tc[t]; f->tot_s += ts[t]; f->tot_l += tl[t]; f->tot_mi += tm[t]; }
    f->t_index = now_s() - t0;
    return GCE_OK;
}

void gce_bam_close(gce_bam *f) { delete f; }
const char *gce_bam_error(const gce_bam *f) { return f ? f->err.c_str() : "null"; }

int gce_bam_get_info(const gce_bam *f, gce_bam_info *o) {
    if (!f || !o) return GCE_ERR_INVALID;
    o->n_targets = (int32_t)f->lens.size(); o->target_len = f->lens.data(); o->target_name = f->name_ptr.data();
    o->text = f->text.data(); o->l_text = (int64_t)f->text.size();
    o->n_records = (int64_t)f->rec.size();
    o->qname_bytes = f->tot_q; o->cigar_words = f->tot_c; o->seq_bytes = f->tot_s; o->qual_bytes = f->tot_l; o->mi_bytes = f->tot_mi;
    o->read_s = f->t_read; o->inflate_s = f->t_inflate; o->index_s = f->t_index;
    return GCE_OK;
}

// records [first, first + count) as a gce_batch in one of the two chunk slots (offsets relative to the slot's blobs)
// records first .. first + count - 1, or (sel != nullptr) the records sel[0 .. count - 1] in that order
static int bam_chunk_impl(gce_bam *f, int64_t first, const int64_t *sel, int64_t count, Slot &s, gce_batch *out) {
    const uint8_t *u = f->u.data();
    const int T = f->threads;
    std::vector<uint64_t> tq(T + 1, 0), tc(T + 1, 0), ts(T + 1, 0), tl(T + 1, 0), tm(T + 1, 0);
    parallel_for(T, count, [&](int t, int64_t a, int64_t e) {
        uint64_t q = 0, c = 0, sq = 0, l = 0, m = 0;
        for (int64_t k = a; k < e; k++) {
            const uint8_t *r = u + f->rec[sel ? sel[k] : first + k] + 4; const uint32_t bs = rd32(u + f->rec[sel ? sel[k] : first + k]);
            const uint32_t lq = r[8], nc = rd16(r + 12); const int32_t ls = rdi32(r + 16);
            q += lq; c += nc; sq += (uint64_t)(ls + 1) / 2; l += (uint64_t)ls;
            if (f->tot_mi) { const AuxInfo ai = scan_aux(r + 32 + lq + 4 * nc + (ls + 1) / 2 + ls, r + bs); if (ai.mi) m += strlen(ai.mi) + 1; }
        }
        tq[t + 1] = q; tc[t + 1] = c; ts[t + 1] = sq; tl[t + 1] = l; tm[t + 1] = m;
    });
    for (int t = 0; t < T; t++) { tq[t + 1] += tq[t]; tc[t + 1] += tc[t]; ts[t + 1] += ts[t]; tl[t + 1] += tl[t]; tm[t + 1] += tm[t]; }
    const bool mi = f->tot_mi != 0;
    s.core.resize(count); s.qoff.resize(count); s.coff.resize(count); s.soff.resize(count); s.loff.resize(count); s.nm.resize(count); s.nmt.resize(count);
    s.qname.resize(tq[T] + 64); s.cigar.resize(tc[T] + 16); s.seq.resize(ts[T] + 64); s.qual.resize(tl[T] + 64);
    if (mi) { s.mioff.resize(count); s.mi.resize(tm[T] + 64); }
    if (!s.core.ok() || !s.qoff.ok() || !s.coff.ok() || !s.soff.ok() || !s.loff.ok() || !s.nm.ok() || !s.nmt.ok() || !s.qname.ok() || !s.cigar.ok() || !s.seq.ok() || !s.qual.ok() ||
        (mi && (!s.mioff.ok() || !s.mi.ok()))) { f->err = "out of host memory for a batch"; return GCE_ERR_OOM; }
    const int nthreads_used = (int)std::max<int64_t>(1, std::min<int64_t>(T, count));
    parallel_for(T, count, [&](int t, int64_t a, int64_t e) {
        // parallel_for hands thread t the same range as in the counting pass, so the prefix sums are this range's start offsets
        uint64_t q = tq[t], c = tc[t], sq = ts[t], l = tl[t], m = tm[t];
        for (int64_t k = a; k < e; k++) {
            const uint8_t *r = u + f->rec[sel ? sel[k] : first + k] + 4; const uint32_t bs = rd32(u + f->rec[sel ? sel[k] : first + k]);
            memcpy(&s.core[k], r, 32);                                        // gce_core IS the 32-byte BAM core block
            const uint32_t lq = r[8], nc = rd16(r + 12); const int32_t ls = rdi32(r + 16);
            const uint8_t *pq = r + 32, *pc = pq + lq, *ps = pc + 4 * nc, *pl = ps + (ls + 1) / 2, *pa = pl + ls;
            s.qoff[k] = q; memcpy(s.qname.data() + q, pq, lq); q += lq;
            s.coff[k] = c; memcpy(s.cigar.data() + c, pc, 4 * nc); c += nc;
            s.soff[k] = sq; memcpy(s.seq.data() + sq, ps, (ls + 1) / 2); sq += (uint64_t)(ls + 1) / 2;
            s.loff[k] = l; memcpy(s.qual.data() + l, pl, ls); l += (uint64_t)ls;
            const AuxInfo ai = scan_aux(pa, r + bs);
            s.nmt[k] = ai.nm_type; s.nm[k] = ai.nm;
            if (mi) {
                if (ai.mi) { const size_t n = strlen(ai.mi) + 1; s.mioff[k] = m; memcpy(s.mi.data() + m, ai.mi, n); m += n; }
                else s.mioff[k] = UINT64_MAX;
            }
        }
    });
    (void)nthreads_used;
    memset(out, 0, sizeof *out);
    out->n_reads = count; out->core = s.core.data();
    out->qname_off = s.qoff.data(); out->qname = s.qname.data(); out->cigar_off = s.coff.data(); out->cigar = s.cigar.data();
    out->seq_off = s.soff.data(); out->seq = s.seq.data(); out->qual_off = s.loff.data(); out->qual = s.qual.data();
    out->nm = s.nm.data(); out->nm_type = s.nmt.data();
    if (mi) { out->mi_off = s.mioff.data(); out->mi = s.mi.data(); out->mi_bytes = tm[T]; }
    out->qname_bytes = tq[T]; out->cigar_words = tc[T]; out->seq_bytes = ts[T]; out->qual_bytes = tl[T];
    return GCE_OK;
}

// Gencore::writeBam for every row of the result (src/gencore.cpp:85-111): the input record res->src[k] with the row's bases,
// qualities, name (BamUtil::copyQName, src/bamutil.cpp:338-364), NM byte (src/group.cpp:570) and FR / RR aux (src/pair.cpp:57-67).

int gce_bam_chunk(gce_bam *f, int64_t first, int64_t count, int slot_id, gce_batch *out) {
    if (!f || !out || first < 0 || count < 0 || first + count > (int64_t)f->rec.size() || (slot_id != 0 && slot_id != 1)) return GCE_ERR_INVALID;
    return bam_chunk_impl(f, first, nullptr, count, f->slot[slot_id], out);
}

// the rows of an output table as BAM records: row k = input record src[k] with the name of record qname_src[k], the bases / qualities
// at seqp(k) / qualp(k), NM patched, FR / RR appended (gce_result's meaning; one table, or several engines' tables merged)
struct OutRows { int64_t n; const uint32_t *src, *qname_src; const int32_t *nm_new; const int16_t *fr, *rr; };
extern "C++" {
template <class SP, class QP>
static int bam_write_rows(const char *path, const gce_bam *in, const OutRows rows, SP seqp, QP qualp, int threads, int level) {
    const OutRows *res = &rows;
    const int T = threads > 0 ? threads : in->threads;
    const uint8_t *u = in->u.data();
    const int64_t n = rows.n;
    for (int64_t k = 0; k < n; k++) if (res->src[k] >= in->rec.size() || res->qname_src[k] >= in->rec.size()) return GCE_ERR_INVALID;
    // ---- header bytes
    std::vector<uint8_t> hdr;
    auto put32 = [&](std::vector<uint8_t> &v, uint32_t x) { const uint8_t *p = (const uint8_t *)&x; v.insert(v.end(), p, p + 4); };
    hdr.insert(hdr.end(), {'B', 'A', 'M', 1});
    put32(hdr, (uint32_t)in->text.size()); hdr.insert(hdr.end(), in->text.begin(), in->text.end());
    put32(hdr, (uint32_t)in->lens.size());
    for (size_t r = 0; r < in->lens.size(); r++) {
        put32(hdr, (uint32_t)in->names[r].size() + 1);
        hdr.insert(hdr.end(), in->names[r].begin(), in->names[r].end()); hdr.push_back(0);
        put32(hdr, in->lens[r]);
    }
    // ---- record sizes, then the records
    std::vector<uint64_t> roff((size_t)n + 1, 0);
    parallel_for(T, n, [&](int, int64_t a, int64_t e) {
        for (int64_t k = a; k < e; k++) {
            const uint64_t ro = in->rec[res->src[k]]; const uint32_t bs = rd32(u + ro);
            const uint32_t lq_old = u[ro + 4 + 8], lq_new = u[in->rec[res->qname_src[k]] + 4 + 8];
            roff[k + 1] = 4ull + bs - lq_old + lq_new + (res->fr[k] >= 0 ? 4 : 0) + (res->rr[k] >= 0 ? 4 : 0);
        }
    });
    for (int64_t k = 0; k < n; k++) roff[k + 1] += roff[k];
    Raw<uint8_t> body; body.resize(hdr.size() + roff[n]);
    if (!body.ok()) return GCE_ERR_OOM;
    memcpy(body.data(), hdr.data(), hdr.size());
    uint8_t *rb = body.data() + hdr.size();
    parallel_for(T, n, [&](int, int64_t a, int64_t e) {
        for (int64_t k = a; k < e; k++) {
            const uint8_t *r = u + in->rec[res->src[k]] + 4; const uint32_t bs = rd32(r - 4);
            const uint8_t *nr = u + in->rec[res->qname_src[k]] + 4;
            const uint32_t lq_old = r[8], lq_new = nr[8], nc = rd16(r + 12); const int32_t ls = rdi32(r + 16);
            uint8_t *o = rb + roff[k];
            const uint32_t nbs = (uint32_t)(roff[k + 1] - roff[k] - 4);
            memcpy(o, &nbs, 4); memcpy(o + 4, r, 32);
            o[4 + 8] = (uint8_t)lq_new;
            uint8_t *w = o + 36;
            memcpy(w, nr + 32, lq_new); w += lq_new;
            memcpy(w, r + 32 + lq_old, 4 * nc); w += 4 * nc;
            memcpy(w, seqp(k), (ls + 1) / 2); w += (ls + 1) / 2;
            memcpy(w, qualp(k), ls); w += ls;
            const uint8_t *aux = r + 32 + lq_old + 4 * nc + (ls + 1) / 2 + ls; const size_t al = (size_t)(r + bs - aux);
            memcpy(w, aux, al);
            if (res->nm_new[k] >= 0) {                                       // dataNM[1] = newValNM (type 'C' only, checked by the engine)
                uint8_t *p = w, *end = w + al;
                while (p + 3 <= end) {
                    const size_t sz = aux_size(p[2], p + 3, end);
                    if (sz == SIZE_MAX) break;
                    if (p[0] == 'N' && p[1] == 'M') { p[3] = (uint8_t)res->nm_new[k]; break; }
                    p += 3 + sz;
                }
            }
            w += al;
            if (res->fr[k] >= 0) { w[0] = 'F'; w[1] = 'R'; w[2] = 'C'; w[3] = (uint8_t)res->fr[k]; w += 4; }
            if (res->rr[k] >= 0) { w[0] = 'R'; w[1] = 'R'; w[2] = 'C'; w[3] = (uint8_t)res->rr[k]; w += 4; }
        }
    });
    return write_bgzf(path, body, T, level);
}
}  // extern "C++"

int gce_bam_write(const char *path, const gce_bam *in, const gce_result *res, int threads, int level) {
    if (!path || !in || !res) return GCE_ERR_INVALID;
    const OutRows rows{res->n_out, res->src, res->qname_src, res->nm_new, res->fr, res->rr};
    return bam_write_rows(path, in, rows, [&](int64_t k) { return res->seq + res->seq_off[k]; }, [&](int64_t k) { return res->qual + res->qual_off[k]; }, threads, level);
}

// The inverse of gce_bam_chunk: a gce_batch (host pointers) as a BAM file -- header, one record per read with its NM tag (type and
// value as given) and MI:Z tag.  Used to materialise synthetic streams as files (tools/bam_bench.py, tests).
int gce_bam_from_batch(const char *path, const gce_batch *b, int32_t n_targets, const uint32_t *target_len, const char *const *target_name,
                       const char *text, int threads, int level) {
    if (!path || !b || n_targets < 0) return GCE_ERR_INVALID;
    const int T = threads > 0 ? threads : default_threads();
    std::vector<uint8_t> hdr;
    auto put32 = [&](std::vector<uint8_t> &v, uint32_t x) { const uint8_t *p = (const uint8_t *)&x; v.insert(v.end(), p, p + 4); };
    const std::string tx = text ? text : "";
    hdr.insert(hdr.end(), {'B', 'A', 'M', 1});
    put32(hdr, (uint32_t)tx.size()); hdr.insert(hdr.end(), tx.begin(), tx.end());
    put32(hdr, (uint32_t)n_targets);
    for (int32_t r = 0; r < n_targets; r++) {
        const std::string nm = target_name && target_name[r] ? target_name[r] : ("contig" + std::to_string(r));
        put32(hdr, (uint32_t)nm.size() + 1); hdr.insert(hdr.end(), nm.begin(), nm.end()); hdr.push_back(0);
        put32(hdr, target_len[r]);
    }
    const int64_t n = b->n_reads;
    auto nm_bytes = [](uint8_t t) -> size_t { return t == 0 ? 0 : 3 + ((t == 'c' || t == 'C') ? 1 : (t == 's' || t == 'S') ? 2 : 4); };
    std::vector<uint64_t> roff((size_t)n + 1, 0);
    parallel_for(T, n, [&](int, int64_t a, int64_t e) {
        for (int64_t k = a; k < e; k++) {
            const gce_core &c = b->core[k];
            size_t mi = 0;
            if (b->mi && b->mi_off && b->mi_off[k] != UINT64_MAX) mi = 3 + strlen(b->mi + b->mi_off[k]) + 1;
            roff[k + 1] = 4ull + 32 + c.l_qname + 4ull * c.n_cigar + (uint64_t)(c.l_qseq + 1) / 2 + (uint64_t)c.l_qseq + nm_bytes(b->nm_type[k]) + mi;
        }
    });
    for (int64_t k = 0; k < n; k++) roff[k + 1] += roff[k];
    Raw<uint8_t> body; body.resize(hdr.size() + roff[n]);
    if (!body.ok()) return GCE_ERR_OOM;
    memcpy(body.data(), hdr.data(), hdr.size());
    uint8_t *rb = body.data() + hdr.size();
    parallel_for(T, n, [&](int, int64_t a, int64_t e) {
        for (int64_t k = a; k < e; k++) {
            const gce_core &c = b->core[k];
            uint8_t *o = rb + roff[k];
            const uint32_t bs = (uint32_t)(roff[k + 1] - roff[k] - 4);
            memcpy(o, &bs, 4); memcpy(o + 4, &c, 32);
            uint8_t *w = o + 36;
            memcpy(w, b->qname + b->qname_off[k], c.l_qname); w += c.l_qname;
            memcpy(w, b->cigar + b->cigar_off[k], 4ull * c.n_cigar); w += 4ull * c.n_cigar;
            memcpy(w, b->seq + b->seq_off[k], (c.l_qseq + 1) / 2); w += (c.l_qseq + 1) / 2;
            memcpy(w, b->qual + b->qual_off[k], c.l_qseq); w += c.l_qseq;
            const uint8_t t = b->nm_type[k];
            if (t) {
                w[0] = 'N'; w[1] = 'M'; w[2] = t;
                const int32_t v = b->nm[k];
                const size_t sz = nm_bytes(t) - 3;
                memcpy(w + 3, &v, sz);                                           // little endian: the low bytes are the value
                w += 3 + sz;
            }
            if (b->mi && b->mi_off && b->mi_off[k] != UINT64_MAX) {
                const char *m = b->mi + b->mi_off[k]; const size_t ml = strlen(m) + 1;
                w[0] = 'M'; w[1] = 'I'; w[2] = 'Z'; memcpy(w + 3, m, ml); w += 3 + ml;
            }
        }
    });
    return write_bgzf(path, body, T, level);
}

// SAM text -> BAM and back on the host alone (no engine, no GPU): what sam_read1 / sam_write1 do when the reference is given SAM text
// (src/gencore.cpp:164-173,205,104 via htslib); gce_run_bam takes and writes SAM text through the same line functions (gce_samtext.hpp).
int gce_sam_to_bam(const char *sam_path, const char *bam_path, int threads, int level, char err[256]) {
    auto fail = [&](const char *m) { if (err) { strncpy(err, m, 255); err[255] = 0; } return GCE_ERR_INVALID; };
    if (err) err[0] = 0;
    if (!sam_path || !bam_path) return GCE_ERR_INVALID;
    const int T = threads > 0 ? threads : default_threads();
    std::vector<char> tx;
    if (!read_file(sam_path, tx)) return fail("cannot read the input SAM");
    if (!tx.empty() && tx.back() != '\n') tx.push_back('\n');
    const char *d = tx.data(); const size_t lim = tx.size();
    size_t p = 0; std::string text;
    while (p < lim && d[p] == '@') { const char *q = (const char *)memchr(d + p, '\n', lim - p); const size_t z = (size_t)(q - d) + 1; text.append(d + p, z - p); p = z; }
    std::vector<std::string> names; std::vector<uint32_t> lens;
    if (!samtext::parse_header_text(text, names, lens)) return fail("bad @SQ line");
    samtext::NameMap nmap; nmap.build(names);
    std::vector<std::vector<uint8_t>> parts((size_t)T); std::vector<std::string> perr((size_t)T);
    std::vector<size_t> cut((size_t)T + 1, lim); cut[0] = p;
    for (int t = 1; t < T; t++) { size_t c = p + (lim - p) * (size_t)t / (size_t)T; if (c > p) { const char *q = (const char *)memchr(d + c - 1, '\n', lim - (c - 1)); c = q ? (size_t)(q - d) + 1 : lim; } cut[(size_t)t] = std::max(c, cut[(size_t)t - 1]); }
    std::atomic<int> bad{0};
    parallel_for(T, T, [&](int, int64_t a, int64_t b2) {
        for (int64_t t = a; t < b2; t++) {
            size_t x = cut[(size_t)t]; const size_t xe = cut[(size_t)t + 1];
            while (x < xe) {
                const char *q = (const char *)memchr(d + x, '\n', lim - x); const size_t le = q ? (size_t)(q - d) : lim;
                if (le > x && !(le == x + 1 && d[x] == '\r') && !samtext::line_to_bam(d + x, d + le, nmap, parts[(size_t)t], perr[(size_t)t])) { bad = 1; return; }
                x = le + 1;
            }
        }
    });
    if (bad) { for (auto &m : perr) if (!m.empty()) return fail(m.c_str()); return fail("malformed SAM line"); }
    std::vector<uint8_t> hdr;
    auto put32 = [&](uint32_t x) { const uint8_t *q = (const uint8_t *)&x; hdr.insert(hdr.end(), q, q + 4); };
    hdr.insert(hdr.end(), {'B', 'A', 'M', 1});
    put32((uint32_t)text.size()); hdr.insert(hdr.end(), text.begin(), text.end());
    put32((uint32_t)lens.size());
    for (size_t r = 0; r < lens.size(); r++) { put32((uint32_t)names[r].size() + 1); hdr.insert(hdr.end(), names[r].begin(), names[r].end()); hdr.push_back(0); put32(lens[r]); }
    size_t tot = hdr.size(); for (auto &v : parts) tot += v.size();
    Raw<uint8_t> body; body.resize(tot);
    if (!body.ok()) return GCE_ERR_OOM;
    memcpy(body.data(), hdr.data(), hdr.size());
    size_t o = hdr.size(); for (auto &v : parts) { if (!v.empty()) memcpy(body.data() + o, v.data(), v.size()); o += v.size(); }
    const int rc = write_bgzf(bam_path, body, T, level);
    if (rc != GCE_OK) fail("cannot write the output BAM");
    return rc;
}

int gce_bam_to_sam(const char *bam_path, const char *sam_path, int threads, char err[256]) {
    auto fail = [&](const char *m) { if (err) { strncpy(err, m, 255); err[255] = 0; } return GCE_ERR_INVALID; };
    if (err) err[0] = 0;
    if (!bam_path || !sam_path) return GCE_ERR_INVALID;
    const int T = threads > 0 ? threads : default_threads();
    gce_bam *f = nullptr;
    const int rc = gce_bam_open(bam_path, T, &f);
    if (rc != GCE_OK) { fail(f ? f->err.c_str() : "cannot open the input BAM"); if (f) gce_bam_close(f); return rc; }
    FILE *fo = fopen(sam_path, "w");
    if (!fo) { gce_bam_close(f); return fail("cannot open the output SAM"); }
    const std::string ht = samtext::header_text_for_sam(f->text, f->names, f->lens);
    bool ok = fwrite(ht.data(), 1, ht.size(), fo) == ht.size();
    std::vector<std::string> lines((size_t)T);
    std::atomic<int> bad{0};
    const size_t nr = f->rec.size();
    parallel_for(T, T, [&](int, int64_t x, int64_t y) { for (int64_t t = x; t < y; t++) { std::string &L = lines[(size_t)t]; const size_t ra = nr * (size_t)t / (size_t)T, rb = nr * (size_t)(t + 1) / (size_t)T; for (size_t q = ra; q < rb; q++) if (!samtext::bam_to_line(f->u.data() + f->rec[q], f->names, L)) { bad = 1; return; } } });
    for (int t = 0; t < T && ok && !bad; t++) ok = fwrite(lines[(size_t)t].data(), 1, lines[(size_t)t].size(), fo) == lines[(size_t)t].size();
    ok = (fclose(fo) == 0) && ok;
    gce_bam_close(f);
    if (bad) return fail("bad record in the input BAM");
    return ok ? GCE_OK : fail("cannot write the output SAM");
}

// ------------------------------------------------------------------------------------------------------------ FASTA
// FastaReader(file) + readAll (src/fastareader.cpp:7-41,57-104,157-168) with its quirks: the FIRST character of every line is
// taken by get(c) and appended without the validity filter of str_keep_valid_sequence (util.h:194-210) -- an empty line inside a
// contig therefore contributes its '\n' as one base (code 0) and the line after it is taken whole; lower case is folded
// (forceUpperCase = true, fastareader.h:21); the contig ID is the header up to the first blank.  Later contigs of the same name
// replace earlier ones (std::map assignment).  Returns the contigs as upper-cased ASCII, ready for gce_set_reference_ascii.
//
// The walk of FastaReader::readNext is a chain of ITERATIONS: get(c) takes one character -- '>' ends the record, anything else goes
// unfiltered into the header (first iteration of a record) or the sequence --, then getline takes the rest of the line.  Where an
// iteration starts depends on the file only through its line feeds, so the file (read by parallel preads) is cut into one range per
// thread at positions that are PROVABLY iteration starts: a position q with d[q-1] == '\n' and d[q-2] neither '\n' nor '>' (that
// line feed cannot be the first character of an iteration, because the character in front of it neither ended an iteration nor was
// a one-character '>' iteration: it ends one, and q starts the next; q is never the start of a header iteration, which follows a
// '>').  Every thread then walks its iterations exactly as the reference does; the pieces are stitched per contig.  threads == 1 is
// the literal one-pass walk (the test oracle of the parallel one); a file without such cut positions falls back to it.
struct FaPiece { int64_t hdr_a = -1, hdr_b = -1; char *seq = nullptr; size_t n = 0; };     // one (part of a) record seen by one thread: a slice of the thread's arena
struct gce_fasta {
    std::vector<std::string> ids; std::vector<Raw<char>> seqs, arena; Raw<uint8_t> file; std::vector<const char *> idp, seqp; std::vector<int64_t> len;
    double t_map = 0, t_parse = 0, t_stitch = 0;
};

namespace {
inline char fa_upper(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - ('a' - 'A')) : c; }
// walk the iterations of [p, end): `end` is an iteration start (or the file's end).  `in_header`: the first iteration is a header's.
// Sequence characters are appended to pieces.back(); a '>' iteration opens a new piece whose header follows.
void fa_walk(const uint8_t *d, size_t n, size_t p, size_t end, bool in_header, std::vector<FaPiece> &pieces, char *arena) {
    // (an iteration never emits more characters than it consumes: the arena, as long as the range, cannot overflow)
    bool header_next = in_header;
    char *o = arena;
    pieces.back().seq = o;
    while (p < end) {
        const char c = (char)d[p++];
        if (c == '>') { pieces.back().n = (size_t)(o - pieces.back().seq); pieces.emplace_back(); pieces.back().seq = o; header_next = true; continue; }   // `if(c == '>' || eof) break;` -- in a header iteration too: that record has an empty header
        size_t e = p;
        if (e < n) { const void *nl = memchr(d + p, '\n', n - p); e = nl ? (size_t)((const uint8_t *)nl - d) : n; }
        if (header_next) { pieces.back().hdr_a = (int64_t)p - 1; pieces.back().hdr_b = (int64_t)e; header_next = false; }   // header = c + rest of the line
        else {
            *o++ = fa_upper(c);                                                                      // get(c): no validity filter
            for (size_t k = p; k < e; k++) { const char ch = fa_upper((char)d[k]); if ((ch >= 'A' && ch <= 'Z') || ch == '-' || ch == '*') *o++ = ch; }   // str_keep_valid_sequence (util.h:194-210); isalpha after the fold = A-Z
        }
        p = e < n ? e + 1 : n;
    }
    pieces.back().n = (size_t)(o - pieces.back().seq);
}
}  // namespace

int gce_fasta_load(const char *path, int threads, gce_fasta **out) {
    if (!path || !out) return GCE_ERR_INVALID;
    *out = nullptr;
    double t0 = now_s();
    int T = threads > 0 ? threads : default_threads();
    gce_fasta *fa = new gce_fasta();
    *out = fa;
    if (!read_file_parallel(path, fa->file, T)) return GCE_ERR_INVALID;          // parallel preads into one 2 MB-page buffer (page faults of an mmap'ed file
    const size_t n = fa->file.size();                                              //  from eight threads, next to their allocations, serialised on the mm lock: slower than one thread)
    const uint8_t *d = fa->file.data();
    fa->t_map = now_s() - t0; t0 = now_s();
    size_t p0 = 0;
    while (p0 < n && d[p0] != '>') p0++;                                      // seek to the first contig
    if (p0 < n) p0++;
    std::vector<std::vector<FaPiece>> per;
    if (p0 < n) {
        // ---- cut positions
        { const char *mb = getenv("GCE_FASTA_MIN_PARALLEL"); const size_t min_par = mb ? (size_t)atoll(mb) : (size_t)(1 << 20); if ((n - p0) < min_par) T = 1; }   // (tests set it to 0)
        std::vector<size_t> cut{p0};
        for (int t = 1; t < T; t++) {
            size_t q = std::max(p0 + (n - p0) / T * t, cut.back() + 2);
            const size_t lim = std::min(n, q + (size_t)(16 << 20));              // (a cut is found within a line or two; give up on odd files)
            for (; q < lim; q++) if (d[q - 1] == '\n' && d[q - 2] != '\n' && d[q - 2] != '>' && q - 2 >= p0) break;
            if (q >= lim) { cut.assign(1, p0); break; }                          // no provable iteration start: one thread walks it all
            if (q > cut.back()) cut.push_back(q);
        }
        const int nt = (int)cut.size();
        cut.push_back(n);
        per.resize(nt); fa->arena.resize(nt);
        for (int t = 0; t < nt; t++) { fa->arena[t].resize(cut[t + 1] - cut[t] + 64); if (!fa->arena[t].ok()) return GCE_ERR_OOM; }   // (all allocations in front of the threads)
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t] {
            per[t].emplace_back();                                               // thread 0: the first record; others: the record that is open at the cut
            fa_walk(d, n, cut[t], cut[t + 1], t == 0, per[t], fa->arena[t].data());
        });
        for (auto &x : th) x.join();
    }
    fa->t_parse = now_s() - t0; t0 = now_s();
    // ---- stitch: a thread's first piece continues the last piece of the thread in front of it (unless it is thread 0's)
    struct Rec { std::string hdr; std::vector<std::pair<int, int>> parts; size_t total = 0; };
    std::vector<Rec> recs;
    bool any = false;
    for (size_t t = 0; t < per.size(); t++)
        for (size_t k = 0; k < per[t].size(); k++) {
            FaPiece &pc = per[t][k];
            if (!(t > 0 && k == 0)) { recs.emplace_back(); if (pc.hdr_a >= 0) recs.back().hdr.assign((const char *)d + pc.hdr_a, (size_t)(pc.hdr_b - pc.hdr_a)); }
            recs.back().parts.emplace_back((int)t, (int)k); recs.back().total += pc.n; any = true;
        }
    (void)any;
    // readAll (fastareader.cpp:157-168): `while(!eof) readNext()` -- a record that a '>' opened right at the end of the file exists
    // (empty id, empty sequence) exactly when the walk above produced its piece; contigs of the same id: the LAST one stays, at the
    // place of the first (std::map assignment).
    std::unordered_map<std::string, size_t> where;
    std::vector<int64_t> rec_of;                                               // contig -> record that supplies its sequence
    for (size_t r = 0; r < recs.size(); r++) {
        const size_t sp = recs[r].hdr.find(' ');
        const std::string id = recs[r].hdr.substr(0, sp);
        auto it = where.find(id);
        if (it == where.end()) { where.emplace(id, fa->ids.size()); fa->ids.push_back(id); rec_of.push_back((int64_t)r); }
        else rec_of[it->second] = (int64_t)r;
    }
    fa->seqs.resize(fa->ids.size()); fa->seqp.assign(fa->ids.size(), nullptr); fa->len.assign(fa->ids.size(), 0);
    static const char empty_seq[1] = {0};
    {
        std::vector<std::thread> th;                                           // a contig inside one thread's range stays where it is; one that spans ranges is
        for (size_t c = 0; c < fa->ids.size(); c++) {                          // copied together, one thread per part
            Rec &r = recs[(size_t)rec_of[c]];
            fa->len[c] = (int64_t)r.total;
            if (r.total == 0) { fa->seqp[c] = empty_seq; continue; }
            if (r.parts.size() == 1) { fa->seqp[c] = per[r.parts[0].first][r.parts[0].second].seq; continue; }
            fa->seqs[c].resize(r.total + 1);
            if (!fa->seqs[c].ok()) { for (auto &x : th) x.join(); return GCE_ERR_OOM; }
            fa->seqp[c] = fa->seqs[c].p;
            size_t at = 0;
            for (auto &pr : r.parts) { const FaPiece *pc = &per[pr.first][pr.second]; char *dst = fa->seqs[c].p + at; if (pc->n) th.emplace_back([pc, dst] { memcpy(dst, pc->seq, pc->n); }); at += pc->n; }
        }
        for (auto &x : th) x.join();
    }
    fa->file.release();
    for (size_t k = 0; k < fa->ids.size(); k++) fa->idp.push_back(fa->ids[k].c_str());
    fa->t_stitch = now_s() - t0;
    if (getenv("GCE_FASTA_TIMING")) fprintf(stderr, "gce_fasta_load: %zu bytes, %d threads: map %.3f s, parse %.3f s, stitch %.3f s\n", n, (int)per.size(), fa->t_map, fa->t_parse, fa->t_stitch);
    return GCE_OK;
}
int gce_fasta_get(const gce_fasta *fa, int32_t *n, const char *const **ids, const char *const **seqs, const int64_t **lens) {
    if (!fa || !n) return GCE_ERR_INVALID;
    *n = (int32_t)fa->ids.size();
    if (ids) *ids = fa->idp.data();
    if (seqs) *seqs = fa->seqp.data();
    if (lens) *lens = fa->len.data();
    return GCE_OK;
}
void gce_fasta_free(gce_fasta *fa) { delete fa; }

// ------------------------------------------------------------------------------------------------------------ BED
// Bed::loadFromFile (src/bed.cpp:111-168) with util.h's trim / split (util.h:44-85).  An empty (or blank-only) line indexes an
// empty vector in the reference (undefined behaviour); it is skipped here.
int gce_bed_load(const char *path, int32_t n_targets, const char *const *target_name, int32_t *n_regions, int32_t **tid_out, int32_t **start_out,
                 int32_t **end_out, char ***name_out) {
    if (!path || !n_regions || !tid_out || !start_out || !end_out) return GCE_ERR_INVALID;
    std::vector<uint8_t> d;
    if (!read_file(path, d)) return GCE_ERR_INVALID;
    auto trim = [](const std::string &x) -> std::string {                          // spaces only (util.h:44-57)
        const size_t a = x.find_first_not_of(' ');
        if (a == std::string::npos) return "";
        const size_t b = x.find_last_not_of(' ');
        return x.substr(a, b - a + 1);
    };
    std::vector<int32_t> tids, starts, ends; std::vector<std::string> names;
    size_t p = 0; const size_t n = d.size();
    while (p < n) {                                                                // file.getline(line, 4096)
        size_t e = p;
        while (e < n && d[e] != '\n') e++;
        if (e - p > 4095) break;                                                   // the line does not fit the buffer: failbit, the loop ends
        std::string line((const char *)d.data() + p, e - p);
        const size_t nul = line.find('\0');                                        // strlen(line)
        if (nul != std::string::npos) line.resize(nul);
        p = e < n ? e + 1 : n;
        if (line.size() >= 2 && line.back() == '\r') { line.pop_back(); if (line.back() == '\r') line.pop_back(); }      // bed.cpp:126-133
        line = trim(line);
        std::vector<std::string> tok;                                              // split(linestr, "\t") (util.h:59-85)
        if (!line.empty()) {
            size_t b = line.find_first_not_of('\t');
            while (b != std::string::npos) {
                const size_t c = line.find('\t', b);
                if (c != std::string::npos) { tok.push_back(line.substr(b, c - b)); b = c + 1; }
                else { tok.push_back(line.substr(b)); b = c; }
            }
        }
        if (tok.empty()) continue;
        if (tok[0].compare(0, 1, "#") == 0) continue;                              // bed.cpp:139-140
        if (tok.size() < 3) continue;                                              // :142-143
        const std::string chr = trim(tok[0]);
        int tid = -1;
        for (int32_t t = 0; t < n_targets; t++) if (target_name && target_name[t] && chr == target_name[t]) tid = t;   // :154-162 (the last match wins)
        tids.push_back(tid); starts.push_back(atoi(trim(tok[1]).c_str())); ends.push_back(atoi(trim(tok[2]).c_str()));
        names.push_back(tok.size() > 3 ? trim(tok[3]) : "");
    }
    const size_t m = tids.size();
    *n_regions = (int32_t)m;
    *tid_out = (int32_t *)malloc(std::max<size_t>(m, 1) * 4); *start_out = (int32_t *)malloc(std::max<size_t>(m, 1) * 4); *end_out = (int32_t *)malloc(std::max<size_t>(m, 1) * 4);
    memcpy(*tid_out, tids.data(), m * 4); memcpy(*start_out, starts.data(), m * 4); memcpy(*end_out, ends.data(), m * 4);
    if (name_out) {
        *name_out = (char **)malloc(std::max<size_t>(m, 1) * sizeof(char *));
        for (size_t k = 0; k < m; k++) (*name_out)[k] = strdup(names[k].c_str());
    }
    return GCE_OK;
}
void gce_bed_free(int32_t n_regions, int32_t *tid, int32_t *start, int32_t *end, char **name) {
    free(tid); free(start); free(end);
    if (name) { for (int32_t k = 0; k < n_regions; k++) free(name[k]); free(name); }
}

// Gencore::consensus() for a sorted BAM (src/gencore.cpp:162-293) through the C-ABI.
// The whole-file path of round 2 (gce_bam_open: everything inflated and indexed on the host, struct-of-arrays chunks, gce_bam_write): kept as
// gce_run_bam_hostcodec for callers that want the host codec end to end and as the fallback of gce_run_bam.
int gce_run_bam_hostcodec(const char *in_path, const char *out_path, const char *fasta_path, const gce_params *params, int threads,
                int64_t chunk_reads, int level, gce_bam_run *out, char err[256]) {
    auto seterr = [&](const char *m) { if (err) { strncpy(err, m ? m : "", 255); err[255] = 0; } };
    seterr("");
    if (!in_path || !out_path || !params || !out) return GCE_ERR_INVALID;
    memset(out, 0, sizeof *out);
    const double t_start = now_s();
    gce_bam *f = nullptr; gce_engine *e = nullptr; gce_fasta *fa = nullptr;
    int rc = gce_bam_open(in_path, threads, &f);
    auto done = [&](int code, const char *m) { seterr(m); if (e) gce_destroy(e); if (f) gce_bam_close(f); if (fa) gce_fasta_free(fa); return code; };
    if (rc != GCE_OK) return done(rc, f ? gce_bam_error(f) : "open failed");
    out->open_s = now_s() - t_start;
    gce_bam_info bi; gce_bam_get_info(f, &bi);
    out->read_s = bi.read_s; out->inflate_s = bi.inflate_s; out->index_s = bi.index_s;
    gce_params prm = *params;
    prm.n_targets = bi.n_targets; prm.target_len = bi.target_len;
    if (strcmp(prm.umi_prefix, "auto") == 0) {                                   // src/gencore.cpp:207-220
        memset(prm.umi_prefix, 0, sizeof prm.umi_prefix);
        if (bi.n_records > 0) { gce_batch one; if (gce_bam_chunk(f, 0, 1, 0, &one) == GCE_OK) gce_detect_umi_prefix(one.qname, prm.umi_prefix); }
    }
    if ((rc = gce_create(&prm, &e)) != GCE_OK) return done(rc, gce_status_message(rc));
    if (fasta_path && *fasta_path) {
        if ((rc = gce_fasta_load(fasta_path, threads, &fa)) != GCE_OK) return done(rc, "cannot read the FASTA file");
        int32_t nc; const char *const *ids; const char *const *seqs; const int64_t *lens;
        gce_fasta_get(fa, &nc, &ids, &seqs, &lens);
        for (int32_t t = 0; t < bi.n_targets; t++)                               // Reference::getData looks contigs up by BAM target name (reference.cpp:43-53)
            for (int32_t c = 0; c < nc; c++)
                if (strcmp(ids[c], bi.target_name[t]) == 0 && (rc = gce_set_reference_ascii(e, t, seqs[c], lens[c])) != GCE_OK) return done(rc, gce_last_error(e));
    }
    double t0 = now_s();
    if ((rc = gce_reserve(e, bi.n_records, bi.qname_bytes, bi.cigar_words, bi.seq_bytes, bi.qual_bytes)) != GCE_OK) return done(rc, gce_last_error(e));     // (MI tags travel on the streamed path since round 5)
    if (chunk_reads <= 0) chunk_reads = 1 << 21;
    int32_t tickets[2] = {-1, -1};
    int64_t k = 0;
    for (int64_t first = 0; first < bi.n_records; first += chunk_reads, k++) {
        const int sl = (int)(k & 1);
        if (tickets[sl] >= 0 && (rc = gce_submit_wait(e, tickets[sl])) != GCE_OK) return done(rc, gce_last_error(e));   // the slot's previous copy
        gce_batch b;
        if ((rc = gce_bam_chunk(f, first, std::min(chunk_reads, bi.n_records - first), sl, &b)) != GCE_OK) return done(rc, "chunk");
        rc = gce_submit_async(e, &b, &tickets[sl]);
        if (rc != GCE_OK) return done(rc, gce_last_error(e));
    }
    out->submit_s = now_s() - t0; t0 = now_s();
    if (bi.n_records > 0) {
        if ((rc = gce_process(e)) != GCE_OK) return done(rc, gce_last_error(e)[0] ? gce_last_error(e) : gce_status_message(rc));
        out->process_s = now_s() - t0; t0 = now_s();
        gce_timing tm; if (gce_get_timing(e, &tm) == GCE_OK) out->kernel_ms = tm.total_ms;
        gce_result res;
        if ((rc = gce_drain(e, &res)) != GCE_OK) return done(rc, gce_last_error(e));
        out->drain_s = now_s() - t0; t0 = now_s();
        out->n_reads = res.n_reads; out->n_out = res.n_out; out->pre = res.pre; out->post = res.post;
        if ((rc = gce_bam_write(out_path, f, &res, threads, level)) != GCE_OK) return done(rc, "cannot write the output BAM");
    } else {
        gce_result res; memset(&res, 0, sizeof res);
        if ((rc = gce_bam_write(out_path, f, &res, threads, level)) != GCE_OK) return done(rc, "cannot write the output BAM");
    }
    out->write_s = now_s() - t0;
    out->total_s = now_s() - t_start;
    return done(GCE_OK, "");
}



// ---- the streaming, GPU-assisted file path
int gce_raw_begin(gce_engine *e, size_t capacity_hint);
int gce_raw_push(gce_engine *e, const void *host, size_t bytes, int32_t *ticket);
int gce_raw_finish(gce_engine *e, uint64_t records_begin, int32_t n_ref, int64_t *n_records);
int gce_raw_build_output(gce_engine *e, uint64_t *body_bytes, int64_t *n_out);
int gce_raw_read_output_async(gce_engine *e, uint64_t offset, void *host, size_t bytes, int32_t *ticket);
int gce_host_alloc(size_t bytes, void **out);
int gce_raw_deflate_output(gce_engine *e, uint64_t *comp_bytes);
int gce_raw_read_deflated_async(gce_engine *e, uint64_t offset, void *host, size_t bytes, int32_t *ticket);
int gce_raw_attach_mirror(gce_engine *e, gce_engine *mirror);
int gce_raw_select_shard(gce_engine *e, int32_t world, int32_t rank, int32_t plan_mode);
int gce_stats_payload_device(gce_engine *e, int32_t coverage_step, int32_t n_regions, const int32_t *region_tid, const int32_t *region_start, const int32_t *region_end, const int64_t **payload, gce_payload_layout *layout);
int gce_stats_payload_sum(gce_engine **engs, int32_t n_engs, const int64_t **payload, gce_payload_layout *layout);
int gce_stats_payload_read(gce_engine *e, const int64_t *payload, int64_t n_words, int64_t *host);
int gce_raw_merge_outputs(gce_engine **engs, int32_t n_engs, uint64_t *body_bytes, int64_t *n_out_total, gce_stats *pre, gce_stats *post, int64_t *n_reads_total);
void gce_host_free(void *p);
}  // extern "C" (declarations)
extern "C++" {
namespace {
struct Pinned {                                   // a pinned host buffer that grows
    uint8_t *p = nullptr; size_t cap = 0;
    ~Pinned() { gce_host_free(p); }
    bool ensure(size_t n) { if (n <= cap) return true; gce_host_free(p); p = nullptr; cap = 0; void *q = nullptr; if (gce_host_alloc(n + (n >> 3) + 4096, &q) != GCE_OK) return false; p = (uint8_t *)q; cap = n + (n >> 3) + 4096; return true; }
};
long status_kb(const char *key) { FILE *f = fopen("/proc/self/status", "r"); if (!f) return 0; char line[256]; long v = 0; const size_t kl = strlen(key); while (fgets(line, sizeof line, f)) if (strncmp(line, key, kl) == 0) { v = atol(line + kl); break; } fclose(f); return v; }
}  // namespace
}  // extern "C++"
extern "C" {

// Replaces Gencore::consensus() end to end (src/gencore.cpp:162-293) as a PIPELINE with a bounded host footprint: a reader thread preads the
// file in pieces; the host threads inflate the BGZF blocks of piece k (own decoder + CLMUL CRC) into a pinned window while the DMA engine
// copies window k - 1 into HBM; nothing of the input stays on the host.  Records are indexed, parsed (gce_raw_finish) and -- after
// gce_process -- re-assembled as BAM records (gce_raw_build_output) on the GPU; the output stream comes back in pieces that are deflated by
// all host threads and written in order while the next piece is on its way.  Host memory: two compressed pieces, three inflated windows,
// three output pieces -- independent of the file's size (round 2 held the whole inflated file and every record table on the host).
// gce_run_bam (n_shards == 1) and gce_run_bam_sharded over the same pipeline: with several shards every piece of the file goes to one engine per
// entry of `devices` (the mirrors of the first), each inflates and indexes the stream on its own GPU, plans it there and keeps its share
// (gce_raw_select_shard); the record streams are merged on the first engine's device (gce_raw_merge_outputs) and written as one.
static int run_bam_impl(const char *in_path, const char *out_path, const char *fasta_path, const gce_params *params, int threads,
                        int64_t chunk_reads, int level, gce_bam_run *out, char err[256], int32_t n_shards, const int32_t *devices, int32_t plan_mode,
                        const char *bed_path = nullptr, int32_t coverage_step = 0, gce_depth_run *depth = nullptr) {
    auto seterr = [&](const char *m) { if (err) { strncpy(err, m ? m : "", 255); err[255] = 0; } };
    seterr("");
    if (!in_path || !out_path || !params || !out) return GCE_ERR_INVALID;
    memset(out, 0, sizeof *out);
    out->rss_start_kb = status_kb("VmRSS:");
    const double t_start = now_s();
    const int T = threads > 0 ? threads : default_threads();
    const int fd = open(in_path, O_RDONLY);
    if (fd < 0) { seterr("cannot open the input BAM"); return GCE_ERR_INVALID; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 0) { close(fd); seterr("cannot stat the input BAM"); return GCE_ERR_INVALID; }
    const uint64_t fsz = (uint64_t)st.st_size;
    gce_engine *e = nullptr; gce_fasta *fa = nullptr; FILE *fo = nullptr;
    std::vector<gce_engine *> mir;                      // the engines of shards 1 .. n_shards - 1 (they receive every push made to e)
    auto done = [&](int code, const char *m) { seterr(m); for (auto *x : mir) if (x) gce_destroy(x); if (e) gce_destroy(e); if (fa) gce_fasta_free(fa); if (fo) fclose(fo); close(fd); return code; };
    const size_t PIECE = (size_t)(chunk_reads > 0 && chunk_reads < (1 << 16) ? (1 << 20) : (8 << 20));       // compressed bytes per window (tests shrink it through chunk_reads)
    // GCE_BAM_HOST_INFLATE=1: the BGZF members are inflated by the host threads (the path of the first half of round 3); default: they go to
    // HBM compressed and the GPU inflates them (gce_raw_push_bgzf) -- the host inflates only the window(s) that hold the BAM header
    const bool gpu_inflate = getenv("GCE_BAM_HOST_INFLATE") == nullptr;
    const bool tlap = getenv("GCE_RAW_TIMING") != nullptr; double tl0 = now_s();
    auto lap = [&](const char *what) { if (tlap) { const double x = now_s(); fprintf(stderr, "gce_run_bam %s %.4f s\n", what, x - tl0); tl0 = x; } };
    Pinned comp[2]; int32_t comp_ticket[2] = {-1, -1};
    if (!comp[0].ensure(PIECE + (1 << 17)) || !comp[1].ensure(PIECE + (1 << 17))) return done(GCE_ERR_OOM, "out of pinned host memory");
    std::vector<uint64_t> z_coff; std::vector<uint32_t> z_csize, z_usize;
    lap("compressed-piece buffers");
    Pinned win[3]; int32_t win_ticket[3] = {-1, -1, -1};
    // reader: piece k of the file into comp[k & 1] behind the carry-over of piece k - 1 (a BGZF block cut by the piece border)
    uint64_t file_off = 0; size_t carry = 0; double t_read = 0, t_inflate = 0, t_wait = 0;
    std::vector<Block> blocks;
    std::vector<std::string> names; std::vector<uint32_t> lens; std::string text;
    uint64_t hdr_end = 0; bool have_header = false;
    gce_params prm = *params;
    Raw<uint8_t> head;                                  // the inflated start of the stream until the header is complete (usually one window)
    std::thread reader; ssize_t got_next = 0; bool reader_on = false;
    auto start_read = [&](int slot, size_t keep) {
        const uint64_t at = file_off; const size_t want = (size_t)std::min<uint64_t>(PIECE, fsz - at);
        reader_on = true;
        reader = std::thread([&, slot, keep, at, want] {                                    // the piece in up to four parts, read side by side (one pread stream copies ~7 GB/s out of the page cache)
            const double r0 = now_s();
            static const size_t RP = getenv("GCE_READ_PARTS") ? (size_t)atoi(getenv("GCE_READ_PARTS")) : 4;
            const int R = (int)std::max<size_t>(1, std::min<size_t>({RP, (size_t)T, want >> 20}));
            std::vector<size_t> done_(R, 0); std::vector<std::thread> sub;
            auto part = [&](int r) { const size_t a = want * (size_t)r / R, z2 = want * (size_t)(r + 1) / R; size_t o = a; while (o < z2) { const ssize_t g = pread(fd, comp[slot].p + keep + o, z2 - o, (off_t)(at + o)); if (g <= 0) break; o += (size_t)g; } done_[r] = o - a; };
            for (int r = 1; r < R; r++) sub.emplace_back(part, r);
            part(0);
            for (auto &t : sub) t.join();
            size_t o = 0; for (int r = 0; r < R; r++) { const size_t a = want * (size_t)r / R, z2 = want * (size_t)(r + 1) / R; o += done_[r]; if (done_[r] != z2 - a) break; }      // (bytes in order up to the first short part)
            got_next = (ssize_t)o; t_read += now_s() - r0; });
        file_off += want;
    };
    int rc = GCE_OK;
    std::string emsg;
    // the contig table is known (BAM header / SAM header lines): engine, reference, raw stream.  first_qname: the first record's name or NULL
    auto setup_engine = [&](const char *first_qname, size_t capacity) -> int {
        have_header = true;
        prm.n_targets = (int32_t)lens.size(); prm.target_len = lens.data();
        if (strcmp(prm.umi_prefix, "auto") == 0) {                                       // src/gencore.cpp:207-220: the first record's name
            memset(prm.umi_prefix, 0, sizeof prm.umi_prefix);
            if (first_qname) gce_detect_umi_prefix(first_qname, prm.umi_prefix);
        }
        if (getenv("GCE_RAW_TIMING")) fprintf(stderr, "gce_run_bam: RSS before gce_create %ld MB (entry %ld MB)\n", status_kb("VmRSS:") >> 10, (long)(out->rss_start_kb >> 10));
        lap("up to the header");
        int r2;
        if (devices) prm.device = devices[0];
        // One host thread per engine: gce_create, the reference, and the raw stream's buffers (gce_raw_begin: a few GB of hipMalloc per engine, 0.15 s per GB on a
        // fresh process -- the four engines of a sharded run, set up one after the other, were 0.4 - 1.7 s of "input pipeline" on some boxes) side by side; on a
        // multi-GPU node every device allocates for itself.  The FASTA file is read once, in front.
        int32_t nc = 0; const char *const *ids = nullptr; const char *const *seqs = nullptr; const int64_t *flen = nullptr;
        if (fasta_path && *fasta_path) {
            if ((r2 = gce_fasta_load(fasta_path, threads, &fa)) != GCE_OK) { emsg = "cannot read the FASTA file"; return r2; }
            gce_fasta_get(fa, &nc, &ids, &seqs, &flen);
        }
        std::vector<gce_engine *> made((size_t)std::max(n_shards, 1), nullptr);
        std::vector<int> rcs((size_t)made.size(), GCE_OK); std::vector<std::string> msgs(made.size());
        auto setup = [&](int32_t r) {
            gce_params pr = prm; if (devices) pr.device = devices[r];
            gce_engine *x = nullptr; int c2;
            if ((c2 = gce_create(&pr, &x)) != GCE_OK) { rcs[(size_t)r] = c2; msgs[(size_t)r] = gce_status_message(c2); return; }
            made[(size_t)r] = x;
            for (size_t t = 0; t < lens.size() && fa; t++)                               // Reference::getData looks contigs up by BAM target name (reference.cpp:43-53)
                for (int32_t c = 0; c < nc; c++)
                    if (names[t] == ids[c] && (c2 = gce_set_reference_ascii(x, (int32_t)t, seqs[c], flen[c])) != GCE_OK) { rcs[(size_t)r] = c2; msgs[(size_t)r] = gce_last_error(x); return; }
            if ((c2 = gce_raw_begin(x, capacity)) != GCE_OK) { rcs[(size_t)r] = c2; msgs[(size_t)r] = gce_last_error(x); }
        };
        if (made.size() == 1) setup(0);
        else { std::vector<std::thread> th; for (int32_t r = 0; r < n_shards; r++) th.emplace_back(setup, r); for (auto &t : th) t.join(); }
        e = made[0]; for (size_t r = 1; r < made.size(); r++) if (made[r]) mir.push_back(made[r]);       // (owned by `done` from here on)
        if (fa) { gce_fasta_free(fa); fa = nullptr; }                                    // (packed in HBM: the host copy goes)
        for (size_t r = 0; r < made.size(); r++) if (rcs[r] != GCE_OK) { emsg = msgs[r]; return rcs[r]; }
        lap("gce_create + reference + gce_raw_begin (a thread per engine)");
        for (gce_engine *x : mir) if ((r2 = gce_raw_attach_mirror(e, x)) != GCE_OK) { emsg = gce_last_error(x); return r2; }
        return GCE_OK;
    };
    auto bam_header_bytes = [&]() {                                                       // BAM magic, text, contig table (SAMv1 4.2)
        std::vector<uint8_t> hdr;
        auto put32 = [&](uint32_t x) { const uint8_t *p = (const uint8_t *)&x; hdr.insert(hdr.end(), p, p + 4); };
        hdr.insert(hdr.end(), {'B', 'A', 'M', 1});
        put32((uint32_t)text.size()); hdr.insert(hdr.end(), text.begin(), text.end());
        put32((uint32_t)lens.size());
        for (size_t r = 0; r < lens.size(); r++) { put32((uint32_t)names[r].size() + 1); hdr.insert(hdr.end(), names[r].begin(), names[r].end()); hdr.push_back(0); put32(lens[r]); }
        return hdr;
    };
    uint64_t pushed = 0;
    // sam_open(in, "r") takes either format (src/gencore.cpp:164): a file that does not start with the gzip magic is SAM text
    bool is_sam = false;
    { uint8_t m2[2] = {0, 0}; is_sam = fsz > 0 && !(fsz >= 2 && pread(fd, m2, 2, 0) == 2 && m2[0] == 0x1f && m2[1] == 0x8b); }
    if (is_sam) {
        // Pieces of the text are cut at line feeds; the '@' lines in front give the header text and the contig table; alignment lines become BAM
        // records on all host threads (gce_samtext.hpp) in a pinned window that goes to HBM like an inflated BAM window, behind BAM header bytes
        // made from the SAM header: from there on the stream is the one a BAM file gives.
        Raw<char> tbuf[2]; uint64_t at = 0; bool in_header = true; samtext::NameMap nmap; int wk = 0, kb = 0;
        std::vector<std::vector<uint8_t>> parts((size_t)T); std::vector<std::string> perr((size_t)T);
        const size_t TP = PIECE < ((size_t)8 << 20) ? PIECE : ((size_t)64 << 20);          // text bytes per piece (tests: 1 MB pieces that cut lines)
        std::thread rd; bool rd_on = false; ssize_t rd_got = 0;
        auto read_into = [&](char *dst, size_t want, uint64_t off) { const double r0 = now_s(); size_t o = 0; while (o < want) { const ssize_t g = pread(fd, dst + o, want - o, (off_t)(off + o)); if (g <= 0) break; o += (size_t)g; } rd_got = (ssize_t)o; t_read += now_s() - r0; };
        auto bail = [&](int code, const char *m) { if (rd_on) { rd.join(); rd_on = false; } return done(code, m); };
        size_t n = 0;                                                                      // bytes in tbuf[kb]: what the last piece left over + this piece
        {
            const size_t want = (size_t)std::min<uint64_t>(TP, fsz);
            tbuf[0].resize(want + 1); if (!tbuf[0].ok()) return done(GCE_ERR_OOM, "out of host memory");
            read_into(tbuf[0].data(), want, 0);
            if ((size_t)rd_got != want) return done(GCE_ERR_INVALID, "cannot read the input SAM");
            at = want; n = want;
        }
        for (;;) {
            char *cur = tbuf[kb].data();
            const bool last = at >= fsz;
            size_t lim = n;
            if (!last) {
                const char *nl = (const char *)memrchr(cur, '\n', n);
                lim = nl ? (size_t)(nl - cur) + 1 : 0;
                if (!nl && n > ((size_t)256 << 20)) return done(GCE_ERR_INVALID, "SAM line longer than 256 MB");
            } else if (n && cur[n - 1] != '\n') { cur[n] = '\n'; lim = n + 1; }
            // the next piece is read behind what this one leaves over (the line its end cut) while this one is converted
            size_t want2 = 0, left = lim >= n ? 0 : n - lim;
            if (!last) {
                want2 = (size_t)std::min<uint64_t>(TP, fsz - at);
                Raw<char> &nx = tbuf[kb ^ 1];
                nx.resize(left + want2 + 1); if (!nx.ok()) return done(GCE_ERR_OOM, "out of host memory");
                if (left) memcpy(nx.data(), cur + lim, left);
                char *dst = nx.data() + left; const uint64_t off = at;
                rd_on = true; rd = std::thread([&, dst, off, want2] { read_into(dst, want2, off); });
                at += want2;
            }
            size_t p = 0;
            if (in_header) {
                while (p < lim && cur[p] == '@') { const char *q = (const char *)memchr(cur + p, '\n', lim - p); const size_t z2 = (size_t)(q - cur) + 1; text.append(cur + p, z2 - p); p = z2; }
                if (p < lim || last) {
                    in_header = false;
                    if (!samtext::parse_header_text(text, names, lens) || lens.empty()) return bail(GCE_ERR_INVALID, "this SAM file has no header");      // src/gencore.cpp:186-189
                    nmap.build(names);
                    std::string fq;
                    if (p < lim) { const char *q = cur + p; const char *t = (const char *)memchr(q, '\t', lim - p); if (t) fq.assign(q, t); }
                    if ((rc = setup_engine(fq.empty() ? nullptr : fq.c_str(), (size_t)std::max<uint64_t>(fsz + (1u << 20), 1u << 20))) != GCE_OK) return bail(rc, emsg.c_str());
                    const std::vector<uint8_t> hb = bam_header_bytes();
                    hdr_end = hb.size();
                    int32_t tk; if ((rc = gce_raw_push(e, hb.data(), hb.size(), &tk)) != GCE_OK || (rc = gce_submit_wait(e, tk)) != GCE_OK) return bail(rc, gce_last_error(e));
                    pushed += hb.size();
                }
            }
            if (!in_header && p < lim) {
                const double i0 = now_s();
                const char *d = cur;
                std::vector<size_t> cut((size_t)T + 1, lim);                              // thread t converts the lines that START in [cut[t], cut[t + 1])
                cut[0] = p;
                for (int t = 1; t < T; t++) { size_t c = p + (lim - p) * (size_t)t / (size_t)T; if (c > p) { const char *q = (const char *)memchr(d + c - 1, '\n', lim - (c - 1)); c = q ? (size_t)(q - d) + 1 : lim; } cut[(size_t)t] = std::max(c, cut[(size_t)t - 1]); }
                std::atomic<int> bad{0};
                parallel_for(T, T, [&](int, int64_t a, int64_t b2) {
                    for (int64_t t = a; t < b2; t++) {
                        std::vector<uint8_t> &o = parts[(size_t)t]; o.clear();
                        size_t x = cut[(size_t)t]; const size_t xe = cut[(size_t)t + 1];
                        while (x < xe) {
                            const char *q = (const char *)memchr(d + x, '\n', lim - x); const size_t le = q ? (size_t)(q - d) : lim;
                            if (le > x && !(le == x + 1 && d[x] == '\r') && !samtext::line_to_bam(d + x, d + le, nmap, o, perr[(size_t)t])) { bad = 1; return; }
                            x = le + 1;
                        }
                    }
                });
                if (bad) { for (auto &m : perr) if (!m.empty()) return bail(GCE_ERR_INVALID, m.c_str()); return bail(GCE_ERR_INVALID, "malformed SAM line"); }
                size_t tot = 0; std::vector<size_t> po((size_t)T + 1, 0);
                for (int t = 0; t < T; t++) { po[(size_t)t] = tot; tot += parts[(size_t)t].size(); }
                if (tot) {
                    const int ws = wk % 3;
                    if (win_ticket[ws] >= 0) { const double w0 = now_s(); if ((rc = gce_submit_wait(e, win_ticket[ws])) != GCE_OK) return bail(rc, gce_last_error(e)); t_wait += now_s() - w0; win_ticket[ws] = -1; }
                    if (!win[ws].ensure(tot + 64)) return bail(GCE_ERR_OOM, "out of pinned host memory");
                    parallel_for(T, T, [&](int, int64_t a, int64_t b2) { for (int64_t t = a; t < b2; t++) if (!parts[(size_t)t].empty()) memcpy(win[ws].p + po[(size_t)t], parts[(size_t)t].data(), parts[(size_t)t].size()); });
                    if ((rc = gce_raw_push(e, win[ws].p, tot, &win_ticket[ws])) != GCE_OK) return bail(rc, gce_last_error(e));
                    pushed += tot; wk++;
                }
                t_inflate += now_s() - i0;                                                 // (lines -> records: reported where a BAM input reports its inflate)
            }
            if (last) break;
            rd.join(); rd_on = false;
            if ((size_t)rd_got != want2) return done(GCE_ERR_INVALID, "cannot read the input SAM");
            n = left + want2; kb ^= 1;
        }
    } else {
    int k = 0;
    size_t have = 0;                                    // bytes in comp[k & 1]: carry + piece
    if (fsz) { start_read(0, 0); reader.join(); reader_on = false; have = (size_t)got_next; }
    while (have > 0) {
        const int cs = k & 1, ws = k % 3;
        uint8_t *z = comp[cs].p;
        // ---- BGZF members of this piece
        blocks.clear();
        size_t off = 0; uint64_t uoff = 0;
        while (off + 18 <= have) {
            const uint8_t *p = z + off;
            if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return done(GCE_ERR_INVALID, "not a BGZF file");
            const uint16_t xlen = rd16(p + 10);
            if (off + 12 + (size_t)xlen > have) break;
            uint32_t bsize = 0; bool found = false;
            for (uint32_t x = 0; x + 4 <= xlen; ) {
                const uint8_t *sf = p + 12 + x; const uint16_t sl = rd16(sf + 2);
                if (x + 4 + (uint32_t)sl > xlen) break;
                if (sf[0] == 'B' && sf[1] == 'C' && sl == 2) { bsize = (uint32_t)rd16(sf + 4) + 1; found = true; }
                x += 4 + sl;
            }
            if (!found || bsize < 12u + xlen + 8u) return done(GCE_ERR_INVALID, "bad BGZF block");
            if (off + bsize > have) break;                                              // cut by the piece border: carried over
            Block b; b.coff = off; b.csize = bsize; b.usize = rd32(p + bsize - 4); b.uoff = uoff;
            if (b.usize > 0x10000u) return done(GCE_ERR_INVALID, "bad BGZF block (ISIZE above 64 KB)");
            blocks.push_back(b); off += bsize; uoff += b.usize;
        }
        const bool last = file_off >= fsz;
        if (last && off != have) return done(GCE_ERR_INVALID, "truncated BGZF block at the end of the file");
        if (!last && blocks.empty()) return done(GCE_ERR_INVALID, "BGZF block larger than a window");
        // ---- the next piece is read while this one is inflated
        carry = have - off;
        if (!last) {
            if (comp_ticket[cs ^ 1] >= 0) { const double w0 = now_s(); if ((rc = gce_submit_wait(e, comp_ticket[cs ^ 1])) != GCE_OK) return done(rc, gce_last_error(e)); t_wait += now_s() - w0; comp_ticket[cs ^ 1] = -1; }   // (its members are in HBM)
            memcpy(comp[cs ^ 1].p, z + off, carry); start_read(cs ^ 1, carry);
        }
        if (have_header && gpu_inflate) {                                               // this piece's members: to HBM as they are
            z_coff.clear(); z_csize.clear(); z_usize.clear();
            for (const Block &bk : blocks) { z_coff.push_back(bk.coff); z_csize.push_back(bk.csize); z_usize.push_back(bk.usize); }
            if ((rc = gce_raw_push_bgzf(e, z, off, (int32_t)blocks.size(), z_coff.data(), z_csize.data(), z_usize.data(), &comp_ticket[cs])) != GCE_OK) { if (reader_on) reader.join(); return done(rc, gce_last_error(e)); }
            pushed += uoff;
            if (reader_on) { reader.join(); reader_on = false; have = carry + (size_t)got_next; } else have = 0;
            k++;
            continue;
        }
        if (win_ticket[ws] >= 0) { const double w0 = now_s(); if ((rc = gce_submit_wait(e, win_ticket[ws])) != GCE_OK) { if (reader_on) reader.join(); return done(rc, gce_last_error(e)); } t_wait += now_s() - w0; win_ticket[ws] = -1; }
        if (!win[ws].ensure((size_t)uoff + 64)) { if (reader_on) reader.join(); return done(GCE_ERR_OOM, "out of pinned host memory"); }
        const double i0 = now_s();
        std::atomic<int> bad{0};
        parallel_for(T, (int64_t)blocks.size(), [&](int, int64_t a, int64_t b2) { for (int64_t q = a; q < b2; q++) if (blocks[q].usize && !inflate_block(z + blocks[q].coff, blocks[q], win[ws].p + blocks[q].uoff)) bad = 1; });
        t_inflate += now_s() - i0;
        if (bad) { if (reader_on) reader.join(); return done(GCE_ERR_INVALID, "inflate / CRC failure"); }
        // ---- header (first window(s)), engine, reference
        if (!have_header) {
            const size_t old = head.size();
            Raw<uint8_t> h2; h2.resize(old + (size_t)uoff + 1); if (!h2.ok()) { if (reader_on) reader.join(); return done(GCE_ERR_OOM, "out of host memory"); }
            if (old) memcpy(h2.data(), head.data(), old);
            memcpy(h2.data() + old, win[ws].p, (size_t)uoff); h2.n = old + (size_t)uoff; head = std::move(h2);
            const uint8_t *u = head.data(); const uint64_t n = head.size();
            bool complete = false;
            if (n >= 12 && memcmp(u, "BAM\1", 4) != 0) { if (reader_on) reader.join(); return done(GCE_ERR_INVALID, "not a BAM stream"); }
            if (n >= 12) {
                uint64_t p = 4; const uint32_t l_text = rd32(u + p); p += 4;
                if (p + l_text + 4 <= n) {
                    const uint64_t tp = p; p += l_text;
                    const uint32_t n_ref = rd32(u + p); p += 4;
                    names.clear(); lens.clear(); bool ok = true;
                    for (uint32_t r = 0; r < n_ref && ok; r++) {
                        if (p + 4 > n) { ok = false; break; }
                        const uint32_t ln = rd32(u + p); p += 4;
                        if (ln == 0 || p + ln + 4 > n) { ok = false; break; }
                        names.emplace_back((const char *)u + p, ln - 1); p += ln; lens.push_back(rd32(u + p)); p += 4;
                    }
                    if (ok && lens.empty()) { if (reader_on) reader.join(); return done(GCE_ERR_INVALID, "this SAM file has no header"); }      // src/gencore.cpp:186-189 (n_targets == 0), as the SAM-text branch does
                    if (ok) { complete = true; hdr_end = p; text.assign((const char *)u + tp, l_text); }
                }
            }
            if (!complete && last) { if (reader_on) reader.join(); return done(GCE_ERR_INVALID, "truncated header"); }
            if (complete) {
                const char *fq = nullptr;                                                 // src/gencore.cpp:207-220: the first record's name
                if (hdr_end + 36 < n) { const uint32_t lq = u[hdr_end + 12]; if (hdr_end + 36 + lq <= n) fq = (const char *)u + hdr_end + 36; }
                if ((rc = setup_engine(fq, (size_t)std::max<uint64_t>(fsz * 5, head.size()))) != GCE_OK) { if (reader_on) reader.join(); return done(rc, emsg.c_str()); }
                // what was inflated so far goes up in one piece (normally: this very window)
                if (old) { int32_t tk; if ((rc = gce_raw_push(e, head.data(), old, &tk)) != GCE_OK || (rc = gce_submit_wait(e, tk)) != GCE_OK) { if (reader_on) reader.join(); return done(rc, gce_last_error(e)); } pushed += old; }
                head.release();
                lap("gce_raw_begin + first push");
            }
        }
        if (have_header && uoff) {
            if ((rc = gce_raw_push(e, win[ws].p, (size_t)uoff, &win_ticket[ws])) != GCE_OK) { if (reader_on) reader.join(); return done(rc, gce_last_error(e)); }
            pushed += uoff;
        }
        if (reader_on) { reader.join(); reader_on = false; have = carry + (size_t)got_next; } else have = 0;
        k++;
    }
    }   // (BAM input)
    if (!have_header) return done(GCE_ERR_INVALID, fsz ? "truncated header" : "empty file");
    lap("rest of the input loop");
    out->read_s = t_read; out->inflate_s = t_inflate; out->submit_s = t_wait;
    out->open_s = now_s() - t_start;
    if (getenv("GCE_RAW_TIMING")) fprintf(stderr, "gce_run_bam: RSS after the input pipeline %ld MB\n", status_kb("VmRSS:") >> 10);
    double t0 = now_s();
    int64_t n_rec = 0;
    uint64_t body = 0; int64_t n_out = 0;
    // ---- the depth statistics of the report (Options::coverageStep, Options::bedFile; stats.cpp:56-83, bed.cpp:64-79): every engine adds up its own reads and
    //      records on its GPU right behind its run -- BEFORE the merge adds the other engines' Stats blocks into the first one's -- (gce_stats_payload_device)
    if (depth) {
        memset(depth, 0, sizeof *depth);
        if (coverage_step <= 0) return done(GCE_ERR_INVALID, "coverage_step must be positive");
        if (bed_path && *bed_path) {
            std::vector<const char *> np; for (auto &x : names) np.push_back(x.c_str());
            if ((rc = gce_bed_load(bed_path, (int32_t)np.size(), np.data(), &depth->n_regions, &depth->region_tid, &depth->region_start, &depth->region_end, nullptr)) != GCE_OK) return done(rc, "cannot read the BED file");
        }
        const int nt = (int)lens.size();
        depth->n_targets = nt;
        depth->bin_off = (int64_t *)calloc((size_t)nt + 1, 8);
        if (!depth->bin_off) return done(GCE_ERR_OOM, "out of host memory");
        for (int t = 0; t < nt; t++) depth->bin_off[t + 1] = depth->bin_off[t] + 1 + (int64_t)lens[(size_t)t] / coverage_step;
        const int64_t nb = depth->bin_off[nt];
        depth->n_bins = nb;
        depth->pre_depth = (int64_t *)calloc((size_t)std::max<int64_t>(nb, 1), 8); depth->post_depth = (int64_t *)calloc((size_t)std::max<int64_t>(nb, 1), 8);
        depth->pre_bed = (int64_t *)calloc((size_t)std::max(depth->n_regions, 1), 8); depth->post_bed = (int64_t *)calloc((size_t)std::max(depth->n_regions, 1), 8);
        if (!depth->pre_depth || !depth->post_depth || !depth->pre_bed || !depth->post_bed) return done(GCE_ERR_OOM, "out of host memory");
        depth->payload_bytes = (2 * (int64_t)GCE_STATS_WORDS + 2 * nb + 2 * (int64_t)depth->n_regions) * 8;
    }
    auto engine_payload = [&](gce_engine *x) -> int {
        if (!depth) return GCE_OK;
        const int64_t *pay = nullptr; gce_payload_layout lay;
        return gce_stats_payload_device(x, coverage_step, depth->n_regions, depth->region_tid, depth->region_start, depth->region_end, &pay, &lay);
    };
    if (n_shards > 1) {
        // ---- several engines: each indexes the stream it received, keeps its shard, runs it and assembles its records -- side by side, one host
        //      thread per engine, nothing exchanged; then the streams are merged on the first engine's device
        std::vector<gce_engine *> all; all.push_back(e); for (auto *x : mir) all.push_back(x);
        std::vector<int> rcs((size_t)n_shards, GCE_OK); std::vector<std::string> msgs((size_t)n_shards); std::vector<int64_t> nrec((size_t)n_shards, 0); std::vector<double> kms((size_t)n_shards, 0.0), t_idx((size_t)n_shards, 0.0);
        std::vector<std::thread> th;
        for (int32_t r = 0; r < n_shards; r++) th.emplace_back([&, r] {
            gce_engine *x = all[(size_t)r]; int c2; int64_t n0 = 0; uint64_t b2 = 0; int64_t o2 = 0;
            auto failr = [&](int code) { rcs[(size_t)r] = code; const char *m = gce_last_error(x); msgs[(size_t)r] = m && m[0] ? m : gce_status_message(code); };
            const double a0 = now_s();
            if ((c2 = gce_raw_finish(x, hdr_end, prm.n_targets, &n0)) != GCE_OK) return failr(c2);
            if (n0 > 0 && (c2 = gce_raw_select_shard(x, n_shards, r, plan_mode)) != GCE_OK) return failr(c2);
            t_idx[(size_t)r] = now_s() - a0;
            gce_result rs;
            if (n0 > 0 && ((c2 = gce_process(x)) != GCE_OK || (c2 = gce_result_device(x, &rs)) != GCE_OK || (c2 = gce_raw_build_output(x, &b2, &o2)) != GCE_OK || (c2 = engine_payload(x)) != GCE_OK)) return failr(c2);
            gce_timing tm; if (n0 > 0 && gce_get_timing(x, &tm) == GCE_OK) kms[(size_t)r] = tm.total_ms;
            nrec[(size_t)r] = n0;
        });
        for (auto &t : th) t.join();
        for (int32_t r = 0; r < n_shards; r++) if (rcs[(size_t)r] != GCE_OK) return done(rcs[(size_t)r], msgs[(size_t)r].c_str());
        for (int32_t r = 0; r < n_shards; r++) { out->kernel_ms = std::max(out->kernel_ms, kms[(size_t)r]); out->index_s = std::max(out->index_s, t_idx[(size_t)r]); }
        out->process_s = now_s() - t0 - out->index_s; t0 = now_s();
        n_rec = nrec[0];
        if (n_rec > 0) {
            if ((rc = gce_raw_merge_outputs(all.data(), n_shards, &body, &n_out, &out->pre, &out->post, &out->n_reads)) != GCE_OK) return done(rc, gce_last_error(e));
            out->n_out = n_out;
        }
        out->drain_s = now_s() - t0; t0 = now_s();
    } else {
    if ((rc = gce_raw_finish(e, hdr_end, prm.n_targets, &n_rec)) != GCE_OK) return done(rc, gce_last_error(e));
    out->index_s = now_s() - t0; t0 = now_s();
    if (n_rec > 0) {
        if ((rc = gce_process(e)) != GCE_OK) return done(rc, gce_last_error(e)[0] ? gce_last_error(e) : gce_status_message(rc));
        out->process_s = now_s() - t0; t0 = now_s();
        gce_timing tm; if (gce_get_timing(e, &tm) == GCE_OK) out->kernel_ms = tm.total_ms;
        gce_result res;
        if ((rc = gce_result_device(e, &res)) != GCE_OK) return done(rc, gce_last_error(e));
        out->n_reads = res.n_reads; out->n_out = res.n_out; out->pre = res.pre; out->post = res.post;
        if ((rc = gce_raw_build_output(e, &body, &n_out)) != GCE_OK) return done(rc, gce_last_error(e));
        if ((rc = engine_payload(e)) != GCE_OK) return done(rc, gce_last_error(e));
        out->drain_s = now_s() - t0; t0 = now_s();
        if (getenv("GCE_RAW_TIMING")) fprintf(stderr, "gce_run_bam: RSS after process + output records %ld MB\n", status_kb("VmRSS:") >> 10);
    }
    }   // (one engine)
    // ---- the depth statistics of the report: the engines' payloads (computed above, each on its own Stats blocks) summed in device memory, to the host once
    if (depth && n_rec > 0) {
        std::vector<gce_engine *> all; all.push_back(e); for (auto *x : mir) all.push_back(x);
        const int64_t *pay = nullptr; gce_payload_layout lay;
        if ((rc = gce_stats_payload_sum(all.data(), (int32_t)all.size(), &pay, &lay)) != GCE_OK) return done(rc, gce_last_error(e));
        std::vector<int64_t> host((size_t)lay.total_words);
        if ((rc = gce_stats_payload_read(e, pay, lay.total_words, host.data())) != GCE_OK) return done(rc, gce_last_error(e));
        const int64_t nb = depth->n_bins; const int32_t nreg = depth->n_regions;
        if (lay.n_bins != nb || lay.n_regions != nreg) return done(GCE_ERR_INVALID, "payload layout");
        memcpy(&depth->pre, host.data(), sizeof(gce_stats)); memcpy(&depth->post, host.data() + GCE_STATS_WORDS, sizeof(gce_stats));
        const int64_t *d0 = host.data() + lay.stats_words;
        memcpy(depth->pre_depth, d0, (size_t)nb * 8); memcpy(depth->post_depth, d0 + nb, (size_t)nb * 8);
        memcpy(depth->pre_bed, d0 + 2 * nb, (size_t)nreg * 8); memcpy(depth->post_bed, d0 + 2 * nb + nreg, (size_t)nreg * 8);
    }
    // ---- the output file: header bytes + the record stream from HBM, in pieces; deflate by all threads, written in order
    const size_t opl = strlen(out_path);
    if (opl >= 3 && strcmp(out_path + opl - 3, "sam") == 0) {
        // an output name that ends in "sam" is written as SAM text (src/gencore.cpp:170-173: sam_open(out, "w")): the header text (with @SQ
        // lines from the contig table if it has none), then the record stream piece by piece, records -> lines on all host threads
        fo = fopen(out_path, "w");
        if (!fo) return done(GCE_ERR_INVALID, "cannot open the output SAM");
        const std::string ht = samtext::header_text_for_sam(text, names, lens);
        if (fwrite(ht.data(), 1, ht.size(), fo) != ht.size()) return done(GCE_ERR_INVALID, "cannot write the output SAM");
        const uint64_t OC = PIECE < ((size_t)8 << 20) ? ((uint64_t)64 << 10) : ((uint64_t)16 << 20);       // (tests: pieces that cut records)
        const int64_t npieces = (int64_t)((body + OC - 1) / OC);
        Pinned obuf[2]; int32_t otk[2] = {-1, -1};
        auto fetch = [&](int64_t pc) -> int { const uint64_t a2 = (uint64_t)pc * OC, z2 = std::min<uint64_t>(body, a2 + OC); if (!obuf[pc & 1].ensure((size_t)(z2 - a2) + 64)) return GCE_ERR_OOM; return gce_raw_read_output_async(e, a2, obuf[pc & 1].p, (size_t)(z2 - a2), &otk[pc & 1]); };
        std::vector<uint8_t> cur; std::vector<uint64_t> ro; std::vector<std::string> lines((size_t)T);
        if (npieces > 0 && (rc = fetch(0)) != GCE_OK) return done(rc, "output piece");
        for (int64_t pc = 0; pc < npieces; pc++) {
            if (otk[pc & 1] >= 0 && (rc = gce_submit_wait(e, otk[pc & 1])) != GCE_OK) return done(rc, gce_last_error(e));
            const uint64_t a2 = (uint64_t)pc * OC, z2 = std::min<uint64_t>(body, a2 + OC);
            cur.insert(cur.end(), obuf[pc & 1].p, obuf[pc & 1].p + (z2 - a2));                // behind the record a piece border cut
            if (pc + 1 < npieces && (rc = fetch(pc + 1)) != GCE_OK) return done(rc, "output piece");
            ro.clear();
            uint64_t o = 0;
            while (o + 4 <= cur.size()) { const uint32_t bs = rd32(cur.data() + o); if (bs < 32) return done(GCE_ERR_INVALID, "bad record in the output stream"); if (o + 4 + bs > cur.size()) break; ro.push_back(o); o += 4ull + bs; }
            std::atomic<int> bad{0};
            parallel_for(T, T, [&](int, int64_t x, int64_t y) { for (int64_t t = x; t < y; t++) { std::string &L = lines[(size_t)t]; L.clear(); const size_t ra = ro.size() * (size_t)t / (size_t)T, rb = ro.size() * (size_t)(t + 1) / (size_t)T; for (size_t q = ra; q < rb; q++) if (!samtext::bam_to_line(cur.data() + ro[q], names, L)) { bad = 1; return; } } });
            if (bad) return done(GCE_ERR_INVALID, "bad record in the output stream");
            for (int t = 0; t < T; t++) if (!lines[(size_t)t].empty() && fwrite(lines[(size_t)t].data(), 1, lines[(size_t)t].size(), fo) != lines[(size_t)t].size()) return done(GCE_ERR_INVALID, "cannot write the output SAM");
            cur.erase(cur.begin(), cur.begin() + (ptrdiff_t)o);
        }
        if (!cur.empty()) return done(GCE_ERR_INVALID, "truncated record at the end of the output stream");
        const bool closed = fclose(fo) == 0; fo = nullptr;
        if (!closed) return done(GCE_ERR_INVALID, "cannot write the output SAM");
        out->write_s = now_s() - t0;
        out->total_s = now_s() - t_start;
        out->peak_rss_kb = status_kb("VmHWM:"); out->rss_end_kb = status_kb("VmRSS:");
        return done(GCE_OK, "");
    }
    const std::vector<uint8_t> hdr = bam_header_bytes();
    fo = fopen(out_path, "wb");
    if (!fo) return done(GCE_ERR_INVALID, "cannot open the output BAM");
    const uint64_t BS = 0xff00, OC = BS * 256, total = hdr.size() + body;
    const int64_t npieces = (int64_t)((total + OC - 1) / OC);
    Pinned obuf[3]; int32_t otk[3] = {-1, -1, -1};
    Raw<uint8_t> zbuf; zbuf.resize((size_t)256 * 0x10000 + 64);
    if (!zbuf.ok()) return done(GCE_ERR_OOM, "out of host memory");
    if (level == -2) {
        // level -2: the record stream is deflated BY THE GPU (gce_deflate.hpp: fixed Huffman codes, one lane per BGZF block) -- the host compresses
        // the header's few blocks, then only copies the file image out of HBM piece by piece and writes it
        for (uint64_t o = 0; o < hdr.size(); o += BS) {
            const uint32_t zs = (uint32_t)deflate_block(hdr.data() + o, (uint32_t)std::min<uint64_t>(BS, hdr.size() - o), 1, zbuf.data());
            if (zs == 0 || fwrite(zbuf.data(), 1, zs, fo) != zs) return done(GCE_ERR_INVALID, "cannot write the output BAM");
        }
        uint64_t cb = 0;
        if (body && (rc = gce_raw_deflate_output(e, &cb)) != GCE_OK) return done(rc, gce_last_error(e));
        const uint64_t PC = (uint64_t)16 << 20; const int64_t np2 = (int64_t)((cb + PC - 1) / PC);
        auto fetch2 = [&](int64_t pc) -> int { const uint64_t a2 = (uint64_t)pc * PC, z2 = std::min<uint64_t>(cb, a2 + PC); if (!obuf[pc & 1].ensure((size_t)(z2 - a2) + 64)) return GCE_ERR_OOM; return gce_raw_read_deflated_async(e, a2, obuf[pc & 1].p, (size_t)(z2 - a2), &otk[pc & 1]); };
        if (np2 > 0 && (rc = fetch2(0)) != GCE_OK) return done(rc, "output piece");
        for (int64_t pc = 0; pc < np2; pc++) {
            if ((rc = gce_submit_wait(e, otk[pc & 1])) != GCE_OK) return done(rc, gce_last_error(e));
            if (pc + 1 < np2 && (rc = fetch2(pc + 1)) != GCE_OK) return done(rc, "output piece");
            const uint64_t a2 = (uint64_t)pc * PC, z2 = std::min<uint64_t>(cb, a2 + PC);
            if (fwrite(obuf[pc & 1].p, 1, (size_t)(z2 - a2), fo) != (size_t)(z2 - a2)) return done(GCE_ERR_INVALID, "cannot write the output BAM");
        }
        static const uint8_t eof2[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bool eof_ok2 = fwrite(eof2, 1, 28, fo) == 28;
        const bool closed2 = fclose(fo) == 0; fo = nullptr;
        if (!eof_ok2 || !closed2) return done(GCE_ERR_INVALID, "cannot write the output BAM");
        out->write_s = now_s() - t0;
        out->total_s = now_s() - t_start;
        out->peak_rss_kb = status_kb("VmHWM:"); out->rss_end_kb = status_kb("VmRSS:");
        return done(GCE_OK, "");
    }
    auto fetch = [&](int64_t pc) -> int {                                                 // piece pc of (header ++ body) into obuf[pc % 3]
        const uint64_t a = (uint64_t)pc * OC, z2 = std::min<uint64_t>(total, a + OC);
        Pinned &b = obuf[pc % 3];
        if (!b.ensure((size_t)(z2 - a) + 64)) return GCE_ERR_OOM;
        uint64_t at = a;
        if (at < hdr.size()) { const uint64_t hn = std::min<uint64_t>(hdr.size(), z2) - at; memcpy(b.p, hdr.data() + at, hn); at += hn; }
        if (at < z2) return gce_raw_read_output_async(e, at - hdr.size(), b.p + (at - a), (size_t)(z2 - at), &otk[pc % 3]);
        otk[pc % 3] = -1; return GCE_OK;
    };
    if (npieces > 0 && (rc = fetch(0)) != GCE_OK) return done(rc, "output piece");
    for (int64_t pc = 0; pc < npieces; pc++) {
        if (pc + 1 < npieces && (rc = fetch(pc + 1)) != GCE_OK) return done(rc, "output piece");
        if (otk[pc % 3] >= 0 && (rc = gce_submit_wait(e, otk[pc % 3])) != GCE_OK) return done(rc, gce_last_error(e));
        const uint64_t a = (uint64_t)pc * OC, z2 = std::min<uint64_t>(total, a + OC);
        const int64_t nb = (int64_t)((z2 - a + BS - 1) / BS);
        std::vector<uint32_t> zs((size_t)nb, 0);
        const uint8_t *src = obuf[pc % 3].p;
        parallel_for(T, nb, [&](int, int64_t x, int64_t y) { for (int64_t q = x; q < y; q++) { const uint64_t o = (uint64_t)q * BS; zs[q] = (uint32_t)deflate_block(src + o, (uint32_t)std::min<uint64_t>(BS, z2 - a - o), level, zbuf.data() + (size_t)q * 0x10000); } });
        for (int64_t q = 0; q < nb; q++) { if (zs[q] == 0 || fwrite(zbuf.data() + (size_t)q * 0x10000, 1, zs[q], fo) != zs[q]) return done(GCE_ERR_INVALID, "cannot write the output BAM"); }
    }
    static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool eof_ok = fwrite(eof_block, 1, 28, fo) == 28;
    const bool closed = fclose(fo) == 0; fo = nullptr;
    if (!eof_ok || !closed) return done(GCE_ERR_INVALID, "cannot write the output BAM");
    out->write_s = now_s() - t0;
    out->total_s = now_s() - t_start;
    out->peak_rss_kb = status_kb("VmHWM:"); out->rss_end_kb = status_kb("VmRSS:");
    (void)pushed;
    return done(GCE_OK, "");
}

// Gencore::consensus() for one BAM over SEVERAL engines: the stream is cut into n_shards ranges of the cluster key by the
// GPU planner (gce_stream_context + gce_plan_shards on devices[0]); shard r runs on HIP device devices[r] (ordinals may repeat: several
// engines on one GPU, which is how a one-GPU box tests it) with its reads, their global ticks, the flush events of the whole stream and
// -- per-shard staging -- only the reference window its reads can touch; the engines run side by side on host threads.  No data-path
// exchange: the output tables (each in bamComp order) are merged k-way by (tid, pos, mtid, mpos, isize, input index), mate rows are
// re-pointed, the two Stats blocks are SUMMED ON THE HOST (one process owns all engines here; one process per GPU merges them with one
// RCCL all-reduce instead, bench.py).  The result equals gce_run_bam's: same records, same order, same Stats.
int gce_run_bam_sharded_hostcodec(const char *in_path, const char *out_path, const char *fasta_path, const gce_params *params, int32_t n_shards, const int32_t *devices,
                                  int32_t plan_mode, int threads, int level, gce_bam_run *out, char err[256]) {
    auto seterr = [&](const char *m) { if (err) { strncpy(err, m ? m : "", 255); err[255] = 0; } };
    seterr("");
    if (!in_path || !out_path || !params || !out || n_shards < 1 || n_shards > 64 || !devices) return GCE_ERR_INVALID;
    memset(out, 0, sizeof *out);
    const double t_start = now_s();
    gce_bam *f = nullptr; gce_fasta *fa = nullptr;
    std::vector<gce_engine *> eng((size_t)n_shards, nullptr);
    int32_t *ev_tid = nullptr, *ev_pos = nullptr;
    int rc = gce_bam_open(in_path, threads, &f);
    auto done = [&](int code, const char *m) { seterr(m); for (auto *e : eng) if (e) gce_destroy(e); if (f) gce_bam_close(f); if (fa) gce_fasta_free(fa); gce_free(ev_tid); gce_free(ev_pos); return code; };
    if (rc != GCE_OK) return done(rc, f ? gce_bam_error(f) : "open failed");
    out->open_s = now_s() - t_start;
    gce_bam_info bi; gce_bam_get_info(f, &bi);
    out->read_s = bi.read_s; out->inflate_s = bi.inflate_s; out->index_s = bi.index_s;
    gce_params prm = *params;
    prm.n_targets = bi.n_targets; prm.target_len = bi.target_len; prm.tick_offset = 0; prm.trailing_flush = 0;
    if (strcmp(prm.umi_prefix, "auto") == 0) {                                   // src/gencore.cpp:207-220
        memset(prm.umi_prefix, 0, sizeof prm.umi_prefix);
        if (bi.n_records > 0) { gce_batch one; if (gce_bam_chunk(f, 0, 1, 0, &one) == GCE_OK) gce_detect_umi_prefix(one.qname, prm.umi_prefix); }
    }
    const int64_t n = bi.n_records;
    const int T = f->threads;
    const uint8_t *u = f->u.data();
    double t0 = now_s();
    // ---- the key records of the whole stream, the plan
    Raw<gce_core> cores; cores.resize((size_t)std::max<int64_t>(n, 1));
    Raw<uint64_t> tick; tick.resize((size_t)std::max<int64_t>(n, 1));
    Raw<int32_t> shard; shard.resize((size_t)std::max<int64_t>(n, 1));
    if (!cores.ok() || !tick.ok() || !shard.ok()) return done(GCE_ERR_OOM, "out of host memory");
    parallel_for(T, n, [&](int, int64_t a, int64_t e) { for (int64_t k = a; k < e; k++) memcpy(&cores[k], u + f->rec[k] + 4, 32); });
    int32_t n_ev = 0;
    const int period = prm.flush_period > 0 ? prm.flush_period : 10000;
    // --quit_after_contig (gencore.cpp:243-246): ONE cut on the whole stream, in front of the plan; the cut read goes to shard 0 behind its own reads (that
    // engine counts it and drops it), the other engines do not look for a cut (see gce_raw_select_shard)
    int64_t n_plan = n, cut = -1;
    if (prm.max_contig > 0) for (int64_t k = 0; k < n; k++) if (cores[k].tid >= prm.max_contig) { cut = k; n_plan = k; break; }
    if (n_plan > 0) {
        if ((rc = gce_stream_context(devices[0], cores.data(), n_plan, period, tick.data(), &n_ev, &ev_tid, &ev_pos)) != GCE_OK)
            return done(rc, rc == GCE_ERR_INVALID ? "not shardable by cluster key: a mapped read follows the first unmapped read" : gce_status_message(rc));
        if ((rc = gce_plan_shards(devices[0], cores.data(), n_plan, n_shards, plan_mode, shard.data())) != GCE_OK) return done(rc, gce_status_message(rc));
    }
    std::vector<std::vector<int64_t>> idx((size_t)n_shards);
    { std::vector<int64_t> cnt((size_t)n_shards, 0); for (int64_t k = 0; k < n_plan; k++) cnt[shard[k]]++; for (int r = 0; r < n_shards; r++) idx[r].reserve((size_t)cnt[r] + 1); for (int64_t k = 0; k < n_plan; k++) idx[shard[k]].push_back(k); }
    if (cut >= 0) { idx[0].push_back(cut); tick[cut] = 0; }
    if (fasta_path && *fasta_path && (rc = gce_fasta_load(fasta_path, threads, &fa)) != GCE_OK) return done(rc, "cannot read the FASTA file");
    int32_t nc = 0; const char *const *ids = nullptr; const char *const *seqs = nullptr; const int64_t *lens = nullptr;
    if (fa) gce_fasta_get(fa, &nc, &ids, &seqs, &lens);
    std::vector<int32_t> fa_of((size_t)bi.n_targets, -1);                        // Reference::getData looks contigs up by BAM target name (reference.cpp:43-53)
    for (int32_t t = 0; t < bi.n_targets; t++) for (int32_t c = 0; c < nc; c++) if (strcmp(ids[c], bi.target_name[t]) == 0) fa_of[t] = c;
    out->submit_s = now_s() - t0; t0 = now_s();
    // ---- the engines, side by side
    std::vector<gce_result> res((size_t)n_shards);
    std::vector<int> rcs((size_t)n_shards, GCE_OK); std::vector<std::string> msgs((size_t)n_shards);
    std::vector<Slot> slots((size_t)n_shards);
    std::vector<Raw<uint64_t>> ticks((size_t)n_shards);
    std::vector<double> kms((size_t)n_shards, 0.0);
    std::vector<std::thread> th;
    const int Tsub = std::max(1, T / n_shards);
    for (int r = 0; r < n_shards; r++) th.emplace_back([&, r] {
        auto failr = [&](int code, const char *m) { rcs[r] = code; msgs[r] = m ? m : ""; };
        const int64_t cnt = (int64_t)idx[r].size();
        memset(&res[r], 0, sizeof res[r]);
        gce_params pr = prm; pr.device = devices[r];
        if (r != 0 || cut < 0) pr.max_contig = 0;                                // (the cut is made above, once)
        int c2;
        if ((c2 = gce_create(&pr, &eng[r])) != GCE_OK) return failr(c2, gce_status_message(c2));
        if (cnt == 0) return;
        gce_batch b;
        {   // (gce_bam's own thread count is shared: the chunker of a shard uses its share of the host threads)
            const int keep = f->threads; (void)keep;
            if ((c2 = bam_chunk_impl(f, 0, idx[r].data(), cnt, slots[r], &b)) != GCE_OK) return failr(c2, "chunk");
        }
        ticks[r].resize((size_t)cnt);
        if (!ticks[r].ok()) return failr(GCE_ERR_OOM, "out of host memory");
        for (int64_t k = 0; k < cnt; k++) ticks[r][k] = tick[idx[r][k]];
        b.tick = ticks[r].data();
        if (fa) {                                                                // per-shard staging: per contig the bases between the shard's first read and its last reference position
            std::vector<int64_t> lo((size_t)bi.n_targets, INT64_MAX), hi((size_t)bi.n_targets, -1);
            for (int64_t k = 0; k < cnt; k++) {
                const gce_core &c = b.core[k];
                if (c.tid < 0 || c.tid >= bi.n_targets || c.pos < 0) continue;
                int64_t rl = 0; const uint32_t *cg = b.cigar + b.cigar_off[k];
                for (uint32_t q = 0; q < c.n_cigar; q++) { const uint32_t op = cg[q] & 0xF; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += cg[q] >> 4; }
                lo[c.tid] = std::min<int64_t>(lo[c.tid], c.pos); hi[c.tid] = std::max<int64_t>(hi[c.tid], (int64_t)c.pos + rl);
            }
            for (int32_t t = 0; t < bi.n_targets; t++) {
                if (fa_of[t] < 0 || hi[t] < 0) continue;
                const int64_t clen = lens[fa_of[t]];
                const int64_t a = std::min<int64_t>(lo[t] & ~(int64_t)1, clen & ~(int64_t)1), z = std::min<int64_t>(hi[t], clen);
                if ((c2 = gce_set_reference_window(eng[r], t, clen, a, seqs[fa_of[t]] + a, std::max<int64_t>(z - a, 0))) != GCE_OK) return failr(c2, gce_last_error(eng[r]));
            }
        }
        if ((c2 = gce_set_flush_events(eng[r], n_ev, ev_tid, ev_pos)) != GCE_OK || (c2 = gce_submit(eng[r], &b)) != GCE_OK) return failr(c2, gce_last_error(eng[r]));
        if ((c2 = gce_process(eng[r])) != GCE_OK) return failr(c2, gce_last_error(eng[r])[0] ? gce_last_error(eng[r]) : gce_status_message(c2));
        gce_timing tm; if (gce_get_timing(eng[r], &tm) == GCE_OK) kms[r] = tm.total_ms;
        if ((c2 = gce_drain(eng[r], &res[r])) != GCE_OK) return failr(c2, gce_last_error(eng[r]));
    });
    for (auto &x : th) x.join();
    (void)Tsub;
    for (int r = 0; r < n_shards; r++) if (rcs[r] != GCE_OK) return done(rcs[r], msgs[r].c_str());
    out->process_s = now_s() - t0; t0 = now_s();
    for (int r = 0; r < n_shards; r++) out->kernel_ms = std::max(out->kernel_ms, kms[r]);
    // ---- merge: k-way by bamComp over the global input index
    int64_t n_out = 0;
    for (int r = 0; r < n_shards; r++) n_out += res[r].n_out;
    std::vector<uint32_t> m_src((size_t)n_out), m_qsrc((size_t)n_out); std::vector<int32_t> m_nm((size_t)n_out); std::vector<int16_t> m_fr((size_t)n_out), m_rr((size_t)n_out);
    std::vector<const uint8_t *> m_seq((size_t)n_out), m_qual((size_t)n_out);
    std::vector<std::vector<uint32_t>> rowmap((size_t)n_shards);
    std::vector<int64_t> head((size_t)n_shards, 0);
    for (int r = 0; r < n_shards; r++) rowmap[r].resize((size_t)res[r].n_out);
    auto gsrc = [&](int r, int64_t k) { return (uint32_t)idx[r][res[r].src[k]]; };
    auto less = [&](int ra, int64_t ka, int rb, int64_t kb) {
        const uint32_t ia = gsrc(ra, ka), ib = gsrc(rb, kb);
        const gce_core &a = cores[ia], &b = cores[ib];
        if (a.tid != b.tid) return a.tid < b.tid;
        if (a.pos != b.pos) return a.pos < b.pos;
        if (a.mtid != b.mtid) return a.mtid < b.mtid;
        if (a.mpos != b.mpos) return a.mpos < b.mpos;
        if (a.isize != b.isize) return a.isize < b.isize;
        return ia < ib;
    };
    for (int64_t row = 0; row < n_out; row++) {
        int best = -1;
        for (int r = 0; r < n_shards; r++) if (head[r] < res[r].n_out && (best < 0 || less(r, head[r], best, head[best]))) best = r;
        const int64_t k = head[best]++;
        rowmap[best][k] = (uint32_t)row;
        m_src[row] = gsrc(best, k); m_qsrc[row] = (uint32_t)idx[best][res[best].qname_src[k]];
        m_nm[row] = res[best].nm_new[k]; m_fr[row] = res[best].fr[k]; m_rr[row] = res[best].rr[k];
        m_seq[row] = res[best].seq + res[best].seq_off[k]; m_qual[row] = res[best].qual + res[best].qual_off[k];
    }
    // (the unsigned tid of an unmapped record sorts it last in the engines' tables; bamComp compares the signed field: pass-through
    //  records of unmapped reads do not exist -- unmapped reads are dropped, gencore.cpp:255-266 -- so the two orders agree)
    for (int r = 0; r < n_shards; r++) {
        out->n_reads += res[r].n_reads; out->n_out += res[r].n_out;
        const int64_t *a = (const int64_t *)&res[r].pre, *c = (const int64_t *)&res[r].post;
        int64_t *pa = (int64_t *)&out->pre, *pc = (int64_t *)&out->post;
        for (int q = 0; q < GCE_STATS_WORDS; q++) { pa[q] += a[q]; pc[q] += c[q]; }
    }
    out->drain_s = now_s() - t0; t0 = now_s();
    const OutRows rows{n_out, m_src.data(), m_qsrc.data(), m_nm.data(), m_fr.data(), m_rr.data()};
    if ((rc = bam_write_rows(out_path, f, rows, [&](int64_t k) { return m_seq[k]; }, [&](int64_t k) { return m_qual[k]; }, threads, level)) != GCE_OK) return done(rc, "cannot write the output BAM");
    out->write_s = now_s() - t0;
    out->total_s = now_s() - t_start;
    return done(GCE_OK, "");
}


int gce_run_bam(const char *in_path, const char *out_path, const char *fasta_path, const gce_params *params, int threads,
                int64_t chunk_reads, int level, gce_bam_run *out, char err[256]) {
    if (!in_path || !out_path || !params || !out) return GCE_ERR_INVALID;
    if (getenv("GCE_BAM_HOSTCODEC")) return gce_run_bam_hostcodec(in_path, out_path, fasta_path, params, threads, chunk_reads, level, out, err);
    return run_bam_impl(in_path, out_path, fasta_path, params, threads, chunk_reads, level, out, err, 1, nullptr, 0);
}

void gce_depth_run_free(gce_depth_run *d) {
    if (!d) return;
    free(d->bin_off); free(d->pre_depth); free(d->post_depth); free(d->pre_bed); free(d->post_bed);
    if (d->region_tid || d->region_start || d->region_end) gce_bed_free(d->n_regions, d->region_tid, d->region_start, d->region_end, nullptr);
    memset(d, 0, sizeof *d);
}

// gce_run_bam / gce_run_bam_sharded with the depth statistics of the reference's report: see include/gencore_amd.h.
int gce_run_bam_depth(const char *in_path, const char *out_path, const char *fasta_path, const char *bed_path, int32_t coverage_step, const gce_params *params,
                      int32_t n_shards, const int32_t *devices, int32_t plan_mode, int threads, int level, gce_bam_run *out, gce_depth_run *depth, char err[256]) {
    if (!depth || n_shards < 1 || n_shards > 64 || (n_shards > 1 && !devices)) return GCE_ERR_INVALID;
    gce_params prm;
    if (params && n_shards == 1 && devices) { prm = *params; prm.device = devices[0]; params = &prm; }
    const int rc = run_bam_impl(in_path, out_path, fasta_path, params, threads, 0, level, out, err, n_shards, n_shards > 1 ? devices : nullptr, plan_mode, bed_path, coverage_step, depth);
    if (rc != GCE_OK) gce_depth_run_free(depth);
    return rc;
}

// Gencore::consensus() for one file over SEVERAL engines on the GPU codec (src/gencore.cpp:164-205): see run_bam_impl.  The host
// reads the file once; every engine gets the compressed pieces over its own PCIe link, inflates, indexes and plans on its own GPU and keeps its
// key range; outputs are merged device to device.  GCE_BAM_HOSTCODEC=1: round 2's runner (host inflate and index, host-side cut, host merge).
int gce_run_bam_sharded(const char *in_path, const char *out_path, const char *fasta_path, const gce_params *params, int32_t n_shards, const int32_t *devices,
                        int32_t plan_mode, int threads, int level, gce_bam_run *out, char err[256]) {
    if (!in_path || !out_path || !params || !out || n_shards < 1 || n_shards > 64 || !devices || (plan_mode != 0 && plan_mode != 1)) return GCE_ERR_INVALID;
    if (getenv("GCE_BAM_HOSTCODEC")) return gce_run_bam_sharded_hostcodec(in_path, out_path, fasta_path, params, n_shards, devices, plan_mode, threads, level, out, err);
    return run_bam_impl(in_path, out_path, fasta_path, params, threads, 0, level, out, err, n_shards, devices, plan_mode);
}

}  // extern "C"
