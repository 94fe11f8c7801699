// gce_samtext.hpp — SAM text <-> BAM records on the host (included by bamio.cpp; no GPU code).
//
// Replaces what the reference gets from htslib when the input or the output is SAM TEXT:
//   sam_open(in, "r") detects the format, sam_hdr_read / sam_read1 parse text lines into bam1_t   src/gencore.cpp:164,180,205
//   sam_open(out, "w") for an output name that ends in "sam", sam_hdr_write / sam_write1         src/gencore.cpp:170-173,187,104
// htslib is a pinned dependency that is absent from /root/reference; the format is the published one (SAMv1 sections 1.3-1.5 and 4.2),
// restated here with the two htslib conventions the consensus path can see:
//   * an integer tag (`NM:i:3`) is stored in the SMALLEST type that holds it -- 'C' for 0..255, which group.cpp:569 depends on
//     (`if(type == 'C')` patches NM in place) -- c/s/i for negative values, C/S/I otherwise;
//   * the bin is reg2bin(pos, pos + reference length of the CIGAR), reference length 1 for an unmapped read or an empty CIGAR.
// A line becomes exactly the bytes of the BAM record (block_size first) that the raw-stream path (gce_raw_push) takes, so a SAM file
// goes through the same GPU record index / parse / re-assembly as a BAM file; a BAM record becomes the line sam_format1 prints.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace samtext {

inline int reg2bin(int64_t beg, int64_t end) {             // SAMv1 5.3
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (int)(beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (int)(beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (int)(beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (int)(beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (int)(beg >> 26);
    return 0;
}

// @SQ lines of a header text -> contig names and lengths (in order of appearance = tid)
inline bool parse_header_text(const std::string &text, std::vector<std::string> &names, std::vector<uint32_t> &lens) {
    names.clear(); lens.clear();
    size_t a = 0;
    while (a < text.size()) {
        size_t e = text.find('\n', a); if (e == std::string::npos) e = text.size();
        if (e - a >= 3 && text.compare(a, 3, "@SQ") == 0) {
            std::string sn; long long ln = -1;
            size_t f = a;
            while (f < e) {
                size_t t = text.find('\t', f); if (t == std::string::npos || t > e) t = e;
                if (t - f > 3 && text.compare(f, 3, "SN:") == 0) sn.assign(text, f + 3, t - f - 3);
                else if (t - f > 3 && text.compare(f, 3, "LN:") == 0) ln = atoll(text.substr(f + 3, t - f - 3).c_str());
                f = t + 1;
            }
            if (sn.empty() || ln < 0 || ln > 0xFFFFFFFFll) return false;
            if (!sn.empty() && sn.back() == '\r') sn.pop_back();
            names.push_back(sn); lens.push_back((uint32_t)ln);
        }
        a = e + 1;
    }
    return true;
}

struct NameMap { std::unordered_map<std::string, int32_t> m; void build(const std::vector<std::string> &names) { m.clear(); for (size_t i = 0; i < names.size(); i++) m.emplace(names[i], (int32_t)i); } };

struct Nt16 { uint8_t t[256]; Nt16() { memset(t, 15, sizeof t); const char *codes = "=ACMGRSVTWYHKDBN"; for (int k = 0; k < 16; k++) { t[(uint8_t)codes[k]] = (uint8_t)k; if (codes[k] >= 'A') t[(uint8_t)(codes[k] + 32)] = (uint8_t)k; } } };
inline const uint8_t *nt16_table() { static const Nt16 n; return n.t; }      // IUPAC character -> 4-bit code ("=ACMGRSVTWYHKDBN", either case), anything else 15

template <class V> inline void put(V &v, const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; v.insert(v.end(), b, b + n); }
template <class V, class X> inline void put_le(V &v, X x) { put(v, &x, sizeof x); }

inline bool parse_int(const char *a, const char *e, long long &out) {
    if (a >= e) return false;
    bool neg = false; const char *p = a;
    if (*p == '-' || *p == '+') { neg = *p == '-'; p++; }
    if (p >= e) return false;
    long long v = 0;
    for (; p < e; p++) { if (*p < '0' || *p > '9') return false; v = v * 10 + (*p - '0'); if (v > (1ll << 40)) return false; }
    out = neg ? -v : v; return true;
}

// one alignment line [s, e) (no line feed) -> one BAM record appended to `out`; false + msg for a malformed line
template <class V> inline bool line_to_bam(const char *s, const char *e, const NameMap &nm, V &out, std::string &msg) {
    if (e > s && e[-1] == '\r') e--;
    const char *fb_[11], *fe_[11]; int nf = 0; const char *aux = s;
    fb_[0] = s;
    while (nf < 11) {
        const char *t = (const char *)memchr(aux, '\t', (size_t)(e - aux));
        if (!t) { fe_[nf++] = e; aux = e; break; }
        fe_[nf++] = t; aux = t + 1;
        if (nf < 11) fb_[nf] = aux;
    }
    if (nf < 11) { msg = "SAM line with fewer than 11 fields"; return false; }
    auto fb = [&](int k) { return fb_[k]; };
    auto fe = [&](int k) { return fe_[k]; };
    long long flag, pos, mapq, pnext, tlen;
    if (!parse_int(fb(1), fe(1), flag) || !parse_int(fb(3), fe(3), pos) || !parse_int(fb(4), fe(4), mapq) || !parse_int(fb(7), fe(7), pnext) || !parse_int(fb(8), fe(8), tlen)
        || flag < 0 || flag > 0xFFFF || mapq < 0 || mapq > 255 || pos < 0 || pos > 0x7FFFFFFFll || pnext < 0 || pnext > 0x7FFFFFFFll || tlen < -0x7FFFFFFFll || tlen > 0x7FFFFFFFll) { msg = "SAM line with a bad numeric field"; return false; }
    const size_t lq = (size_t)(fe(0) - fb(0));
    if (lq < 1 || lq > 254) { msg = "SAM line with a bad QNAME"; return false; }
    auto lookup = [&](const char *a, const char *z) -> int32_t { if (z - a == 1 && *a == '*') return -1; auto it = nm.m.find(std::string(a, z)); return it == nm.m.end() ? -1 : it->second; };   // (an unknown name: unmapped, as htslib treats it)
    int32_t tid = lookup(fb(2), fe(2));
    // htslib's sam_parse1: "mapped query cannot have zero coordinate; treated as unmapped" (tid = -1), and a read without a contig carries BAM_FUNMAP
    if (pos == 0 && tid >= 0) tid = -1;
    if (tid < 0) flag |= 4;
    const int32_t mtid = (fe(6) - fb(6) == 1 && *fb(6) == '=') ? tid : lookup(fb(6), fe(6));      // (RNEXT is parsed behind POS there: '=' copies the tid as it stands AFTER that reset)
    // CIGAR
    const size_t base = out.size();
    uint32_t zero = 0; put_le(out, zero);                                              // block_size, patched at the end
    uint8_t core[32]; memset(core, 0, sizeof core); put(out, core, 32);
    put(out, fb(0), lq); out.push_back(0);
    uint32_t n_cigar = 0; int64_t rlen = 0, qlen = 0;
    if (fe(5) - fb(5) == 1 && *fb(5) == '*') flag |= 4;                                  // sam_parse1: "mapped query must have a CIGAR; treated as unmapped" (the flag only: contig and position stay)
    else {
        const char *p = fb(5), *z = fe(5);
        while (p < z) {
            uint64_t len = 0; const char *d = p;
            while (p < z && *p >= '0' && *p <= '9') { len = len * 10 + (uint64_t)(*p - '0'); p++; if (len >= (1ull << 28)) { msg = "CIGAR length out of range"; return false; } }
            if (p == d || p >= z) { msg = "malformed CIGAR"; return false; }
            const char *ops = "MIDNSHP=X"; const char *o = strchr(ops, *p);
            if (!o || !*p) { msg = "unknown CIGAR operation"; return false; }
            const uint32_t op = (uint32_t)(o - ops);
            put_le(out, (uint32_t)(len << 4 | op));
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += (int64_t)len;
            if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qlen += (int64_t)len;
            n_cigar++; p++;
            if (n_cigar > 65535) { msg = "more than 65535 CIGAR operations"; return false; }
        }
    }
    // SEQ / QUAL
    size_t lseq = 0;
    if (!(fe(9) - fb(9) == 1 && *fb(9) == '*')) {
        lseq = (size_t)(fe(9) - fb(9));
        const uint8_t *t16 = nt16_table(); const char *q = fb(9);
        const size_t o0 = out.size(); out.resize(o0 + (lseq + 1) / 2);                      // (written in place: a push_back per base was most of the parse)
        uint8_t *wp = &out[o0];
        for (size_t k = 0; k + 1 < lseq; k += 2) *wp++ = (uint8_t)(t16[(uint8_t)q[k]] << 4 | t16[(uint8_t)q[k + 1]]);
        if (lseq & 1) *wp = (uint8_t)(t16[(uint8_t)q[lseq - 1]] << 4);
        if (n_cigar > 0 && qlen != (int64_t)lseq) { msg = "CIGAR and query sequence are of different length"; return false; }      // (sam_parse1's error: such a record would reach the vote, which walks bases by CIGAR)
    }
    {
        const char *q = fb(10), *z = fe(10);
        if (z - q == 1 && *q == '*') out.insert(out.end(), lseq, (uint8_t)0xFF);
        else {
            if ((size_t)(z - q) != lseq) { msg = "SEQ and QUAL of different length"; return false; }
            const size_t o0 = out.size(); out.resize(o0 + lseq);
            uint8_t *wp = lseq ? &out[o0] : nullptr;
            for (size_t k = 0; k < lseq; k++) wp[k] = (uint8_t)(q[k] - 33);
        }
    }
    // optional fields
    for (const char *p = aux; p < e;) {
        const char *z = (const char *)memchr(p, '\t', (size_t)(e - p)); if (!z) z = e;
        if (z - p < 5 || p[2] != ':' || p[4] != ':') { msg = "malformed optional field"; return false; }
        const char type = p[3]; const char *v = p + 5;
        out.push_back((uint8_t)p[0]); out.push_back((uint8_t)p[1]);
        switch (type) {
        case 'A': if (z - v != 1) { msg = "malformed A field"; return false; } out.push_back('A'); out.push_back((uint8_t)*v); break;
        case 'i': {
            long long x; if (!parse_int(v, z, x) || x < -(1ll << 31) || x > 0xFFFFFFFFll) { msg = "integer field out of range"; return false; }
            if (*v == '-') { if (x >= -128) { out.push_back('c'); put_le(out, (int8_t)x); } else if (x >= -32768) { out.push_back('s'); put_le(out, (int16_t)x); } else { out.push_back('i'); put_le(out, (int32_t)x); } }
            else { if (x <= 255) { out.push_back('C'); put_le(out, (uint8_t)x); } else if (x <= 65535) { out.push_back('S'); put_le(out, (uint16_t)x); } else { out.push_back('I'); put_le(out, (uint32_t)x); } }
            break;
        }
        case 'f': { out.push_back('f'); const float x = strtof(std::string(v, z).c_str(), nullptr); put_le(out, x); break; }
        case 'd': { out.push_back('d'); const double x = strtod(std::string(v, z).c_str(), nullptr); put_le(out, x); break; }
        case 'Z': case 'H': out.push_back((uint8_t)type); put(out, v, (size_t)(z - v)); out.push_back(0); break;
        case 'B': {
            if (z - v < 1) { msg = "malformed B field"; return false; }
            const char sub = *v; const char *ops = "cCsSiIf"; if (!strchr(ops, sub) || !sub) { msg = "unknown B subtype"; return false; }
            out.push_back('B'); out.push_back((uint8_t)sub);
            uint32_t cnt = 0; for (const char *q = v + 1; q < z; q++) cnt += *q == ',';
            put_le(out, cnt);
            const char *q = v + 1;
            while (q < z) {
                q++;                                                                    // the comma
                const char *n = (const char *)memchr(q, ',', (size_t)(z - q)); if (!n) n = z;
                if (sub == 'f') { const float x = strtof(std::string(q, n).c_str(), nullptr); put_le(out, x); }
                else {
                    long long x; if (!parse_int(q, n, x)) { msg = "malformed B value"; return false; }
                    const long long lo_ = sub == 'c' ? -128 : sub == 's' ? -32768 : sub == 'i' ? -(1ll << 31) : 0, hi_ = sub == 'c' ? 127 : sub == 'C' ? 255 : sub == 's' ? 32767 : sub == 'S' ? 65535 : sub == 'i' ? 0x7FFFFFFFll : 0xFFFFFFFFll;
                    if (x < lo_ || x > hi_) { msg = "B value out of its type's range"; return false; }
                    switch (sub) { case 'c': put_le(out, (int8_t)x); break; case 'C': put_le(out, (uint8_t)x); break; case 's': put_le(out, (int16_t)x); break;
                                   case 'S': put_le(out, (uint16_t)x); break; case 'i': put_le(out, (int32_t)x); break; default: put_le(out, (uint32_t)x); break; }
                }
                q = n;
            }
            break;
        }
        default: msg = "unknown optional field type"; return false;
        }
        p = z < e ? z + 1 : e;
    }
    // core block
    const int64_t p0 = pos - 1;
    int64_t span = (flag & 4) ? 1 : rlen; if (span == 0) span = 1;
    const uint16_t bin = (uint16_t)reg2bin(p0, p0 + span);                              // (POS 0 -> pos -1: the shifts are arithmetic, bin 4680, as htslib gives an unplaced read)
    uint8_t *c = &out[base + 4];
    auto w32 = [&](int o, uint32_t x) { memcpy(c + o, &x, 4); };
    auto w16 = [&](int o, uint16_t x) { memcpy(c + o, &x, 2); };
    w32(0, (uint32_t)tid); w32(4, (uint32_t)(int32_t)p0); c[8] = (uint8_t)(lq + 1); c[9] = (uint8_t)mapq; w16(10, bin); w16(12, (uint16_t)n_cigar); w16(14, (uint16_t)flag);      // (flag: with BAM_FUNMAP added where sam_parse1 adds it)
    w32(16, (uint32_t)lseq); w32(20, (uint32_t)mtid); w32(24, (uint32_t)(int32_t)(pnext - 1)); w32(28, (uint32_t)(int32_t)tlen);
    const uint32_t bs = (uint32_t)(out.size() - base - 4);
    memcpy(&out[base], &bs, 4);
    return true;
}

inline void put_num(std::string &o, long long x) { char b[24]; const int n = snprintf(b, sizeof b, "%lld", x); o.append(b, (size_t)n); }

// one BAM record (r points at block_size; the record is complete and sane: the caller checked) -> one SAM line with its line feed
inline bool bam_to_line(const uint8_t *r, const std::vector<std::string> &names, std::string &o) {
    uint32_t bs; memcpy(&bs, r, 4);
    const uint8_t *c = r + 4, *end = r + 4 + bs;
    auto r32 = [&](int k) { int32_t x; memcpy(&x, c + k, 4); return x; };
    auto r16 = [&](int k) { uint16_t x; memcpy(&x, c + k, 2); return x; };
    const int32_t tid = r32(0), pos = r32(4), lseq = r32(16), mtid = r32(20), mpos = r32(24), tlen = r32(28);
    const uint32_t lq = c[8], mapq = c[9], nc = r16(12), flag = r16(14);
    if (bs < 32 || lq < 1 || lseq < 0 || 32ull + lq + 4ull * nc + ((uint64_t)lseq + 1) / 2 + (uint64_t)lseq > bs) return false;
    const uint8_t *qn = c + 32, *cg = qn + lq, *sq = cg + 4 * nc, *ql = sq + (lseq + 1) / 2, *ax = ql + lseq;
    o.append((const char *)qn, strnlen((const char *)qn, lq)); o.push_back('\t');
    put_num(o, flag); o.push_back('\t');
    if (tid >= 0 && (size_t)tid < names.size()) o += names[(size_t)tid]; else o.push_back('*');
    o.push_back('\t'); put_num(o, (long long)pos + 1); o.push_back('\t'); put_num(o, mapq); o.push_back('\t');
    if (nc == 0) o.push_back('*');
    else for (uint32_t k = 0; k < nc; k++) { uint32_t w; memcpy(&w, cg + 4 * k, 4); put_num(o, w >> 4); o.push_back("MIDNSHP=X???????"[w & 15]); }
    o.push_back('\t');
    if (mtid < 0) o.push_back('*'); else if (mtid == tid) o.push_back('='); else if ((size_t)mtid < names.size()) o += names[(size_t)mtid]; else o.push_back('*');
    o.push_back('\t'); put_num(o, (long long)mpos + 1); o.push_back('\t'); put_num(o, tlen); o.push_back('\t');
    if (lseq == 0) o.push_back('*');
    else { const size_t o0 = o.size(); o.resize(o0 + (size_t)lseq); char *wp = &o[o0]; for (int32_t k = 0; k < lseq; k++) wp[k] = "=ACMGRSVTWYHKDBN"[(sq[k >> 1] >> ((~k & 1) << 2)) & 15]; }
    o.push_back('\t');
    if (lseq == 0 || ql[0] == 0xFF) o.push_back('*');
    else { const size_t o0 = o.size(); o.resize(o0 + (size_t)lseq); char *wp = &o[o0]; for (int32_t k = 0; k < lseq; k++) wp[k] = (char)(ql[k] + 33); }
    for (const uint8_t *p = ax; p + 3 <= end;) {
        const uint8_t type = p[2]; const uint8_t *v = p + 3;
        o.push_back('\t'); o.push_back((char)p[0]); o.push_back((char)p[1]); o.push_back(':');
        char b[64];
        auto need = [&](size_t n) { return v + n <= end; };
        switch (type) {
        case 'A': if (!need(1)) return false; o += "A:"; o.push_back((char)v[0]); p = v + 1; break;
        case 'c': if (!need(1)) return false; o += "i:"; put_num(o, (int8_t)v[0]); p = v + 1; break;
        case 'C': if (!need(1)) return false; o += "i:"; put_num(o, v[0]); p = v + 1; break;
        case 's': { if (!need(2)) return false; int16_t x; memcpy(&x, v, 2); o += "i:"; put_num(o, x); p = v + 2; break; }
        case 'S': { if (!need(2)) return false; uint16_t x; memcpy(&x, v, 2); o += "i:"; put_num(o, x); p = v + 2; break; }
        case 'i': { if (!need(4)) return false; int32_t x; memcpy(&x, v, 4); o += "i:"; put_num(o, x); p = v + 4; break; }
        case 'I': { if (!need(4)) return false; uint32_t x; memcpy(&x, v, 4); o += "i:"; put_num(o, x); p = v + 4; break; }
        case 'f': { if (!need(4)) return false; float x; memcpy(&x, v, 4); o.append(b, (size_t)snprintf(b, sizeof b, "f:%g", x)); p = v + 4; break; }
        case 'd': { if (!need(8)) return false; double x; memcpy(&x, v, 8); o.append(b, (size_t)snprintf(b, sizeof b, "d:%g", x)); p = v + 8; break; }
        case 'Z': case 'H': {
            const uint8_t *z = (const uint8_t *)memchr(v, 0, (size_t)(end - v)); if (!z) return false;
            o.push_back((char)type); o.push_back(':'); o.append((const char *)v, (size_t)(z - v)); p = z + 1; break;
        }
        case 'B': {
            if (!need(5)) return false;
            const uint8_t sub = v[0]; uint32_t cnt; memcpy(&cnt, v + 1, 4);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
            if (!es || v + 5 + (uint64_t)es * cnt > end) return false;
            o += "B:"; o.push_back((char)sub);
            const uint8_t *q = v + 5;
            for (uint32_t k = 0; k < cnt; k++, q += es) {
                o.push_back(',');
                switch (sub) {
                case 'c': put_num(o, (int8_t)q[0]); break; case 'C': put_num(o, q[0]); break;
                case 's': { int16_t x; memcpy(&x, q, 2); put_num(o, x); break; } case 'S': { uint16_t x; memcpy(&x, q, 2); put_num(o, x); break; }
                case 'i': { int32_t x; memcpy(&x, q, 4); put_num(o, x); break; } case 'I': { uint32_t x; memcpy(&x, q, 4); put_num(o, x); break; }
                default: { float x; memcpy(&x, q, 4); o.append(b, (size_t)snprintf(b, sizeof b, "%g", x)); break; }
                }
            }
            p = q; break;
        }
        default: return false;
        }
    }
    o.push_back('\n');
    return true;
}

// the header a SAM file starts with: the text as it is; @SQ lines made from the contig table when the text holds none (sam_hdr_write)
inline std::string header_text_for_sam(const std::string &text, const std::vector<std::string> &names, const std::vector<uint32_t> &lens) {
    std::string t = text;
    while (!t.empty() && t.back() == 0) t.pop_back();
    if (!t.empty() && t.back() != '\n') t.push_back('\n');
    if (t.find("@SQ\t") == std::string::npos) for (size_t k = 0; k < names.size(); k++) { t += "@SQ\tSN:" + names[k] + "\tLN:" + std::to_string(lens[k]) + "\n"; }
    return t;
}

}  // namespace samtext
