// Probe over the ONE reference file on the hot path that compiles without htslib: src/util.h (header-only).
// It is included from where it lies under /root/reference (nothing is copied) and its `split` -- the tokenizer behind
// Cluster::isDuplex (cluster.cpp:246-258) and Group's duplex check (group.cpp:55-66) -- is exposed through a C entry point,
// so tests can compare the oracle's restatement (orc_is_duplex) with the reference's own code.  Output: oracle/_ref/libref_util.so.
#include "util.h"
#include <cstring>

extern "C" {
// split(str, sep): returns the token count, writes the tokens joined by '\n' into out (truncated to cap-1 bytes)
int ref_split(const char *str, const char *sep, char *out, int cap) {
    std::vector<std::string> toks;
    split(std::string(str), toks, std::string(sep));
    std::string joined;
    for (size_t i = 0; i < toks.size(); i++) { if (i) joined += '\n'; joined += toks[i]; }
    if (cap > 0) { std::strncpy(out, joined.c_str(), cap - 1); out[cap - 1] = 0; }
    return (int)toks.size();
}
// the decision of Cluster::isDuplex expressed with the reference's split: two tokens each, crosswise equal
int ref_is_duplex(const char *umi1, const char *umi2) {
    std::vector<std::string> a, b;
    split(std::string(umi1), a, "_");
    split(std::string(umi2), b, "_");
    if (a.size() != 2 || b.size() != 2) return 0;
    return (a[0] == b[1] && a[1] == b[0]) ? 1 : 0;
}
}
