"""Temporary instrumentation (not for commit): per-phase clock64() deltas of k_pairing_fast, sampled 1/64 clusters,
printed by gce_process when GCE_PROF is set.  Apply, build, run bench with GCE_PROF=1, then `git checkout` the sources."""
p='/root/repo/gencore_amd/csrc/gce_kernels.hpp'
s=open(p).read()
s=s.replace("uint32_t *ev_read; int max_events;","uint32_t *ev_read; int max_events; unsigned long long *prof;")
K0 = s.index("void k_pairing_fast")
def ins(before, text):
    global s
    i = s.index(before, K0)
    s = s[:i] + text + s[i:]
ins("    const uint32_t start = w.cl_start[c], n = w.cl_n[c];\n    uint64_t entry = w.table[w.cl_slot[c]];\n    uint32_t mode = d_thr_mode((uint32_t)(entry >> 32), w.si, p);\n    if (mode == THR_NEVER) { if (lane == 0) { w.cl_npairs[c] = 0; w.cl_ngroups[c] = 0; w.cl_hasumi[c] = 0; } return; }\n    const int thr = mode == THR_PROPER ? p.proper_thr : p.unproper_thr;\n    bool defer = n > 64;", "    long long t_[10]; t_[0] = clock64();\n")
ins("    const bool act = lane < (int)n;\n    uint64_t nw[8];", "    t_[1] = clock64();\n")
ins("    const unsigned long long ACT = __ballot(act);", "    t_[2] = clock64();\n")
ins("    {   // exact verification of every hash match", "    t_[3] = clock64();\n")
ins("    // ---- pairs (cluster.cpp:260-273, pair.cpp:188-216): first read of a name = mLeft", "    t_[4] = clock64();\n")
ins("    const uint32_t pidx = __popcll(LT);", "    t_[5] = clock64();\n")
ins("    // ---- lanes now stand for pairs (qname order)", "    t_[6] = clock64();\n")
ins("    // ---- lay the pairs out group by group (qname order inside a group)", "    t_[7] = clock64();\n")
ins("    if (lane == 0) { const bool cross = d_key(b.core[w.members[start]], p).right < 0;", "    t_[8] = clock64();\n    if (w.prof && lane == 0 && (c & 63) == 5) { for (int k = 0; k < 8; k++) atomicAdd(&w.prof[k], (unsigned long long)(t_[k + 1] - t_[k])); atomicAdd(&w.prof[15], 1ull); }\n")
open(p,'w').write(s)
p='/root/repo/gencore_amd/csrc/engine.hip'
s=open(p).read()
s=s.replace("    w.max_events = (int)max_events;","""    w.max_events = (int)max_events;
    static unsigned long long *d_prof = nullptr;
    if (getenv("GCE_PROF")) { if (!d_prof) hipMalloc(&d_prof, 128); hipMemsetAsync(d_prof, 0, 128, e->stream); w.prof = d_prof; }""")
old="    e->timing.total_ms = el(EV_START, EV_FINISH);"
new="""    if (w.prof) { unsigned long long h[16]; hipMemcpy(h, w.prof, 128, hipMemcpyDeviceToHost);
        fprintf(stderr, "PROF items=%llu cycles/item:", h[15]); for (int k = 0; k < 8; k++) fprintf(stderr, " p%d=%.0f", k, h[k]/(double)h[15]); fprintf(stderr, "\\n"); }
    e->timing.total_ms = el(EV_START, EV_FINISH);"""
assert old in s
s=s.replace(old,new)
open(p,'w').write(s)
