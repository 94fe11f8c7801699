/* tools/stress_lifetimes.c — engine lifetimes in a plain C process, to look for the intermittent wrong word of the drained post-Stats block
 *  without Python in the address space:  stress_lifetimes <dump> <lifetimes>
 * Every lifetime: gce_create -> gce_set_reference_ascii -> gce_submit -> gce_process -> gce_drain into a freshly calloc'ed gce_result ->
 * compare both Stats blocks with the FIRST lifetime's -> on a difference drain the same engine AGAIN into another fresh struct (is the
 * engine's own copy wrong, or was the first struct written to behind our back?) -> gce_destroy.  Between lifetimes a few heap blocks of
 * the result struct's size class are allocated, zeroed and checked later: a stray decrement that lands in one of them is reported too.
 * The dump format is tests/cabi_driver.c's (tests/test_cabi_driver.py::dump). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gencore_amd.h"

#define MAGIC 0x3142414345434721ull
static void *rd(FILE *f, size_t bytes) { void *p = malloc(bytes + 64); if (!p || (bytes && fread(p, 1, bytes, f) != bytes)) { fprintf(stderr, "short read\n"); exit(2); } memset((char *)p + bytes, 0, 64); return p; }
static uint64_t rd64(FILE *f) { uint64_t v; if (fread(&v, 8, 1, f) != 1) exit(2); return v; }
#define NCANARY 8

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: stress_lifetimes <dump> <lifetimes>\n"); return 2; }
    FILE *f = fopen(argv[1], "rb"); if (!f || rd64(f) != MAGIC) return 2;
    const long lifetimes = atol(argv[2]);
    const uint64_t n = rd64(f), nt = rd64(f), qb = rd64(f), cw = rd64(f), sb = rd64(f), lb = rd64(f);
    int32_t opts[2]; char prefix[32];
    if (fread(opts, 4, 2, f) != 2 || fread(prefix, 1, 32, f) != 32) return 2;
    uint32_t *tl = rd(f, nt * 4);
    char **bases = calloc(nt, sizeof *bases); uint64_t *blen = calloc(nt, 8);
    for (uint64_t t = 0; t < nt; t++) { blen[t] = rd64(f); bases[t] = rd(f, blen[t]); }
    gce_core *core = rd(f, n * sizeof(gce_core));
    uint64_t *qo = rd(f, n * 8), *co = rd(f, n * 8), *so = rd(f, n * 8), *lo = rd(f, n * 8);
    int32_t *nm = rd(f, n * 4); uint8_t *nmt = rd(f, n);
    char *qname = rd(f, qb); uint32_t *cigar = rd(f, cw * 4); uint8_t *seq = rd(f, sb), *qual = rd(f, lb);
    fclose(f);
    gce_params prm; gce_params_default(&prm);
    prm.cluster_size_req = opts[0]; prm.flush_period = opts[1]; memcpy(prm.umi_prefix, prefix, 32); prm.n_targets = (int32_t)nt; prm.target_len = tl;
    gce_stats pre0, post0; int have0 = 0; long bad = 0, canary_hits = 0;
    void *canary[NCANARY] = {0};
    for (long it = 0; it < lifetimes; it++) {
        gce_engine *e = NULL; int rc;
        if ((rc = gce_create(&prm, &e)) != GCE_OK) { fprintf(stderr, "gce_create %d\n", rc); return 3; }
        for (uint64_t t = 0; t < nt; t++) if (blen[t] && (rc = gce_set_reference_ascii(e, (int32_t)t, bases[t], (int64_t)blen[t])) != GCE_OK) return 3;
        gce_batch b; memset(&b, 0, sizeof b);
        b.n_reads = (int64_t)n; b.core = core; b.qname_off = qo; b.qname = qname; b.cigar_off = co; b.cigar = cigar; b.seq_off = so; b.seq = seq; b.qual_off = lo; b.qual = qual;
        b.nm = nm; b.nm_type = nmt; b.qname_bytes = qb; b.cigar_words = cw; b.seq_bytes = sb; b.qual_bytes = lb;
        if ((rc = gce_submit(e, &b)) != GCE_OK || (rc = gce_process(e)) != GCE_OK) { fprintf(stderr, "engine %d: %s\n", rc, gce_last_error(e)); return 3; }
        gce_result *r = calloc(1, sizeof *r);
        if ((rc = gce_drain(e, r)) != GCE_OK) return 3;
        if (!have0) { pre0 = r->pre; post0 = r->post; have0 = 1; }
        /* (a short pause: the wrong word appeared between gce_process and the caller's look at the struct) */
        for (volatile int spin = 0; spin < 20000; spin++) { }
        if (memcmp(&r->pre, &pre0, sizeof pre0) || memcmp(&r->post, &post0, sizeof post0)) {
            bad++;
            const int64_t *a = (const int64_t *)&r->post, *z = (const int64_t *)&post0, *a1 = (const int64_t *)&r->pre, *z1 = (const int64_t *)&pre0;
            printf("lifetime %ld: the drained Stats differ from the first lifetime's:", it);
            for (size_t k = 0; k < sizeof post0 / 8; k++) { if (a[k] != z[k]) printf(" post[%zu] = %lld (want %lld)", k, (long long)a[k], (long long)z[k]); if (a1[k] != z1[k]) printf(" pre[%zu] = %lld (want %lld)", k, (long long)a1[k], (long long)z1[k]); }
            gce_result *r2 = calloc(1, sizeof *r2);
            gce_drain(e, r2);
            printf("; a second drain of the same engine %s\n", (memcmp(&r2->pre, &pre0, sizeof pre0) || memcmp(&r2->post, &post0, sizeof post0)) ? "is wrong too (the engine's host copy holds it)" : "is RIGHT (the first struct was written to after gce_drain filled it)");
            free(r2); fflush(stdout);
        }
        free(r);
        gce_destroy(e);
        /* heap blocks of the same size class, zeroed now, looked at one lifetime later */
        for (int c = 0; c < NCANARY; c++) {
            if (canary[c]) { const int64_t *w = canary[c]; for (size_t k = 0; k < sizeof(gce_result) / 8; k++) if (w[k]) { canary_hits++; printf("lifetime %ld: zeroed heap block %d has word %zu = %lld\n", it, c, k, (long long)w[k]); break; } free(canary[c]); }
            canary[c] = calloc(1, sizeof(gce_result));
        }
    }
    printf("stress_lifetimes: %ld lifetimes of a %llu-read stream, %ld with wrong Stats, %ld dirty heap blocks\n", lifetimes, (unsigned long long)n, bad, canary_hits);
    return 0;
}
