/* tests/cabi_driver.c — a plain C caller of include/gencore_amd.h (no Python, no ctypes): replays a dumped batch through
 *   gce_params_default -> gce_create -> gce_set_reference_ascii -> gce_submit (in two halves) -> gce_process -> gce_drain
 * and writes the result table to a file that tests/test_cabi_driver.py compares with the oracle.  Built by the test with
 *   gcc -std=c11 -Iinclude tests/cabi_driver.c -Lgencore_amd/csrc -lgencore_amd -Wl,-rpath,...
 * This is what a maintainer's adapter inside Gencore::consensus() does, minus htslib.
 *
 * dump file (little endian):  u64 magic, u64 n_reads, u64 n_targets, u64 qname_bytes, cigar_words, seq_bytes, qual_bytes,
 *   i32 cluster_size_req, i32 flush_period, char umi_prefix[32], u32 target_len[n_targets],
 *   u64 ref_len[n_targets] + that many ASCII bases each (0 = contig absent from the FASTA),
 *   gce_core[n], u64 qname_off[n], cigar_off[n], seq_off[n], qual_off[n], i32 nm[n], u8 nm_type[n], qname, cigar, seq, qual
 * exit codes: 0 ok, 2 usage / io, 3 the engine returned an error (its status and message go to stderr and the out file) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gencore_amd.h"

#define MAGIC 0x3142414345434721ull
static void *rd(FILE *f, size_t bytes) {
    void *p = malloc(bytes + 64);
    if (!p || (bytes && fread(p, 1, bytes, f) != bytes)) { fprintf(stderr, "short read\n"); exit(2); }
    memset((char *)p + bytes, 0, 64);
    return p;
}
static uint64_t rd64(FILE *f) { uint64_t v; if (fread(&v, 8, 1, f) != 1) { fprintf(stderr, "short read\n"); exit(2); } return v; }
static void fail(FILE *out, gce_engine *e, int rc) {
    fprintf(stderr, "gce status %d: %s (%s)\n", rc, gce_status_message(rc), e ? gce_last_error(e) : "");
    int64_t hdr[2] = { (int64_t)rc, 0 };
    fwrite(hdr, 8, 2, out); fclose(out);
    exit(3);
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: cabi_driver <dump> <out>\n"); return 2; }
    FILE *f = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
    if (!f || !out) { fprintf(stderr, "cannot open files\n"); return 2; }
    if (rd64(f) != MAGIC) { fprintf(stderr, "bad magic\n"); return 2; }
    const uint64_t n = rd64(f), nt = rd64(f), qb = rd64(f), cw = rd64(f), sb = rd64(f), lb = rd64(f);
    int32_t opts[2]; char prefix[32];
    if (fread(opts, 4, 2, f) != 2 || fread(prefix, 1, 32, f) != 32) return 2;
    uint32_t *tl = rd(f, nt * 4);

    gce_params prm;
    gce_params_default(&prm);
    if (prm.abi_version != GCE_ABI_VERSION || gce_abi_version() != GCE_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }
    prm.cluster_size_req = opts[0]; prm.flush_period = opts[1];
    memcpy(prm.umi_prefix, prefix, 32);
    prm.n_targets = (int32_t)nt; prm.target_len = tl;
    gce_engine *e = NULL;
    int rc = gce_create(&prm, &e);
    if (rc != GCE_OK) fail(out, NULL, rc);
    for (uint64_t t = 0; t < nt; t++) {
        const uint64_t len = rd64(f);
        char *bases = rd(f, len);
        if (len && (rc = gce_set_reference_ascii(e, (int32_t)t, bases, (int64_t)len)) != GCE_OK) fail(out, e, rc);
        free(bases);
    }
    gce_core *core = rd(f, n * sizeof(gce_core));
    uint64_t *qo = rd(f, n * 8), *co = rd(f, n * 8), *so = rd(f, n * 8), *lo = rd(f, n * 8);
    int32_t *nm = rd(f, n * 4); uint8_t *nmt = rd(f, n);
    char *qname = rd(f, qb); uint32_t *cigar = rd(f, cw * 4); uint8_t *seq = rd(f, sb), *qual = rd(f, lb);
    fclose(f);

    /* two submits continuing one sorted stream: offsets of the second half are relative to its own blobs */
    const uint64_t h = n / 2;
    for (int part = 0; part < 2; part++) {
        const uint64_t a = part ? h : 0, z = part ? n : h, m = z - a;
        if (m == 0) continue;
        gce_batch b; memset(&b, 0, sizeof b);
        uint64_t *q2 = malloc(m * 8), *c2 = malloc(m * 8), *s2 = malloc(m * 8), *l2 = malloc(m * 8);
        for (uint64_t i = 0; i < m; i++) { q2[i] = qo[a + i] - qo[a]; c2[i] = co[a + i] - co[a]; s2[i] = so[a + i] - so[a]; l2[i] = lo[a + i] - lo[a]; }
        b.n_reads = (int64_t)m; b.core = core + a;
        b.qname_off = q2; b.qname = qname + qo[a]; b.cigar_off = c2; b.cigar = cigar + co[a];
        b.seq_off = s2; b.seq = seq + so[a]; b.qual_off = l2; b.qual = qual + lo[a];
        b.nm = nm + a; b.nm_type = nmt + a;
        b.qname_bytes = (z < n ? qo[z] : qb) - qo[a]; b.cigar_words = (z < n ? co[z] : cw) - co[a];
        b.seq_bytes = (z < n ? so[z] : sb) - so[a]; b.qual_bytes = (z < n ? lo[z] : lb) - lo[a];
        if ((rc = gce_submit(e, &b)) != GCE_OK) fail(out, e, rc);
        free(q2); free(c2); free(s2); free(l2);
    }
    if ((rc = gce_process(e)) != GCE_OK) fail(out, e, rc);
    if (gce_process(e) == GCE_OK) { fprintf(stderr, "a second gce_process without a submit must fail\n"); return 2; }
    gce_result r;
    if ((rc = gce_drain(e, &r)) != GCE_OK) fail(out, e, rc);
    if ((uint64_t)r.n_reads != n) { fprintf(stderr, "n_reads %lld\n", (long long)r.n_reads); return 2; }

    /* out file: i64 status, i64 n_out, then per row: u32 src, u8 kind, u32 qname_src, i32 nm_new, i16 fr, i16 rr, u32 mate,
     * u32 l_qseq, packed bases, quals; then the two gce_stats blocks */
    int64_t hdr[2] = { 0, r.n_out };
    fwrite(hdr, 8, 2, out);
    for (int64_t k = 0; k < r.n_out; k++) {
        const uint32_t lq = (uint32_t)core[r.src[k]].l_qseq;
        fwrite(&r.src[k], 4, 1, out); fwrite(&r.kind[k], 1, 1, out); fwrite(&r.qname_src[k], 4, 1, out); fwrite(&r.nm_new[k], 4, 1, out);
        fwrite(&r.fr[k], 2, 1, out); fwrite(&r.rr[k], 2, 1, out); fwrite(&r.mate[k], 4, 1, out); fwrite(&lq, 4, 1, out);
        fwrite(r.seq + r.seq_off[k], 1, (lq + 1) / 2, out); fwrite(r.qual + r.qual_off[k], 1, lq, out);
    }
    fwrite(&r.pre, sizeof(gce_stats), 1, out); fwrite(&r.post, sizeof(gce_stats), 1, out);
    fclose(out);
    gce_timing t;
    if (gce_get_timing(e, &t) == GCE_OK) fprintf(stderr, "engine: %.3f ms, %lld clusters, %lld groups\n", t.total_ms, (long long)t.n_clusters, (long long)t.n_groups);
    gce_destroy(e);
    return 0;
}
