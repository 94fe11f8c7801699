/* tools/run_bam.c — gce_run_bam from a plain C process (no Python, no numpy, no torch in the address space): what the file path costs a
 * host like gencore's main() in wall time and in resident memory.   gcc -std=c11 -O1 -Iinclude tools/run_bam.c -Lgencore_amd/csrc -lgencore_amd
 *   run_bam <in.bam|in.sam> <out.bam|out.sam> <fasta or -> <threads> <level> <supporting reads> <repeats>
 * Prints one JSON line for the LAST repeat (page cache warm): stage times, records, VmHWM of the process. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gencore_amd.h"

int main(int argc, char **argv) {
    if (argc != 8) { fprintf(stderr, "usage: run_bam <in> <out> <fasta|-> <threads> <level> <supporting reads> <repeats>\n"); return 2; }
    gce_params prm;
    gce_params_default(&prm);
    memset(prm.umi_prefix, 0, sizeof prm.umi_prefix); strcpy(prm.umi_prefix, "auto");
    prm.cluster_size_req = atoi(argv[6]);
    const int reps = atoi(argv[7]) > 0 ? atoi(argv[7]) : 1;
    gce_bam_run r; char err[256];
    for (int k = 0; k < reps; k++) {
        const int rc = gce_run_bam(argv[1], argv[2], (strcmp(argv[3], "-") && k == 0) ? argv[3] : NULL, &prm, atoi(argv[4]), 1 << 21, atoi(argv[5]), &r, err);
        if (rc != 0) { fprintf(stderr, "gce_run_bam: %d %s\n", rc, err); return 3; }
    }
    printf("{\"caller\": \"C (tools/run_bam.c)\", \"n_reads\": %lld, \"n_out\": %lld, \"total_s\": %.4f, \"input_pipeline_s\": %.4f, \"index_s\": %.4f, \"process_s\": %.4f, "
           "\"output_records_s\": %.4f, \"write_s\": %.4f, \"kernel_ms\": %.3f, \"rss_at_entry_mb\": %.1f, \"peak_rss_mb\": %.1f}\n",
           (long long)r.n_reads, (long long)r.n_out, r.total_s, r.open_s, r.index_s, r.process_s, r.drain_s, r.write_s, r.kernel_ms, r.rss_start_kb / 1024.0, r.peak_rss_kb / 1024.0);
    return 0;
}
