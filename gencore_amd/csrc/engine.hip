// engine.hip — C-ABI implementation (include/gencore_amd.h) of the MI355X consensus engine.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC engine.hip -o libgencore_amd.so
// No CPU fallback: every entry point that computes needs a HIP device.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "gce_kernels.hpp"
#include "gce_pair2.hpp"
#include "gce_vote.hpp"
#include "gce_deep.hpp"
#include "gce_output.hpp"
#include "gce_depth.hpp"
#include "gce_inflate.hpp"
#include "gce_deflate.hpp"

namespace {

// what the calling thread spent in hipMalloc / hipFree (GCE_RAW_TIMING prints it per gce_process: the question behind the sharded runner's slow boxes)
static thread_local double t_alloc_s = 0.0; static thread_local long t_alloc_n = 0; static thread_local size_t t_alloc_bytes = 0;
static inline double mono_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        const double t0 = mono_s();
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        t_alloc_s += mono_s() - t0; t_alloc_n++; t_alloc_bytes += want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
};

enum { EV_START = 0, EV_CLUSTER, EV_CSR, EV_DESCRIBE, EV_PAIRING, EV_SCORE, EV_CONSENSUS, EV_FINISH, EV_OUTPUT, EV_COUNT };

}  // namespace

// host buffer that is NOT value-initialised: the drained bases / qualities are hundreds of MB that the copy from the device overwrites
struct HostRaw {
    uint8_t *p = nullptr; size_t n = 0, cap = 0;
    ~HostRaw() { free(p); }
    void resize(size_t k) { if (k > cap) { free(p); p = (uint8_t *)malloc(k ? k : 1); cap = p ? k : 0; } n = p ? k : 0; }
    uint8_t *data() { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
};

struct gce_engine {
    gce_params prm{};
    std::vector<uint32_t> target_len;
    std::string err;
    hipStream_t stream = nullptr;
    hipEvent_t ev[EV_COUNT]{};
    // reference
    std::vector<DevBuf> ref_buf; std::vector<const uint8_t *> ref_ptr; std::vector<int64_t> ref_len, ref_win; bool have_win = false;
    DevBuf d_ref_ptr, d_ref_len, d_ref_win, d_target_len, d_target_cum;
    // what the device copies of the five hold (gce_process uploads them only when something changed)
    std::vector<const uint8_t *> up_ref_ptr; std::vector<int64_t> up_ref_len, up_win; std::vector<uint32_t> up_tl; const void *up_dev[5]{}; bool up_valid = false;
    // host staging (gce_submit)
    std::vector<gce_core> h_core; std::vector<uint64_t> h_qoff, h_coff, h_soff, h_loff, h_mioff;
    std::vector<char> h_qname, h_mi; std::vector<uint32_t> h_cigar; std::vector<uint8_t> h_seq, h_qual, h_nmt; std::vector<int32_t> h_nm; std::vector<uint64_t> h_tick;
    std::vector<int32_t> h_ev_tid, h_ev_pos;   // gce_set_flush_events
    bool have_mi = false, have_tick = false, have_events = false, host_mode = false, device_mode = false, processed = false;
    // streamed submission (gce_reserve): batches go straight to HBM on their own stream while the caller prepares the next one
    bool reserved = false; hipStream_t up_stream = nullptr; std::vector<hipEvent_t> up_events;
    hipStream_t aux_stream = nullptr; hipEvent_t aux_ev[4]{};      // deep streams: k_score2 beside the hand-on + k_deep_prepare (gce_process)
    size_t rs_n = 0, rs_q = 0, rs_c = 0, rs_s = 0, rs_l = 0, st_n = 0, st_q = 0, st_c = 0, st_s = 0, st_l = 0, st_m = 0; bool dev_concat = false;
    gce_batch dev_batch{};              // device pointers (either uploaded or caller-owned)
    DevBuf b_core, b_qoff, b_qname, b_coff, b_cigar, b_soff, b_seq, b_loff, b_qual, b_nm, b_nmt, b_mioff, b_mi, b_tick;
    // work buffers
    DevBuf uinfo, rdesc, spatch, slot, score, out_flag, orec, out_index, nmx;
    // output table (gce_result): device arrays + host copies
    DevBuf o_src, o_kind, o_qsrc, o_nm, o_fr, o_rr, o_mate, o_soff, o_qoff, o_seq, o_qual, o_key, o_rec, o_ksoff, o_kqoff, o_krow, o_rank64, o_part3, ref_ascii;
    int64_t n_out = 0; size_t out_seq_bytes = 0, out_qual_bytes = 0; int dev_error = 0; uint32_t dev_error_read = 0;
    DevBuf lrec, lout, bhdr, blk_base, ev_tid, ev_pos, ev_read, table, toff;
    // the raw BAM stream in HBM (gce_bamdev.hpp)
    DevBuf raw, rw_bad, rw_guess, rw_leave, rw_cnt, rw_base, rw_misc, rw_tmp, rw_off, rw_ncig, rw_nmpos, rw_rsize, rw_roff, rw_body;
    size_t raw_n = 0; bool raw_mode = false; int64_t raw_records = 0; uint64_t raw_body_bytes = 0;
    DevBuf z_comp, z_dir, z_err; size_t z_n = 0; std::vector<InfDir> z_members;      // BGZF members waiting for the GPU inflate (gce_raw_push_bgzf)
    // the sharded file runner (gce_raw_attach_mirror / gce_raw_select_shard): engines that receive every push to this one's raw stream; this engine's
    // share of the stream (reads gathered from the full batch; sh_sel = their places in the whole stream)
    DevBuf zo_slots, zo_sizes, zo_off, zo_out; uint64_t zo_bytes = 0;            // the output stream as BGZF blocks (gce_raw_deflate_output)
    std::vector<gce_engine *> mirrors;
    DevBuf sh_tickall, sh_shard, sh_flag, sh_sel, sh_core, sh_qoff, sh_coff, sh_soff, sh_loff, sh_nm, sh_nmt, sh_mioff, sh_tick, sh_roff, sh_nmpos, sh_keys, sh_stage; int64_t shard_n = -1;
    bool shard_cut_done = false;          // gce_raw_select_shard applied --quit_after_contig to the WHOLE stream: this engine's gce_process does not look for the cut again
    bool tab_clean = false; const void *tab_clean_ptr = nullptr;   // the bucket table is all-zero (k_scatter wipes what a step used)
    DevBuf cl_ikey, cl_start, cl_n, cl_npairs, cl_ngroups, cl_gbase, cl_nresult, cl_hasumi;
    DevBuf members, sorted, pl, pr, pu, pg, gpl, gpr, grp_begin, grp_n, gl_cluster, g_begin, g_np;
    DevBuf slow_args, deep_list, k64, slow_list, left_list, pf_flag, pf_list, pq_flag, pq_list, p16_flag, p16_list, pd_slab, gen_flag, gen_list, score_list, gw, g_wbase, vb_start, rp_left, rp_right, rp_merge, rp_rmerge, rp_umi, rp_umilen, rp_state, rp_supp, rp_nm, rp_qsl, rp_qsr, scan_part, si;
    StreamInfo h_si{};
    void *si_pin = nullptr, *si_pin_dev = nullptr; unsigned long long si_seq = 0;      // read_si: the block in mapped host memory + its sequence word
    gce_timing timing{};
    int64_t n = 0, n_pre = 0;            // reads processed; reads counted by the pre-Stats (one more when --quit_after_contig cut the stream)
    // depth statistics (gce_depth_stats)
    DevBuf dp_binoff, dp_regoff, dp_rs, dp_re, dp_pmax, dp_sorted, dp_depth, dp_bed, dp_where;      // dp_depth: the Stats-merge payload (gce_stats_payload_device)
    std::vector<int64_t> h_binoff, h_depth_pre, h_depth_post, h_bed_pre, h_bed_post; int64_t payload_words = 0;
    // host result copies
    std::vector<uint8_t> r_kind; HostRaw r_seq, r_qual; std::vector<uint32_t> r_src, r_qsrc, r_mate; std::vector<int32_t> r_nm; std::vector<int16_t> r_fr, r_rr;
    std::vector<uint64_t> r_soff, r_qoff;
};

static int fail(gce_engine *e, int code, const std::string &msg) { if (e) e->err = msg; return code; }
#define HIPCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return fail(e, _e == hipErrorOutOfMemory ? GCE_ERR_OOM : GCE_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); } while (0)

extern "C" {

int gce_abi_version(void) { return GCE_ABI_VERSION; }

void gce_params_default(gce_params *p) {          // src/options.cpp:4-40
    memset(p, 0, sizeof *p);
    p->abi_version = GCE_ABI_VERSION;
    p->proper_umi_diff_threshold = 1; p->unproper_umi_diff_threshold = 0; p->duplex_mismatch_threshold = 2;
    p->cluster_size_req = 1; p->base_score_req = 6; p->high_quality = 30; p->moderate_quality = 20; p->low_quality = 15;
    p->score_high = 8; p->score_moderate = 6; p->score_low = 4; p->score_bad = 2;
    p->skip_low_complexity_cluster_threshold = 1000; p->flush_period = 10000; p->score_percent_req = 0.8;
}

void gce_detect_umi_prefix(const char *q, char out[32]) {      // src/gencore.cpp:207-216
    memset(out, 0, 32);
    if (strstr(q, "umi_")) strcpy(out, "umi");
    else if (strstr(q, "UMI_")) strcpy(out, "UMI");
}

const char *gce_status_message(int s) {
    switch (s) {
    case GCE_OK: return "ok";
    case GCE_ERR_INVALID: return "invalid argument or call order";
    case GCE_ERR_NO_DEVICE: return "no HIP device available (this engine has no CPU fallback)";
    case GCE_ERR_HIP: return "HIP runtime error";
    case GCE_ERR_OOM: return "out of device memory";
    case GCE_ERR_UNSORTED: return "ERROR: the input is unsorted. Please sort the input first.";
    case GCE_ERR_UMI_MISMATCH: return "The UMI of a read pair should be identical";
    case GCE_ERR_NM_MISSING: return "NM tag missing on a consensus template whose mismatch count changed";
    case GCE_ERR_UMI_PARSE: return "UMI parse: substr start beyond the end of the read name";
    case GCE_ERR_QNAME_SHORT: return "copyQName ERROR: desitination qname is shorter";
    case GCE_ERR_REF_WINDOW: return "reference window: a clustered read lies outside the bases staged for its contig";
    default: return "unknown status";
    }
}

const char *gce_last_error(const gce_engine *e) { return e ? e->err.c_str() : "null engine"; }

int gce_create(const gce_params *params, gce_engine **out) {
    if (!params || !out || params->abi_version != GCE_ABI_VERSION) return GCE_ERR_INVALID;
    if (params->proper_umi_diff_threshold < 0 || params->unproper_umi_diff_threshold < 0 || params->flush_period < 0) return GCE_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || params->device >= ndev) return GCE_ERR_NO_DEVICE;
    gce_engine *e = new gce_engine();
    e->prm = *params;
    if (e->prm.flush_period == 0) e->prm.flush_period = 10000;
    if (params->n_targets > 0 && params->target_len) e->target_len.assign(params->target_len, params->target_len + params->n_targets);
    e->prm.target_len = nullptr;
    if (hipSetDevice(params->device) != hipSuccess || hipStreamCreate(&e->stream) != hipSuccess) { delete e; return GCE_ERR_HIP; }
    for (auto &v : e->ev) (void)hipEventCreate(&v);
    *out = e;
    return GCE_OK;
}

void gce_destroy(gce_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->prm.device);
    (void)hipStreamSynchronize(e->stream);
    DevBuf *all[] = {&e->d_ref_ptr, &e->d_ref_len, &e->d_ref_win, &e->d_target_len, &e->d_target_cum, &e->b_core, &e->b_qoff, &e->b_qname, &e->b_coff, &e->b_cigar, &e->b_soff,
                     &e->b_seq, &e->b_loff, &e->b_qual, &e->b_nm, &e->b_nmt, &e->b_mioff, &e->b_mi, &e->b_tick, &e->uinfo, &e->rdesc, &e->spatch,
                     &e->slot, &e->score, &e->out_flag, &e->orec, &e->out_index, &e->nmx, &e->o_src, &e->o_kind, &e->o_qsrc, &e->o_nm, &e->o_fr, &e->o_rr, &e->o_mate,
                     &e->o_key, &e->o_rec, &e->o_ksoff, &e->o_kqoff, &e->o_krow, &e->o_rank64, &e->o_part3, &e->o_soff, &e->o_qoff, &e->o_seq, &e->o_qual, &e->ref_ascii, &e->lrec, &e->lout, &e->bhdr,
                     &e->blk_base, &e->ev_tid, &e->ev_pos, &e->ev_read, &e->table, &e->toff, &e->cl_ikey, &e->cl_start, &e->cl_n,
                     &e->cl_npairs, &e->cl_ngroups, &e->cl_gbase, &e->cl_nresult, &e->cl_hasumi, &e->members, &e->sorted, &e->pl, &e->pr, &e->pu,
                     &e->pg, &e->gpl, &e->gpr, &e->grp_begin, &e->grp_n, &e->gl_cluster, &e->g_begin, &e->g_np, &e->deep_list, &e->k64, &e->slow_list, &e->pf_flag, &e->pf_list, &e->pq_flag, &e->pq_list, &e->left_list, &e->slow_args, &e->pd_slab, &e->gen_flag, &e->gen_list, &e->score_list, &e->gw, &e->g_wbase, &e->vb_start, &e->rp_left, &e->rp_right, &e->rp_merge, &e->rp_rmerge,
                     &e->rp_umi, &e->rp_umilen, &e->rp_state, &e->rp_supp, &e->rp_nm, &e->rp_qsl, &e->rp_qsr, &e->scan_part, &e->si};
    for (auto *b : all) b->release();
    for (DevBuf *b : {&e->z_comp, &e->z_dir, &e->z_err, &e->raw, &e->rw_bad, &e->rw_guess, &e->rw_leave, &e->rw_cnt, &e->rw_base, &e->rw_misc, &e->rw_tmp, &e->rw_off, &e->rw_ncig, &e->rw_nmpos, &e->rw_rsize, &e->rw_roff, &e->rw_body}) b->release();
    for (DevBuf *b : {&e->zo_slots, &e->zo_sizes, &e->zo_off, &e->zo_out, &e->p16_flag, &e->p16_list}) b->release();
    for (DevBuf *b : {&e->sh_tickall, &e->sh_shard, &e->sh_flag, &e->sh_sel, &e->sh_core, &e->sh_qoff, &e->sh_coff, &e->sh_soff, &e->sh_loff, &e->sh_nm, &e->sh_nmt, &e->sh_mioff, &e->sh_tick, &e->sh_roff, &e->sh_nmpos, &e->sh_keys, &e->sh_stage}) b->release();
    for (DevBuf *b : {&e->dp_binoff, &e->dp_regoff, &e->dp_rs, &e->dp_re, &e->dp_pmax, &e->dp_sorted, &e->dp_depth, &e->dp_bed, &e->dp_where}) b->release();
    for (auto ev : e->up_events) (void)hipEventDestroy(ev);
    if (e->up_stream) { (void)hipStreamSynchronize(e->up_stream); (void)hipStreamDestroy(e->up_stream); }
    if (e->aux_stream) { (void)hipStreamSynchronize(e->aux_stream); (void)hipStreamDestroy(e->aux_stream); }
    for (auto &v : e->aux_ev) if (v) (void)hipEventDestroy(v);
    for (auto &b : e->ref_buf) b.release();
    for (auto &v : e->ev) if (v) (void)hipEventDestroy(v);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->si_pin) (void)hipHostFree(e->si_pin);
    delete e;
}

void gce_pack_reference(const char *bases, int64_t n, uint8_t *out) {      // FastaReader::to4bits, src/fastareader.cpp:139-152
    memset(out, 0, (size_t)((n + 1) / 2));
    for (int64_t i = 0; i < n; i++) {
        uint8_t code;
        switch (bases[i]) { case 'A': code = 1; break; case 'T': code = 2; break; case 'C': code = 3; break; case 'G': code = 4; break; default: code = 0; }
        out[i >> 1] |= (i & 1) ? (uint8_t)(code << 4) : code;
    }
}

int gce_set_reference(gce_engine *e, int32_t tid, const uint8_t *nibbles, int64_t n_bases) {
    if (!e || tid < 0 || n_bases < 0) return GCE_ERR_INVALID;
    (void)hipSetDevice(e->prm.device);
    if ((size_t)tid >= e->ref_buf.size()) { e->ref_buf.resize(tid + 1); e->ref_ptr.resize(tid + 1, nullptr); e->ref_len.resize(tid + 1, 0); }
    if (!nibbles) { e->ref_ptr[tid] = nullptr; e->ref_len[tid] = 0; return GCE_OK; }
    size_t bytes = (size_t)((n_bases + 1) / 2);
    HIPCHK(e->ref_buf[tid].ensure(bytes + 16));
    hipPointerAttribute_t attr; bool is_dev = hipPointerGetAttributes(&attr, nibbles) == hipSuccess && attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    HIPCHK(hipMemcpyAsync(e->ref_buf[tid].p, nibbles, bytes, is_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->ref_ptr[tid] = e->ref_buf[tid].as<uint8_t>(); e->ref_len[tid] = n_bases;
    if (2 * (size_t)tid + 1 < e->ref_win.size()) e->ref_win[2 * tid] = e->ref_win[2 * tid + 1] = 0;
    return GCE_OK;
}

// One contig of FastaReader::mAllContigs as upper-cased ASCII: packed to the 4-bit code on the GPU (fastareader.cpp:139-152).
int gce_set_reference_ascii(gce_engine *e, int32_t tid, const char *bases, int64_t n_bases) {
    if (!e || tid < 0 || n_bases < 0 || (!bases && n_bases > 0)) return GCE_ERR_INVALID;
    (void)hipSetDevice(e->prm.device);
    if ((size_t)tid >= e->ref_buf.size()) { e->ref_buf.resize(tid + 1); e->ref_ptr.resize(tid + 1, nullptr); e->ref_len.resize(tid + 1, 0); }
    const size_t bytes = (size_t)((n_bases + 1) / 2);
    HIPCHK(e->ref_buf[tid].ensure(bytes + 16));
    HIPCHK(e->ref_ascii.ensure((size_t)n_bases + 16));
    hipPointerAttribute_t attr; const bool is_dev = hipPointerGetAttributes(&attr, bases) == hipSuccess && attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    if (n_bases) {
        HIPCHK(hipMemcpyAsync(e->ref_ascii.p, bases, (size_t)n_bases, is_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream));
        const unsigned grid = (unsigned)std::min<uint64_t>((bytes + 255) / 256, 65535u * 4u);
        hipLaunchKernelGGL(k_pack_reference, dim3(grid), dim3(256), 0, e->stream, (const char *)e->ref_ascii.p, n_bases, e->ref_buf[tid].as<uint8_t>());
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    e->ref_ptr[tid] = e->ref_buf[tid].as<uint8_t>(); e->ref_len[tid] = n_bases;
    if (2 * (size_t)tid + 1 < e->ref_win.size()) e->ref_win[2 * tid] = e->ref_win[2 * tid + 1] = 0;
    return GCE_OK;
}

int gce_set_reference_window(gce_engine *e, int32_t tid, int64_t contig_len, int64_t win_start, const char *bases, int64_t n_bases) {
    if (!e || tid < 0 || contig_len < 0 || win_start < 0 || (win_start & 1) || n_bases < 0 || win_start + n_bases > contig_len || (!bases && n_bases > 0)) return GCE_ERR_INVALID;
    const int rc = gce_set_reference_ascii(e, tid, bases, n_bases);             // the window's bases, packed from offset 0 (win_start is even: the nibble parity is that of the contig)
    if (rc != GCE_OK) return rc;
    if (e->ref_win.size() < 2 * e->ref_ptr.size()) e->ref_win.resize(2 * e->ref_ptr.size(), 0);
    e->ref_ptr[tid] = e->ref_buf[tid].as<uint8_t>() - (win_start >> 1);           // virtual origin: ref[pos >> 1] for pos inside the window
    e->ref_len[tid] = contig_len;
    e->ref_win[2 * tid] = win_start; e->ref_win[2 * tid + 1] = win_start + n_bases;
    e->have_win = true;
    return GCE_OK;
}

int gce_reset(gce_engine *e) {
    if (!e) return GCE_ERR_INVALID;
    e->h_core.clear(); e->h_qoff.clear(); e->h_coff.clear(); e->h_soff.clear(); e->h_loff.clear(); e->h_mioff.clear(); e->h_tick.clear();
    e->h_qname.clear(); e->h_mi.clear(); e->h_cigar.clear(); e->h_seq.clear(); e->h_qual.clear(); e->h_nmt.clear(); e->h_nm.clear();
    e->have_mi = e->have_tick = e->host_mode = e->device_mode = e->processed = e->raw_mode = false; e->n = 0; e->n_out = 0; e->shard_n = -1; e->shard_cut_done = false;
    e->st_n = e->st_q = e->st_c = e->st_s = e->st_l = e->st_m = 0; e->dev_concat = false;   // a reservation (gce_reserve) stays
    return GCE_OK;
}

// The flush events of the whole stream (gencore.cpp:319-322), for shards cut inside a contig: see include/gencore_amd.h.
int gce_set_flush_events(gce_engine *e, int32_t n_events, const int32_t *ev_tid, const int32_t *ev_pos) {
    if (!e || n_events < 0 || (n_events > 0 && (!ev_tid || !ev_pos))) return GCE_ERR_INVALID;
    e->h_ev_tid.assign(ev_tid, ev_tid + n_events); e->h_ev_pos.assign(ev_pos, ev_pos + n_events);
    e->have_events = true;
    return GCE_OK;
}

// the six addRead counters of either Stats block are spread over GCE_PRE_SLOTS words each (k_describe, k_out_meta): summed into the
// blocks themselves, the slots cleared, so that the device copy is complete (the host adds the slots of ITS copy: zeros after this)
struct PreExtra { long long v[6]; };      // addRead of the read on which --quit_after_contig ended the loop (counted, then ignored)
__global__ void k_fold_stats(StreamInfo *si, PreExtra x) {
    const int k = threadIdx.x;
    if (k < 6) {
        long long a = 0, c = 0;
        for (int q = 0; q < GCE_PRE_SLOTS; q++) { a += si->pre_slot[q][k]; c += si->post_slot[q][k]; si->pre_slot[q][k] = 0; si->post_slot[q][k] = 0; }
        si->pre[k] += a + x.v[k]; si->post[k] += c;
    }
}
// --quit_after_contig: the first read whose tid >= max_contig (the stream is sorted, unmapped reads -- tid -1 -- never match)
__global__ void k_first_contig_ge(const gce_core *core, int64_t n, int32_t maxc, unsigned int *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool hit = i < n && core[i].tid >= maxc;
    const unsigned long long m = __ballot(hit);
    if (m && (threadIdx.x & 63) == 0) { const unsigned int v = (unsigned int)(i + __ffsll((long long)m) - 1); if (v < *(volatile unsigned int *)out) atomicMin(out, v); }
}

__global__ void k_add_u64(uint64_t *a, uint64_t n, uint64_t base) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += base;
}

// Pre-size the HBM copy of a stream whose totals are known (a BAM file that was indexed, gce_bam_get_info): from then on
// gce_submit / gce_submit_async copy every batch straight into place -- no host staging, and the copy of batch k overlaps whatever
// the caller does to prepare batch k + 1.  MI tags and gce_batch.tick travel too (their buffers grow with the batches).
int gce_reserve(gce_engine *e, int64_t n_reads, size_t qname_bytes, size_t cigar_words, size_t seq_bytes, size_t qual_bytes) {
    if (!e || n_reads < 0) return GCE_ERR_INVALID;
    if (e->host_mode || e->device_mode) return fail(e, GCE_ERR_INVALID, "gce_reserve after a submit");
    (void)hipSetDevice(e->prm.device);
    const size_t n = (size_t)n_reads;
    HIPCHK(e->b_core.ensure(n * sizeof(gce_core) + 64)); HIPCHK(e->b_qoff.ensure(n * 8 + 64)); HIPCHK(e->b_coff.ensure(n * 8 + 64));
    HIPCHK(e->b_soff.ensure(n * 8 + 64)); HIPCHK(e->b_loff.ensure(n * 8 + 64)); HIPCHK(e->b_nm.ensure(n * 4 + 64)); HIPCHK(e->b_nmt.ensure(n + 64));
    HIPCHK(e->b_qname.ensure(qname_bytes + 64)); HIPCHK(e->b_cigar.ensure(cigar_words * 4 + 64)); HIPCHK(e->b_seq.ensure(seq_bytes + 64)); HIPCHK(e->b_qual.ensure(qual_bytes + 64));
    if (!e->up_stream) HIPCHK(hipStreamCreate(&e->up_stream));
    e->reserved = true; e->rs_n = n; e->rs_q = qname_bytes; e->rs_c = cigar_words; e->rs_s = seq_bytes; e->rs_l = qual_bytes;
    e->st_n = e->st_q = e->st_c = e->st_s = e->st_l = 0;
    return GCE_OK;
}

// a buffer of the engine's own copy of the stream that holds `used` bytes and must take `need`: kept, or replaced by a larger one with the old bytes moved
// (on the upload stream, behind the copies already queued there)
static hipError_t grow_keep(DevBuf &d, size_t used, size_t need, hipStream_t s) {
    if (d.p && need <= d.cap) return hipSuccess;
    DevBuf nb;
    hipError_t rc = nb.ensure(need + need / 2 + 256);
    if (rc != hipSuccess) return rc;
    if (used && d.p && (rc = hipMemcpyAsync(nb.p, d.p, used, hipMemcpyDeviceToDevice, s)) != hipSuccess) { nb.release(); return rc; }
    if ((rc = hipStreamSynchronize(s)) != hipSuccess) { nb.release(); return rc; }
    d.release(); d = nb;
    return hipSuccess;
}
__global__ void k_rebase_mioff(uint64_t *a, uint64_t n, uint64_t base) {     // MI offsets of a batch -> offsets of the stream (UINT64_MAX = no tag stays)
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != 0xFFFFFFFFFFFFFFFFull) a[i] += base;
}

// One batch behind the batches the engine's own copy of the stream already holds (b_* buffers, st_* fill levels), asynchronously on the upload stream: from host
// memory (gce_submit_async / gce_submit on a reserved engine) or from device memory (the second and further gce_submit_device of a stream).  The buffers grow when
// there is no reservation or the batch exceeds it.  MI tags and per-read ticks travel too (round 5: a UMI-tagged file, or a key-range shard, through the streamed
// path -- bamutil.cpp:23-38, gencore.cpp:319-322).
static int append_batch(gce_engine *e, const gce_batch *b, hipMemcpyKind kind) {
    const size_t n = (size_t)b->n_reads;
    if (!n) return GCE_OK;
    if (!b->core || !b->qname_off || !b->qname || !b->cigar_off || !b->seq_off || !b->seq || !b->qual_off || !b->qual || !b->nm || !b->nm_type)
        return fail(e, GCE_ERR_INVALID, "null array in gce_batch");
    if (e->st_n > 0 && (b->tick != nullptr) != e->have_tick) return fail(e, GCE_ERR_INVALID, "gce_batch.tick must be given for every batch of a stream or for none");
    if (!e->up_stream) HIPCHK(hipStreamCreate(&e->up_stream));
    hipStream_t s = e->up_stream;
    const bool mi = b->mi && b->mi_off;
    HIPCHK(grow_keep(e->b_core, e->st_n * sizeof(gce_core), (e->st_n + n) * sizeof(gce_core) + 64, s)); HIPCHK(grow_keep(e->b_nm, e->st_n * 4, (e->st_n + n) * 4 + 64, s)); HIPCHK(grow_keep(e->b_nmt, e->st_n, e->st_n + n + 64, s));
    HIPCHK(grow_keep(e->b_qoff, e->st_n * 8, (e->st_n + n) * 8 + 64, s)); HIPCHK(grow_keep(e->b_coff, e->st_n * 8, (e->st_n + n) * 8 + 64, s));
    HIPCHK(grow_keep(e->b_soff, e->st_n * 8, (e->st_n + n) * 8 + 64, s)); HIPCHK(grow_keep(e->b_loff, e->st_n * 8, (e->st_n + n) * 8 + 64, s));
    HIPCHK(grow_keep(e->b_qname, e->st_q, e->st_q + b->qname_bytes + 64, s)); HIPCHK(grow_keep(e->b_cigar, e->st_c * 4, (e->st_c + b->cigar_words) * 4 + 64, s));
    HIPCHK(grow_keep(e->b_seq, e->st_s, e->st_s + b->seq_bytes + 64, s)); HIPCHK(grow_keep(e->b_qual, e->st_l, e->st_l + b->qual_bytes + 64, s));
    auto cp = [&](DevBuf &d, size_t at, const void *src, size_t bytes) { return bytes ? hipMemcpyAsync((char *)d.p + at, src, bytes, kind, s) : hipSuccess; };
    HIPCHK(cp(e->b_core, e->st_n * sizeof(gce_core), b->core, n * sizeof(gce_core)));
    HIPCHK(cp(e->b_nm, e->st_n * 4, b->nm, n * 4)); HIPCHK(cp(e->b_nmt, e->st_n, b->nm_type, n));
    HIPCHK(cp(e->b_qoff, e->st_n * 8, b->qname_off, n * 8)); HIPCHK(cp(e->b_coff, e->st_n * 8, b->cigar_off, n * 8));
    HIPCHK(cp(e->b_soff, e->st_n * 8, b->seq_off, n * 8)); HIPCHK(cp(e->b_loff, e->st_n * 8, b->qual_off, n * 8));
    HIPCHK(cp(e->b_qname, e->st_q, b->qname, b->qname_bytes)); HIPCHK(cp(e->b_cigar, e->st_c * 4, b->cigar, b->cigar_words * 4));
    HIPCHK(cp(e->b_seq, e->st_s, b->seq, b->seq_bytes)); HIPCHK(cp(e->b_qual, e->st_l, b->qual, b->qual_bytes));
    const unsigned nb = (unsigned)((n + 255) / 256);                                // offsets of the batch -> offsets of the stream
    if (e->st_q) hipLaunchKernelGGL(k_add_u64, dim3(nb), dim3(256), 0, s, e->b_qoff.as<uint64_t>() + e->st_n, (uint64_t)n, (uint64_t)e->st_q);
    if (e->st_c) hipLaunchKernelGGL(k_add_u64, dim3(nb), dim3(256), 0, s, e->b_coff.as<uint64_t>() + e->st_n, (uint64_t)n, (uint64_t)e->st_c);
    if (e->st_s) hipLaunchKernelGGL(k_add_u64, dim3(nb), dim3(256), 0, s, e->b_soff.as<uint64_t>() + e->st_n, (uint64_t)n, (uint64_t)e->st_s);
    if (e->st_l) hipLaunchKernelGGL(k_add_u64, dim3(nb), dim3(256), 0, s, e->b_loff.as<uint64_t>() + e->st_n, (uint64_t)n, (uint64_t)e->st_l);
    if (b->tick) {
        HIPCHK(grow_keep(e->b_tick, e->st_n * 8, (e->st_n + n) * 8 + 64, s));
        HIPCHK(cp(e->b_tick, e->st_n * 8, b->tick, n * 8));
        e->have_tick = true;
    }
    if (mi || e->have_mi) {                                                         // MI offsets: UINT64_MAX for the reads of batches without tags
        const bool first = !e->have_mi;
        HIPCHK(grow_keep(e->b_mioff, first ? 0 : e->st_n * 8, (e->st_n + n) * 8 + 64, s));
        if (first && e->st_n) HIPCHK(hipMemsetAsync(e->b_mioff.p, 0xFF, e->st_n * 8, s));
        if (mi) {
            HIPCHK(grow_keep(e->b_mi, e->st_m, e->st_m + b->mi_bytes + 64, s));
            HIPCHK(cp(e->b_mioff, e->st_n * 8, b->mi_off, n * 8)); HIPCHK(cp(e->b_mi, e->st_m, b->mi, b->mi_bytes));
            if (e->st_m) hipLaunchKernelGGL(k_rebase_mioff, dim3(nb), dim3(256), 0, s, e->b_mioff.as<uint64_t>() + e->st_n, (uint64_t)n, (uint64_t)e->st_m);
            e->st_m += b->mi_bytes;
        } else HIPCHK(hipMemsetAsync((char *)e->b_mioff.p + e->st_n * 8, 0xFF, n * 8, s));
        e->have_mi = true;
    }
    e->st_n += n; e->st_q += b->qname_bytes; e->st_c += b->cigar_words; e->st_s += b->seq_bytes; e->st_l += b->qual_bytes;
    return GCE_OK;
}
// the engine's own copy of the stream as the batch gce_process works on
static void publish_own_copy(gce_engine *e) {
    gce_batch &d = e->dev_batch; memset(&d, 0, sizeof d);
    d.n_reads = (int64_t)e->st_n; d.core = e->b_core.as<gce_core>();
    d.qname_off = e->b_qoff.as<uint64_t>(); d.qname = e->b_qname.as<char>(); d.cigar_off = e->b_coff.as<uint64_t>(); d.cigar = e->b_cigar.as<uint32_t>();
    d.seq_off = e->b_soff.as<uint64_t>(); d.seq = e->b_seq.as<uint8_t>(); d.qual_off = e->b_loff.as<uint64_t>(); d.qual = e->b_qual.as<uint8_t>();
    d.nm = e->b_nm.as<int32_t>(); d.nm_type = e->b_nmt.as<uint8_t>();
    d.qname_bytes = e->st_q; d.cigar_words = e->st_c; d.seq_bytes = e->st_s; d.qual_bytes = e->st_l;
    if (e->have_mi) { d.mi_off = e->b_mioff.as<uint64_t>(); d.mi = e->b_mi.as<char>(); d.mi_bytes = e->st_m; }
    if (e->have_tick) d.tick = e->b_tick.as<uint64_t>();
}

// One batch of a reserved stream, asynchronously: the caller's buffers must stay untouched until gce_submit_wait(ticket) (or
// gce_process) returns.  *ticket may be NULL.
int gce_submit_async(gce_engine *e, const gce_batch *b, int32_t *ticket) {
    if (!e || !b || b->n_reads < 0) return GCE_ERR_INVALID;
    if (!e->reserved) return fail(e, GCE_ERR_INVALID, "gce_submit_async needs gce_reserve");
    if (e->device_mode && !e->processed) return fail(e, GCE_ERR_INVALID, "gce_submit after gce_submit_device");
    if (e->processed) gce_reset(e);
    (void)hipSetDevice(e->prm.device);
    e->host_mode = true;
    int rc = append_batch(e, b, hipMemcpyHostToDevice);
    if (rc != GCE_OK) return rc;
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev, e->up_stream));
    e->up_events.push_back(ev);
    if (ticket) *ticket = (int32_t)e->up_events.size() - 1;
    return GCE_OK;
}

int gce_submit_wait(gce_engine *e, int32_t ticket) {
    if (!e || ticket < 0 || (size_t)ticket >= e->up_events.size()) return GCE_ERR_INVALID;
    HIPCHK(hipEventSynchronize(e->up_events[ticket]));
    for (gce_engine *m : e->mirrors)                                                // (the same pushes in the same order: the same ticket numbers)
        if ((size_t)ticket < m->up_events.size() && hipEventSynchronize(m->up_events[ticket]) != hipSuccess) return fail(e, GCE_ERR_HIP, "mirror engine: copy failed");
    return GCE_OK;
}

// Gencore::addToCluster for a whole batch (src/gencore.cpp:272,469): host buffers are staged, then uploaded by gce_process.
int gce_submit(gce_engine *e, const gce_batch *b) {
    if (!e || !b || b->n_reads < 0) return GCE_ERR_INVALID;
    if (e->device_mode && !e->processed) return fail(e, GCE_ERR_INVALID, "gce_submit after gce_submit_device");
    if (e->reserved) {                                                             // straight to HBM; the caller's buffers are free again on return
        int32_t t; int rc = gce_submit_async(e, b, &t);
        return rc != GCE_OK ? rc : gce_submit_wait(e, t);
    }
    if (e->processed) gce_reset(e);
    e->host_mode = true;
    const int64_t n = b->n_reads;
    if (n == 0) return GCE_OK;
    if (!b->core || !b->qname_off || !b->qname || !b->cigar_off || !b->seq_off || !b->seq || !b->qual_off || !b->qual || !b->nm || !b->nm_type)
        return fail(e, GCE_ERR_INVALID, "null array in gce_batch");
    const size_t old = e->h_core.size();
    if (old > 0 && (b->tick != nullptr) != e->have_tick) return fail(e, GCE_ERR_INVALID, "gce_batch.tick must be given for every batch of a stream or for none");
    const uint64_t q0 = e->h_qname.size(), c0 = e->h_cigar.size(), s0 = e->h_seq.size(), l0 = e->h_qual.size(), m0 = e->h_mi.size();
    e->h_core.insert(e->h_core.end(), b->core, b->core + n);
    e->h_nm.insert(e->h_nm.end(), b->nm, b->nm + n);
    e->h_nmt.insert(e->h_nmt.end(), b->nm_type, b->nm_type + n);
    e->h_qname.insert(e->h_qname.end(), b->qname, b->qname + b->qname_bytes);
    if (b->cigar_words) e->h_cigar.insert(e->h_cigar.end(), b->cigar, b->cigar + b->cigar_words);
    e->h_seq.insert(e->h_seq.end(), b->seq, b->seq + b->seq_bytes);
    e->h_qual.insert(e->h_qual.end(), b->qual, b->qual + b->qual_bytes);
    if (b->tick) { e->h_tick.insert(e->h_tick.end(), b->tick, b->tick + n); e->have_tick = true; }
    bool mi = b->mi && b->mi_off;
    if (mi && !e->have_mi) { e->h_mioff.assign(old, UINT64_MAX); e->have_mi = true; }
    if (mi) e->h_mi.insert(e->h_mi.end(), b->mi, b->mi + b->mi_bytes);
    e->h_qoff.reserve(old + n); e->h_coff.reserve(old + n); e->h_soff.reserve(old + n); e->h_loff.reserve(old + n);
    for (int64_t i = 0; i < n; i++) {
        e->h_qoff.push_back(b->qname_off[i] + q0); e->h_coff.push_back(b->cigar_off[i] + c0);
        e->h_soff.push_back(b->seq_off[i] + s0); e->h_loff.push_back(b->qual_off[i] + l0);
        if (e->have_mi) e->h_mioff.push_back(mi && b->mi_off[i] != UINT64_MAX ? b->mi_off[i] + m0 : UINT64_MAX);
    }
    return GCE_OK;
}

// Caller-owned device memory, zero copy -- for ONE batch.  A second gce_submit_device before gce_process makes the engine keep its own copy of the stream: the
// first batch and every further one are appended device to device (round 5; before, a stream had to be one batch).  The in-place mutations of the path
// (pair.cpp:158-159, group.cpp:525) then hit the engine's copy; the results come back through the compact blobs either way.
int gce_submit_device(gce_engine *e, const gce_batch *b) {
    if (!e || !b || b->n_reads < 0) return GCE_ERR_INVALID;
    if (e->processed) gce_reset(e);
    if (e->host_mode) return fail(e, GCE_ERR_INVALID, "gce_submit_device after gce_submit");
    (void)hipSetDevice(e->prm.device);
    if (!e->device_mode) {
        e->device_mode = true; e->dev_concat = false;
        e->dev_batch = *b;
        e->have_tick = b->tick != nullptr;
        return GCE_OK;
    }
    int rc;
    if (!e->dev_concat) {                                                          // the batch that came first, into the engine's own buffers
        const gce_batch first = e->dev_batch;
        e->st_n = e->st_q = e->st_c = e->st_s = e->st_l = e->st_m = 0; e->have_tick = false; e->have_mi = false;
        if ((rc = append_batch(e, &first, hipMemcpyDeviceToDevice)) != GCE_OK) return rc;
        e->dev_concat = true;
    }
    if ((rc = append_batch(e, b, hipMemcpyDeviceToDevice)) != GCE_OK) return rc;
    HIPCHK(hipStreamSynchronize(e->up_stream));                                    // (the caller may free its batch when this returns)
    publish_own_copy(e);
    return GCE_OK;
}

static int upload(gce_engine *e) {
    if (e->reserved) {                                                             // already in place (gce_submit_async): wait for the copies, publish the views
        HIPCHK(hipStreamSynchronize(e->up_stream));
        for (auto ev : e->up_events) (void)hipEventDestroy(ev);
        e->up_events.clear();
        publish_own_copy(e);
        return GCE_OK;
    }
    struct { DevBuf *d; const void *src; size_t bytes; const void **dst; } items[] = {
        {&e->b_core, e->h_core.data(), e->h_core.size() * sizeof(gce_core), (const void **)&e->dev_batch.core},
        {&e->b_qoff, e->h_qoff.data(), e->h_qoff.size() * 8, (const void **)&e->dev_batch.qname_off},
        {&e->b_qname, e->h_qname.data(), e->h_qname.size(), (const void **)&e->dev_batch.qname},
        {&e->b_coff, e->h_coff.data(), e->h_coff.size() * 8, (const void **)&e->dev_batch.cigar_off},
        {&e->b_cigar, e->h_cigar.data(), e->h_cigar.size() * 4, (const void **)&e->dev_batch.cigar},
        {&e->b_soff, e->h_soff.data(), e->h_soff.size() * 8, (const void **)&e->dev_batch.seq_off},
        {&e->b_seq, e->h_seq.data(), e->h_seq.size(), (const void **)&e->dev_batch.seq},
        {&e->b_loff, e->h_loff.data(), e->h_loff.size() * 8, (const void **)&e->dev_batch.qual_off},
        {&e->b_qual, e->h_qual.data(), e->h_qual.size(), (const void **)&e->dev_batch.qual},
        {&e->b_nm, e->h_nm.data(), e->h_nm.size() * 4, (const void **)&e->dev_batch.nm},
        {&e->b_nmt, e->h_nmt.data(), e->h_nmt.size(), (const void **)&e->dev_batch.nm_type},
        {&e->b_mioff, e->h_mioff.data(), e->h_mioff.size() * 8, (const void **)&e->dev_batch.mi_off},
        {&e->b_mi, e->h_mi.data(), e->h_mi.size(), (const void **)&e->dev_batch.mi},
        {&e->b_tick, e->h_tick.data(), e->h_tick.size() * 8, (const void **)&e->dev_batch.tick},
    };
    for (auto &it : items) {
        HIPCHK(it.d->ensure(it.bytes + 64));      // +64: string kernels may look one byte past a name
        if (it.bytes) HIPCHK(hipMemcpyAsync(it.d->p, it.src, it.bytes, hipMemcpyHostToDevice, e->stream));
        *it.dst = it.bytes ? it.d->p : nullptr;
    }
    if (!e->have_mi) { e->dev_batch.mi = nullptr; e->dev_batch.mi_off = nullptr; }
    if (!e->have_tick) e->dev_batch.tick = nullptr;
    e->dev_batch.n_reads = (int64_t)e->h_core.size();
    e->dev_batch.qname_bytes = e->h_qname.size(); e->dev_batch.cigar_words = e->h_cigar.size();
    e->dev_batch.seq_bytes = e->h_seq.size(); e->dev_batch.qual_bytes = e->h_qual.size(); e->dev_batch.mi_bytes = e->h_mi.size();
    if (e->h_cigar.empty()) e->dev_batch.cigar = e->b_cigar.as<uint32_t>();
    return GCE_OK;
}

// The host's look at StreamInfo in the middle of a step (cluster count, group count + batch count, output count: three per step).  A copy +
// hipStreamSynchronize costs ~55 us of idle GPU each time (rocprofv3 trace: 7 us to the copy, 4 us copy, ~44 us until the next kernel starts --
// the interrupt-driven wake-up of the waiting thread).  Instead a one-block kernel writes the block into mapped, coherent host memory and
// then a sequence number (system-scope fences, release store); the host spins on that word (acquire loads), looking at the stream's state
// now and then so that a failed launch cannot hang it.  GCE_SYNC_COPY=1 takes the copy + synchronize path.
__global__ __launch_bounds__(256) void k_publish_si(const StreamInfo *si, StreamInfo *host, unsigned long long *flag, unsigned long long seq) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(si);
    uint32_t *dst = reinterpret_cast<uint32_t *>(host);
    for (unsigned i = threadIdx.x; i < sizeof(StreamInfo) / 4; i += 256) dst[i] = src[i];
    __threadfence_system();                                   // my words have left for host memory ...
    __syncthreads();                                          // ... and so have everybody's
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static_assert(sizeof(StreamInfo) % 4 == 0, "k_publish_si copies words");
// In two halves: the first enqueues the publishing kernel, the second waits for its word -- work that does not need the answer can be enqueued in between
// and keeps the GPU busy while the word travels (gce_process: k_describe behind the cluster count).
static int read_si_begin(gce_engine *e, unsigned long long *seq_out) {
    *seq_out = 0;
    static const bool sync_copy = getenv("GCE_SYNC_COPY") != nullptr;
    if (!sync_copy && !e->si_pin) {
        void *hp = nullptr;
        if (hipHostMalloc(&hp, sizeof(StreamInfo) + 64, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) == hipSuccess) {
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) { memset(hp, 0, sizeof(StreamInfo) + 64); e->si_pin = hp; e->si_pin_dev = dp; }
            else (void)hipHostFree(hp);
        }
        (void)hipGetLastError();
    }
    if (!sync_copy && e->si_pin) {
        unsigned long long *dflag = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(e->si_pin_dev) + ((sizeof(StreamInfo) + 7) & ~size_t(7)));
        const unsigned long long seq = ++e->si_seq;
        hipLaunchKernelGGL(k_publish_si, dim3(1), dim3(256), 0, e->stream, (const StreamInfo *)e->si.p, reinterpret_cast<StreamInfo *>(e->si_pin_dev), dflag, seq);
        HIPCHK(hipGetLastError());
        *seq_out = seq;
    }
    return GCE_OK;
}
static int read_si_end(gce_engine *e, unsigned long long seq) {
    bool published = false;
    if (seq) {
        StreamInfo *hs = reinterpret_cast<StreamInfo *>(e->si_pin);
        unsigned long long *hflag = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(e->si_pin) + ((sizeof(StreamInfo) + 7) & ~size_t(7)));
        for (unsigned spins = 1;; spins++) {
            if (__atomic_load_n(hflag, __ATOMIC_ACQUIRE) == seq) { published = true; break; }
            if ((spins & 0x3FFF) == 0) {                                              // now and then: is the stream still alive?
                const hipError_t q = hipStreamQuery(e->stream);
                if (q == hipSuccess) { published = __atomic_load_n(hflag, __ATOMIC_ACQUIRE) == seq; break; }     // everything ran: the word is there, or this path does not work here
                if (q != hipErrorNotReady) HIPCHK(q);
            }
            __builtin_ia32_pause();
        }
        if (published) memcpy(&e->h_si, hs, sizeof(StreamInfo));
    }
    if (!published) {
        HIPCHK(hipMemcpyAsync(&e->h_si, e->si.p, sizeof(StreamInfo), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    for (int k = 0; k < 6; k++) for (int q = 0; q < GCE_PRE_SLOTS; q++) { e->h_si.pre[k] += e->h_si.pre_slot[q][k]; e->h_si.post[k] += e->h_si.post_slot[q][k]; }    // k_describe's spread counters
    if (e->h_si.err_key != ~0ull) { e->dev_error = -(int)(e->h_si.err_key & 0xFF); e->dev_error_read = (uint32_t)(e->h_si.err_key >> 8); }
    return GCE_OK;
}
static int read_si(gce_engine *e) {
    unsigned long long seq = 0;
    const int rc = read_si_begin(e, &seq);
    return rc != GCE_OK ? rc : read_si_end(e, seq);
}

static inline unsigned cdiv(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }
// A phase clock that is the STOP stamp of the phase's last kernel (an event attached to the dispatch itself) instead of a hipEventRecord behind it: the
// record is a barrier packet of its own, 2.7 us between two kernels (tools/mb/event_gap.hip: 64 short kernels 456 us back to back, 632 us with a record
// behind each, 459 us with attached stop events)
static const bool g_record_events = getenv("GCE_RECORD_EVENTS") != nullptr;       // GCE_RECORD_EVENTS=1: plain launches with hipEventRecord behind them
#define LAUNCH_EV(kernel, grid, block, stream, ev, ...) do { \
    if (g_record_events) { hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__); (void)hipEventRecord(ev, stream); } \
    else hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, nullptr, ev, 0, __VA_ARGS__); } while (0)

// Several byte fills in ONE launch (hipMemsetAsync is a launch of its own per buffer: a step had eleven).  Buffers start on 16-byte boundaries (hipMalloc).
struct FillSeg { void *p; uint64_t bytes; uint32_t val; uint32_t pad; };
struct FillArgs { FillSeg s[6]; };
__global__ __launch_bounds__(256) void k_fill_many(FillArgs a) {
    const FillSeg g = a.s[blockIdx.y];
    const uint64_t n16 = g.bytes >> 4;
    const uint32_t v = g.val * 0x01010101u;
    uint4 *q = reinterpret_cast<uint4 *>(g.p);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) q[i] = make_uint4(v, v, v, v);
    if (blockIdx.x == 0 && threadIdx.x < (g.bytes & 15)) reinterpret_cast<uint8_t *>(g.p)[(n16 << 4) + threadIdx.x] = (uint8_t)g.val;
}
static void fill_many(hipStream_t s, std::initializer_list<FillSeg> segs) {
    FillArgs a{}; int n = 0; uint64_t mx = 0;
    for (const FillSeg &g : segs) { a.s[n++] = g; mx = std::max<uint64_t>(mx, g.bytes); }
    const unsigned gx = (unsigned)std::min<uint64_t>(std::max<uint64_t>((mx >> 4) / (256 * 8), 1), 2048);      // ~8 stores per thread for the longest buffer
    hipLaunchKernelGGL(k_fill_many, dim3(gx, (unsigned)n), dim3(256), 0, s, a);
}

// the second HIP stream of an engine (and its two events): made when a step first wants it
static int aux_ready(gce_engine *e) {
    if (e->aux_stream) return GCE_OK;
    HIPCHK(hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking));
    for (auto &v : e->aux_ev) HIPCHK(hipEventCreateWithFlags(&v, hipEventDisableTiming));
    return GCE_OK;
}

#ifdef GCE_SI_CANARY      // debugging build: after every phase of gce_process, is a word of the Stats blocks that no kernel of that phase writes non-zero?
static void si_canary(gce_engine *e, const char *phase) {
    StreamInfo hs; (void)hipStreamSynchronize(e->stream); (void)hipMemcpy(&hs, e->si.p, sizeof hs, hipMemcpyDeviceToHost);
    for (int k = 14; k < GCE_STATS_WORDS; k++) if ((k != 15 && hs.post[k] != 0) || hs.pre[k] < 0 || hs.post[k] < 0) { fprintf(stderr, "SI_CANARY after %s: pre[%d] = %lld post[%d] = %lld\n", phase, k, hs.pre[k], k, hs.post[k]); break; }
}
#define CANARY(x) si_canary(e, x)
#else
#define CANARY(x) do { } while (0)
#endif

// Every clusterByUMI of the stream (src/gencore.cpp:355 periodic, :409 end of file) + outputPair bookkeeping + the output order.
static int gce_process_impl(gce_engine *e);
int gce_process(gce_engine *e) {
    static const bool tlog = getenv("GCE_RAW_TIMING") != nullptr;
    if (!tlog) return gce_process_impl(e);
    const double a0 = t_alloc_s, w0 = mono_s(); const long n0 = t_alloc_n; const size_t b0 = t_alloc_bytes;
    const int rc = gce_process_impl(e);
    fprintf(stderr, "gce_process (device %d): %.4f s wall, of it %.4f s in %ld hipMalloc / hipFree calls for %.2f GB; kernels %.1f ms\n", e ? e->prm.device : -1, mono_s() - w0, t_alloc_s - a0, t_alloc_n - n0,
            (double)(t_alloc_bytes - b0) / 1e9, e ? e->timing.total_ms : 0.0);
    return rc;
}
static int gce_process_impl(gce_engine *e) {
    if (!e) return GCE_ERR_INVALID;
    if (!e->host_mode && !e->device_mode) return fail(e, GCE_ERR_INVALID, "nothing submitted");
    if (e->processed) return fail(e, GCE_ERR_INVALID, "gce_process called twice without a new submit (the stream was mutated in place)");
    if (e->have_tick && !e->have_events) return fail(e, GCE_ERR_INVALID, "gce_batch.tick needs gce_set_flush_events");
    (void)hipSetDevice(e->prm.device);
    int rc;
    if (e->host_mode && (rc = upload(e)) != GCE_OK) return rc;
    const gce_batch &hb = e->dev_batch;
    int64_t N = hb.n_reads;
    PreExtra pre_extra{};
    e->n_pre = N;
    if (e->prm.max_contig > 0 && N > 0 && !e->shard_cut_done) {                    // src/gencore.cpp:243-246: the loop ends on the first read of contig >= maxContig
        HIPCHK(e->si.ensure(sizeof(StreamInfo)));
        unsigned int first = NONE32;
        HIPCHK(hipMemcpyAsync(e->si.p, &first, 4, hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(k_first_contig_ge, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, e->stream, hb.core, N, e->prm.max_contig, (unsigned int *)e->si.p);
        HIPCHK(hipMemcpyAsync(&first, e->si.p, 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (first != NONE32) {
            gce_core c2; int32_t nm = 0; uint8_t nmt = 0;                            // (the sorted check of :233-241 cannot fail on this read: every read in front has a smaller tid)
            HIPCHK(hipMemcpy(&c2, hb.core + first, sizeof(gce_core), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(&nm, hb.nm + first, 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(&nmt, hb.nm_type + first, 1, hipMemcpyDeviceToHost));
            const int mism = nmt != 0 ? nm : 0;                                     // Stats::addRead (stats.cpp:101-121): the read is mapped (tid >= maxContig > 0)
            pre_extra.v[0] = 1; pre_extra.v[1] = c2.l_qseq; pre_extra.v[4] = mism; pre_extra.v[5] = mism > 0;
            N = (int64_t)first; e->n_pre = N + 1;
        }
    }
    e->n = N; e->n_out = 0; e->out_seq_bytes = e->out_qual_bytes = 0; e->dev_error = 0; e->dev_error_read = 0;
    e->processed = true;
    memset(&e->timing, 0, sizeof e->timing);
    memset(&e->h_si, 0, sizeof e->h_si);
    if (N >= (int64_t)0xFFFFFC00ll) return fail(e, GCE_ERR_INVALID, "more than 2^32 reads in one engine");
    const size_t n1 = (size_t)(N > 0 ? N : 1);

    DevBatch b{}; b.n = N; b.core = hb.core; b.qname_off = hb.qname_off; b.qname = hb.qname; b.cigar_off = hb.cigar_off; b.cigar = hb.cigar;
    b.seq_off = hb.seq_off; b.seq = hb.seq; b.qual_off = hb.qual_off; b.qual = hb.qual; b.nm = hb.nm; b.nm_type = hb.nm_type;
    b.mi_off = hb.mi_off; b.mi = hb.mi; b.tick = e->have_tick ? hb.tick : nullptr;

    // reference + params to the device -- only when they changed since the last step (four small copies and two stream synchronisations per step otherwise:
    // ~60 us of every 6 ms step in front of the first kernel)
    const int nref = (int)e->ref_ptr.size();
    HIPCHK(e->d_ref_ptr.ensure((size_t)(nref + 1) * 8)); HIPCHK(e->d_ref_len.ensure((size_t)(nref + 1) * 8));
    HIPCHK(e->d_target_len.ensure(e->target_len.size() * 4 + 4));
    HIPCHK(e->d_target_cum.ensure(e->target_len.size() * 8 + 8));
    std::vector<int64_t> win;
    if (e->have_win && nref) {                                                     // contigs staged whole: window = [0, length)
        win.assign(2 * (size_t)nref, 0);
        for (int t = 0; t < nref; t++) { const bool wd = 2 * (size_t)t + 1 < e->ref_win.size() && e->ref_win[2 * t + 1] > 0; win[2 * t] = wd ? e->ref_win[2 * t] : 0; win[2 * t + 1] = wd ? e->ref_win[2 * t + 1] : e->ref_len[t]; }
        HIPCHK(e->d_ref_win.ensure(win.size() * 8));
    }
    const void *dev_now[5] = {e->d_ref_ptr.p, e->d_ref_len.p, e->d_target_len.p, e->d_target_cum.p, e->d_ref_win.p};
    const bool same = e->up_valid && e->up_ref_ptr == e->ref_ptr && e->up_ref_len == e->ref_len && e->up_tl == e->target_len && e->up_win == win && memcmp(dev_now, e->up_dev, sizeof dev_now) == 0;
    if (!same) {
        e->up_valid = false;
        if (nref) {
            HIPCHK(hipMemcpyAsync(e->d_ref_ptr.p, e->ref_ptr.data(), (size_t)nref * 8, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(e->d_ref_len.p, e->ref_len.data(), (size_t)nref * 8, hipMemcpyHostToDevice, e->stream));
        }
        std::vector<uint64_t> cum(e->target_len.size()); uint64_t acc = 0;
        for (size_t i = 0; i < cum.size(); i++) { cum[i] = acc; acc += e->target_len[i]; }
        if (!e->target_len.empty()) {
            HIPCHK(hipMemcpyAsync(e->d_target_len.p, e->target_len.data(), e->target_len.size() * 4, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(e->d_target_cum.p, cum.data(), cum.size() * 8, hipMemcpyHostToDevice, e->stream));
        }
        if (!win.empty()) HIPCHK(hipMemcpyAsync(e->d_ref_win.p, win.data(), win.size() * 8, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));                                       // (the sources are locals and vectors that may change)
        e->up_ref_ptr = e->ref_ptr; e->up_ref_len = e->ref_len; e->up_tl = e->target_len; e->up_win = win; memcpy(e->up_dev, dev_now, sizeof dev_now);
        e->up_valid = true;
    }
    DevParams p{};
    p.proper_thr = e->prm.proper_umi_diff_threshold; p.unproper_thr = e->prm.unproper_umi_diff_threshold;
    p.duplex_mismatch_thr = e->prm.duplex_mismatch_threshold; p.cluster_size_req = e->prm.cluster_size_req; p.base_score_req = e->prm.base_score_req;
    p.high_q = e->prm.high_quality; p.moderate_q = e->prm.moderate_quality; p.low_q = e->prm.low_quality;
    p.s_high = e->prm.score_high; p.s_moderate = e->prm.score_moderate; p.s_low = e->prm.score_low; p.s_bad = e->prm.score_bad;
    p.skip_low_complexity_thr = e->prm.skip_low_complexity_cluster_threshold; p.duplex_only = e->prm.duplex_only; p.disable_duplex = e->prm.disable_duplex;
    p.period = e->prm.flush_period; p.score_percent_req = e->prm.score_percent_req;
    {   // scores are qual2score(q) in {s_*}, +4 on overlap match, -3 on overlap mismatch, or 0 (pair.cpp:77-172)
        int mn = std::min(std::min(p.s_high, p.s_moderate), std::min(p.s_low, p.s_bad)), mx = std::max(std::max(p.s_high, p.s_moderate), std::max(p.s_low, p.s_bad));
        p.score_bias = std::max(0, 3 - mn); p.score_max = std::max(0, mx + 4);
        if (p.score_max + p.score_bias > 255) return fail(e, GCE_ERR_INVALID, "score constants out of range");
        auto by = [&](int sc) { return (uint32_t)((sc + p.score_bias) & 0xFF); };
        p.q2s_lut = by(p.s_bad) | by(p.s_low) << 8 | by(p.s_moderate) << 16 | by(p.s_high) << 24;
        p.thr_low4 = 0x01010101u * (uint32_t)(p.low_q & 0xFF); p.thr_mod4 = 0x01010101u * (uint32_t)(p.moderate_q & 0xFF); p.thr_high4 = 0x01010101u * (uint32_t)(p.high_q & 0xFF);
        p.q2s_swar_ok = p.low_q >= 0 && p.low_q <= p.moderate_q && p.moderate_q <= p.high_q && p.high_q <= 127;
        // k_vote (gce_vote.hpp): scores in a sane range (the packed tally keys of decide_column_packed), and whether a unanimous column whose
        // top quality reaches `moderate` is bound to reach baseScoreReq too: that voter scores s_moderate or s_high (or >= min + 4 inside a
        // matching mate overlap) and nobody scores below 0
        p.s_min_lb = mn;
        p.vote_ok = mn >= 0 && mx + 4 <= 120 && p.moderate_q >= 0 && p.moderate_q <= 127 && p.base_score_req <= 100 && p.q2s_swar_ok;   // (nested thresholds: k_vote's items index q2s_lut by the number of thresholds passed)
        p.vote_accept_by_qual = std::min(std::min(p.s_moderate, p.s_high), mn + 4) >= std::max(p.base_score_req, 1);
    }
    memcpy(p.prefix, e->prm.umi_prefix, 32); p.prefix[31] = 0; p.prefix_len = (int)strlen(p.prefix);
    p.n_targets = (int)e->target_len.size(); p.target_len = e->d_target_len.as<uint32_t>(); p.target_cum = e->target_len.empty() ? nullptr : e->d_target_cum.as<uint64_t>();
    p.tick_offset = e->have_tick ? 0 : e->prm.tick_offset; p.tick_epoch0 = p.tick_offset / p.period; p.tick_rem0 = (int32_t)(p.tick_offset % p.period); p.trailing_flush = e->have_tick ? 0 : e->prm.trailing_flush;
    {   // packed cluster key of the bucket table: bits of the largest tid and of the longest contig
        uint32_t mx = 1; for (uint32_t v : e->target_len) mx = std::max(mx, v);
        int bt = 1; while (bt < 31 && (1ll << bt) < (long long)std::max<size_t>(e->target_len.size(), 1)) bt++;
        int bl = 1; while (bl < 32 && (1ull << bl) <= (uint64_t)mx) bl++;
        p.key_bt = bt; p.key_bl = bl;
    }
    p.n_ref = nref; p.ref_data = e->d_ref_ptr.as<const uint8_t *>(); p.ref_len = e->d_ref_len.as<int64_t>(); p.ref_win = win.empty() ? nullptr : e->d_ref_win.as<int64_t>();

    // ---- allocations that only depend on N
    Work w{};
    const int64_t n_sblk = (N + SB_READS - 1) / SB_READS;
    w.n_sblk = n_sblk;
    const int64_t max_events = e->have_tick ? (int64_t)e->h_ev_tid.size() + 2 : (p.tick_offset % p.period + N) / p.period + 2;
    w.max_events = (int)max_events;
    // buckets: 1.25 x reads (worst case, every read its own cluster, still probes at load 0.8; typical load is a few percent).
    uint64_t T = ((uint64_t)n1 + (uint64_t)n1 / 4 + 2 * SCAN_TILE - 1) / SCAN_TILE * SCAN_TILE;
    w.tsize = T; w.tinv = 1.0 / (double)T;
    {   // normal bucket words: OCC | x div T | right - left + 1 | reads (gce_cluster.hpp, k_leaders)
        uint64_t genome = 0; for (uint32_t v : e->target_len) genome += v;
        const uint64_t qmax = (genome * TAB_WAYS + TAB_WAYS) / T + 1;
        int cb = 1; while (cb < 33 && (1ull << cb) <= (uint64_t)N + 1024) cb++;
        int qb = 1; while (qb < 52 && (1ull << qb) <= qmax) qb++;
        p.nw_cb = cb; p.nw_bd = 62 - cb - qb; p.nw_ok = !e->target_len.empty() && p.nw_bd >= 1;
        if (p.nw_bd > 40) p.nw_bd = 40;
    }
    const size_t qual_bytes = hb.qual_bytes ? hb.qual_bytes : 1;
    const size_t nsb1 = (size_t)(n_sblk > 0 ? n_sblk : 1);
#define ENS(buf, bytes) HIPCHK(e->buf.ensure(bytes))
    ENS(uinfo, n1 * 8); ENS(rdesc, n1 * sizeof(ReadDescP)); ENS(spatch, n1 * 4); ENS(slot, n1 * 4 + 16); ENS(score, qual_bytes + 64);
    ENS(out_flag, n1 + 144);       /* (k_out_mate reads whole 64-byte blocks of flags) */ ENS(nmx, n1 * 4); ENS(orec, n1 * sizeof(OutRec)); ENS(out_index, n1 * 4);
    ENS(lrec, nsb1 * SB_READS * sizeof(LeadRec)); ENS(lout, nsb1 * SB_READS * sizeof(LeadOut));
    ENS(bhdr, nsb1 * sizeof(BlkHdr)); ENS(blk_base, nsb1 * 4);
    ENS(ev_tid, (size_t)max_events * 4); ENS(ev_pos, (size_t)max_events * 4); ENS(ev_read, (size_t)max_events * 4);
    ENS(table, T * sizeof(TabEntry)); ENS(toff, T * 4);
    ENS(k64, n1 * 24); ENS(members, n1 * 4); ENS(sorted, n1 * 4); ENS(pl, n1 * 4); ENS(pr, n1 * 4); ENS(pu, n1 * 4); ENS(pg, n1 * 4); ENS(gpl, n1 * 4); ENS(gpr, n1 * 4);
    ENS(grp_begin, n1 * 4); ENS(grp_n, n1 * 4); ENS(slow_list, n1 * 4 + 64);
    w.slow_list = e->slow_list.as<uint32_t>();
    ENS(deep_list, (n1 / 64 + 64) * 16); w.deep_list = e->deep_list.p;
    const unsigned nblk_N = cdiv(n1, SCAN_TILE);
    ENS(scan_part, std::max<size_t>(2 * nsb1, (size_t)2 * nblk_N) * 8 + 16);    /* 2 x: the group-side flags (<= 2 per read); the scan blocks' totals behind those of k_num_reduce's blocks */ ENS(si, sizeof(StreamInfo));
    w.uinfo = e->uinfo.as<uint64_t>(); w.rdesc = e->rdesc.as<ReadDescP>(); w.spatch = e->spatch.as<uint32_t>();
    w.slot = e->slot.as<uint32_t>(); w.score = e->score.as<int8_t>();
    w.out_flag = e->out_flag.as<uint8_t>(); w.nmx = e->nmx.as<uint32_t>(); w.orec = e->orec.as<OutRec>(); w.out_index = e->out_index.as<uint32_t>();
    w.lrec = e->lrec.as<LeadRec>(); w.lout = e->lout.as<LeadOut>(); w.bhdr = e->bhdr.as<BlkHdr>(); w.blk_base = e->blk_base.as<uint32_t>();
    w.ev_tid = e->ev_tid.as<int32_t>(); w.ev_pos = e->ev_pos.as<int32_t>(); w.ev_read = e->ev_read.as<uint32_t>();
    w.tab = e->table.as<TabEntry>(); w.toff = e->toff.as<uint32_t>();
    w.k64 = e->k64.as<uint64_t>(); w.members = e->members.as<uint32_t>(); w.sorted = e->sorted.as<uint32_t>(); w.pl = e->pl.as<uint32_t>(); w.pr = e->pr.as<uint32_t>();
    w.pu = e->pu.as<uint32_t>(); w.pg = e->pg.as<uint32_t>(); w.gpl = e->gpl.as<uint32_t>(); w.gpr = e->gpr.as<uint32_t>();
    w.grp_begin = e->grp_begin.as<uint32_t>(); w.grp_n = e->grp_n.as<uint32_t>();
    w.scan_part = e->scan_part.as<uint64_t>(); w.si = e->si.as<StreamInfo>();
    ENS(cl_ikey, n1 * 4); ENS(cl_start, n1 * 4); ENS(cl_n, n1 * 4);      // cl_* arrays are sized by N (a cluster has >= 1 read)
    w.cl_ikey = e->cl_ikey.as<uint32_t>(); w.cl_start = e->cl_start.as<uint32_t>(); w.cl_n = e->cl_n.as<uint32_t>();

    StreamInfo init{}; init.first_unmapped = NONE32; init.err_key = ~0ull; init.lq_min = 0x7FFFFFFF; init.lq_max = -1;
    if (e->have_tick) { init.n_events = init.n_events_a = (int)e->h_ev_tid.size(); }
    HIPCHK(hipMemcpyAsync(e->si.p, &init, sizeof init, hipMemcpyHostToDevice, e->stream));
    if (e->have_tick && !e->h_ev_tid.empty()) {
        HIPCHK(hipMemcpyAsync(e->ev_tid.p, e->h_ev_tid.data(), e->h_ev_tid.size() * 4, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->ev_pos.p, e->h_ev_pos.data(), e->h_ev_pos.size() * 4, hipMemcpyHostToDevice, e->stream));
    }
    hipStream_t s = e->stream;
    // The bucket table is cleared by its users (k_scatter): a memset only for a new allocation or after a step that did not get that far.
    if (!e->tab_clean || e->tab_clean_ptr != e->table.p) {
        HIPCHK(hipMemsetAsync(e->table.p, 0, e->table.cap, s));
        e->tab_clean_ptr = e->table.p;
    }
    e->tab_clean = false;
    HIPCHK(hipEventRecord(e->ev[EV_START], s));
    HIPCHK(hipMemsetAsync(e->out_flag.p, 0, n1, s));
    // ---- cluster formation (gce_cluster.hpp): the scan, then the leaders (ticks + flush events come with the batch for key-range shards)
    if (N > 0) LAUNCH_EV(k_cluster, dim3((unsigned)n_sblk), dim3(SB_T), s, e->ev[EV_CLUSTER], b, p, w);
#ifdef CL_PROF
    if (N > 0) {
        StreamInfo hs; (void)hipStreamSynchronize(s); (void)hipMemcpy(&hs, e->si.p, sizeof hs, hipMemcpyDeviceToHost);
        static const char *nm[3] = {"key records + keys", "LDS leaders", "stores"};
        const double blocks = (double)hs.prof[15];
        fprintf(stderr, "k_cluster phases, mean per block (us), %.0f blocks:", blocks);
        for (int k = 0; k < 3; k++) fprintf(stderr, " %s %.2f;", nm[k], blocks ? hs.prof[k] / blocks / 100.0 : 0.0);
        fprintf(stderr, "\n");
    }
#endif
    if (N <= 0) HIPCHK(hipEventRecord(e->ev[EV_CLUSTER], s));
    CANARY("EV_CLUSTER");
    if (N > 0) {
        if (!e->have_tick) {
            hipLaunchKernelGGL(k_blk_scan, dim3(1), dim3(1024), 0, s, w, p);
            hipLaunchKernelGGL(k_events, dim3(cdiv(max_events, EV_T / 64)), dim3(EV_T), 0, s, b, p, w);
        }
        const unsigned nb4 = cdiv(n_sblk, 4);
        hipLaunchKernelGGL(k_leaders, dim3(nb4), dim3(256), 0, s, b, p, w);
        hipLaunchKernelGGL(k_num_reduce, dim3(nb4), dim3(256), 0, s, w, p.nw_cb);
        hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, w.scan_part, (uint64_t)nb4, &w.si->n_clusters, (unsigned long long *)nullptr);      // (one partial per k_num_reduce block)
        hipLaunchKernelGGL(k_num_apply, dim3(nb4), dim3(256), 0, s, w);
        LAUNCH_EV(k_scatter, dim3((unsigned)n_sblk), dim3(SB_T), s, e->ev[EV_CSR], N, w);
    } else HIPCHK(hipEventRecord(e->ev[EV_CSR], s));
    CANARY("EV_CSR");
    // ---- per-read descriptors, UMI slices, pre-Stats: independent of the clusters, consumed by pairing and the vote.
    // The host's look at the cluster count is asked for IN FRONT of it and awaited behind it: the word travels while k_describe runs (what k_describe adds to the
    // block -- pre-Stats, read-length range, its errors -- is read by the later looks; the pairing kernels do not mind a read k_describe refused)
    unsigned long long si_seq1 = 0;
    if ((rc = read_si_begin(e, &si_seq1)) != GCE_OK) return rc;
    if (N > 0) {
#ifndef GCE_DESCRIBE_BLOCKS
#define GCE_DESCRIBE_BLOCKS 32768        // at most this many blocks: their Stats partial sums end in six global atomics each
#endif
        const int64_t n_tiles = (N + 255) / 256;
        int tpb = (int)((n_tiles + GCE_DESCRIBE_BLOCKS - 1) / GCE_DESCRIBE_BLOCKS); if (tpb < 1) tpb = 1;
        LAUNCH_EV(k_describe, dim3(cdiv(n_tiles, tpb)), dim3(256), s, e->ev[EV_DESCRIBE], b, p, w, tpb);
    } else HIPCHK(hipEventRecord(e->ev[EV_DESCRIBE], s));
    CANARY("EV_DESCRIBE");
    if ((rc = read_si_end(e, si_seq1)) != GCE_OK) return rc;
    HIPCHK(hipGetLastError());
    e->tab_clean = true;                          // k_scatter ran to the end
    const uint32_t C = (uint32_t)e->h_si.n_clusters;
    const size_t c1 = C ? C : 1;
    ENS(cl_npairs, c1 * 4); ENS(cl_ngroups, c1 * 4); ENS(cl_gbase, c1 * 4); ENS(cl_nresult, c1 * 4); ENS(cl_hasumi, c1);
    w.cl_npairs = e->cl_npairs.as<uint32_t>(); w.cl_ngroups = e->cl_ngroups.as<uint32_t>(); w.cl_gbase = e->cl_gbase.as<uint32_t>();
    w.cl_nresult = e->cl_nresult.as<uint32_t>(); w.cl_hasumi = e->cl_hasumi.as<uint8_t>();
    uint32_t NG = 0;
    bool si_behind_describe = false;               // the host's copy of StreamInfo holds what k_describe found (read-length range)
    if (C > 0 && e->dev_error == 0) {
        // (gpl / gpr need no clearing: every reader -- k_vote, k_score2 behind the slot flags, the per-side kernels, k_group_tail -- looks at the pair slots
        //  [g_begin, g_begin + g_np) of a group only, and the pairing kernels write both words of every one of those)
        // three tiers: 16 lanes per cluster, then 32 for what that flags, then the full wave; each hand-over is a flag array
        // compacted by the scan kernels (never one shared append counter)
        ENS(left_list, c1 * 4 + 64); w.left_list = e->left_list.as<uint32_t>();
        ENS(pf_flag, c1 + 64); ENS(pf_list, c1 * 4); ENS(pq_flag, c1 + 64); ENS(pq_list, c1 * 4); ENS(p16_flag, c1 + 64); ENS(p16_list, c1 * 4);
        w.pf_flag = e->pf_flag.as<uint8_t>(); w.pf_list = e->pf_list.as<uint32_t>(); w.pq_flag = e->pq_flag.as<uint8_t>(); w.pq_list = e->pq_list.as<uint32_t>(); w.p16_flag = e->p16_flag.as<uint8_t>(); w.p16_list = e->p16_list.as<uint32_t>();
        // size classes in front of the quarter-wave kernel (k_pair_classes) when a good share of the clusters is beyond it: mean cluster beyond 10 reads
        // (cfg3: 16 reads, 45 % of the clusters beyond 16 -- pairing 1.41 -> 1.26 ms; cfg2: 7 reads, nearly none -- the class pass would be 16 us for nothing)
        const bool pair_classes = (double)N > 10.0 * (double)C;
        if (!pair_classes) fill_many(s, {FillSeg{e->pf_flag.p, c1, 0u, 0u}, FillSeg{e->pq_flag.p, c1, 0u, 0u}});          // (with size classes every entry of the three flag arrays is written by k_pair_classes)
        const unsigned nbc = cdiv(C, SCAN_TILE);
        auto compact = [&](uint8_t *flag, uint32_t *list, unsigned long long *count) {
            hipLaunchKernelGGL(k_flag_reduce, dim3(nbc), dim3(256), 0, s, (const uint8_t *)flag, (uint64_t)C, w.scan_part);
            hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, w.scan_part, (uint64_t)nbc, count, (unsigned long long *)nullptr);
            hipLaunchKernelGGL(k_flag_apply, dim3(nbc), dim3(256), 0, s, (const uint8_t *)flag, (uint64_t)C, (const uint64_t *)w.scan_part, list);
        };
#ifdef PS_STOP
        {   // experiment builds (tools/pair_stop.sh): time the truncated k_pairing_sub<16> (all clusters) and <32> (all clusters as well: no list) alone and stop
            hipEvent_t a_, b_, c_; (void)hipEventCreate(&a_); (void)hipEventCreate(&b_); (void)hipEventCreate(&c_);
            (void)hipEventRecord(a_, s);
            hipLaunchKernelGGL(k_pairing_sub<16>, dim3(cdiv(C, 4 * WAVES_PER_BLOCK)), dim3(256), 0, s, b, p, w, C, (const uint32_t *)nullptr, (const unsigned long long *)nullptr, w.pq_flag, 0);
            (void)hipEventRecord(b_, s);
            hipLaunchKernelGGL(k_pairing_sub<32>, dim3(cdiv(C, 2 * WAVES_PER_BLOCK)), dim3(256), 0, s, b, p, w, C, (const uint32_t *)nullptr, (const unsigned long long *)nullptr, w.pf_flag, 0);
            (void)hipEventRecord(c_, s); (void)hipEventSynchronize(c_);
            float m1 = 0, m2 = 0; (void)hipEventElapsedTime(&m1, a_, b_); (void)hipEventElapsedTime(&m2, b_, c_);
            fprintf(stderr, "k_pairing_sub up to tick %d: <16> %.3f ms, <32> over ALL clusters %.3f ms\n", PS_STOP, m1, m2);
            return fail(e, GCE_ERR_INVALID, "experiment build");
        }
#endif
        ENS(pd_slab, PD_BIG_BLOCKS * PD_SLAB);
        if (pair_classes) {
            // the three register-resident tiers get their clusters BY SIZE beforehand (<= 16, <= 32, more) and run side by side: quarter- and half-wave kernels on the main
            // stream, the full-wave kernel and the LDS instantiation of k_pairing_deep -- the long, thin tail of the phase: 93 + 60 us with a few thousand waves -- on the second
            // one.  What a tier cannot take for its names / UMIs goes straight to the generic kernels' list (`direct`), which every tier would have handed it to in turn.
            hipLaunchKernelGGL(k_pair_class_reduce, dim3(nbc), dim3(256), 0, s, (const uint32_t *)w.cl_n, C, w.scan_part);
            hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, w.scan_part, (uint64_t)nbc, &w.si->n_p16_items, &w.si->n_pq_items);
            hipLaunchKernelGGL(k_pair_class_apply, dim3(nbc), dim3(256), 0, s, (const uint32_t *)w.cl_n, C, (const uint64_t *)w.scan_part, w.p16_list, w.pq_list, w.pf_list, w.si);
            static const bool tiers_aux = getenv("GCE_NO_AUX_STREAM") == nullptr;
            hipStream_t s2 = s;
            if (tiers_aux) {
                if ((rc = aux_ready(e)) != GCE_OK) return rc;
                s2 = e->aux_stream;
                HIPCHK(hipEventRecord(e->aux_ev[0], s));
                HIPCHK(hipStreamWaitEvent(s2, e->aux_ev[0], 0));
            }
            hipLaunchKernelGGL(k_pairing_fast, dim3(2048), dim3(256), 0, s2, b, p, w, C, (const uint32_t *)w.pf_list);
            hipLaunchKernelGGL(k_pairing_deep<false>, dim3(256), dim3(PD_T), 0, s2, b, p, w, (uint8_t *)nullptr);      // deep clusters in LDS; the rest -> left_list
            if (tiers_aux) HIPCHK(hipEventRecord(e->aux_ev[1], s2));
            hipLaunchKernelGGL(k_pairing_sub<16>, dim3(cdiv(C, 4 * WAVES_PER_BLOCK)), dim3(256), 0, s, b, p, w, C, (const uint32_t *)w.p16_list, (const unsigned long long *)&w.si->n_p16_items, w.pq_flag, 1);
            hipLaunchKernelGGL(k_pairing_sub<32>, dim3(cdiv(C, 2 * WAVES_PER_BLOCK)), dim3(256), 0, s, b, p, w, C, (const uint32_t *)w.pq_list, (const unsigned long long *)&w.si->n_pq_items, w.pf_flag, 1);
            if (tiers_aux) HIPCHK(hipStreamWaitEvent(s, e->aux_ev[1], 0));
        } else {
            hipLaunchKernelGGL(k_pairing_sub<16>, dim3(cdiv(C, 4 * WAVES_PER_BLOCK)), dim3(256), 0, s, b, p, w, C, (const uint32_t *)nullptr, (const unsigned long long *)nullptr, w.pq_flag, 0);
            compact(w.pq_flag, w.pq_list, &w.si->n_pq_items);
            hipLaunchKernelGGL(k_pairing_sub<32>, dim3(cdiv(C, 2 * WAVES_PER_BLOCK)), dim3(256), 0, s, b, p, w, C, (const uint32_t *)w.pq_list, (const unsigned long long *)&w.si->n_pq_items, w.pf_flag, 0);
            compact(w.pf_flag, w.pf_list, &w.si->n_pf_items);
            hipLaunchKernelGGL(k_pairing_fast, dim3(2048), dim3(256), 0, s, b, p, w, C, (const uint32_t *)w.pf_list);
            hipLaunchKernelGGL(k_pairing_deep<false>, dim3(256), dim3(PD_T), 0, s, b, p, w, (uint8_t *)nullptr);      // deep clusters in LDS; the rest -> left_list
        }
        hipLaunchKernelGGL(k_pairing_deep<true>, dim3(PD_BIG_BLOCKS), dim3(PD_T), 0, s, b, p, w, e->pd_slab.as<uint8_t>());   // what that left for its size (<= 65 534 reads), arrays in device memory
        hipLaunchKernelGGL(k_pairing_slow<0>, dim3(1024), dim3(256), 0, s, b, p, w);
        hipLaunchKernelGGL(k_pairing_slow<1>, dim3(1024, 16), dim3(256), 0, s, b, p, w);      // y: a cluster's 64-read blocks over 16 waves
        hipLaunchKernelGGL(k_pairing_slow<2>, dim3(1024), dim3(256), 0, s, b, p, w);
        const unsigned nblk_C = cdiv(C, SCAN_TILE);
        hipLaunchKernelGGL(k_scan_reduce, dim3(nblk_C), dim3(256), 0, s, (const uint32_t *)w.cl_ngroups, (uint64_t)C, w.scan_part);
        hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, w.scan_part, (uint64_t)nblk_C, &w.si->n_pairs /*scratch*/, &w.si->n_groups);
        hipLaunchKernelGGL(k_u32_apply, dim3(nblk_C), dim3(256), 0, s, (const uint32_t *)w.cl_ngroups, w.cl_gbase, (uint64_t)C, (const uint64_t *)w.scan_part);
        // the compact group list and the batches of k_vote: sized by N (groups <= pairs <= reads) so that all of it runs before the one
        // host round trip that fetches the group count and the batch count together
        ENS(gl_cluster, n1 * 4); ENS(g_begin, n1 * 4); ENS(g_np, n1 * 4); ENS(gw, n1 * 8); ENS(g_wbase, n1 * 4);
        const size_t vb_cap = ((VB_MINW + 3) * n1) / VB_W + 8;     // sum of weights <= VB_MINW x groups + pairs + VB_W x deep groups (> 32 pairs each)
        ENS(vb_start, vb_cap * 4);
        w.gl_cluster = e->gl_cluster.as<uint32_t>(); w.g_begin = e->g_begin.as<uint32_t>(); w.g_np = e->g_np.as<uint32_t>();
        w.gw = e->gw.as<uint64_t>(); w.g_wbase = e->g_wbase.as<uint32_t>(); w.vb_start = e->vb_start.as<uint32_t>();
        HIPCHK(hipMemsetAsync(e->vb_start.p, 0xFF, vb_cap * 4, s));
        hipLaunchKernelGGL(k_group_fill, dim3(cdiv(C, 256)), dim3(256), 0, s, w, C, p.skip_low_complexity_thr, (uint32_t)VB_W, (uint32_t)VB_MINW);
        hipLaunchKernelGGL(k_u64_reduce, dim3(nblk_N), dim3(256), 0, s, (const uint64_t *)w.gw, (const unsigned long long *)&w.si->n_groups, w.scan_part);
        hipLaunchKernelGGL(k_u64_partials, dim3(1), dim3(1024), 0, s, w.scan_part, (const unsigned long long *)&w.si->n_groups, &w.si->vote_weight);
        LAUNCH_EV(k_vote_batches, dim3(nblk_N), dim3(256), s, e->ev[EV_PAIRING], w, (const unsigned long long *)&w.si->n_groups, (const uint64_t *)w.scan_part);
        CANARY("EV_PAIRING");
        if ((rc = read_si(e)) != GCE_OK) return rc;
        HIPCHK(hipGetLastError());
        si_behind_describe = true;
        NG = (uint32_t)e->h_si.n_groups;
    } else HIPCHK(hipEventRecord(e->ev[EV_PAIRING], s));
    const size_t g1 = NG ? NG : 1;
    ENS(gen_list, g1 * 8); ENS(gen_flag, g1 * 2 + 64); ENS(score_list, n1 * 4 + 64);      /* one entry per pair SLOT: a pair may hold one read only (mate absent / far / on another contig), so slots <= N, not N / 2 */ ENS(rp_left, g1 * 4); ENS(rp_right, g1 * 4); ENS(rp_merge, g1 * 4); ENS(rp_rmerge, g1 * 4); ENS(rp_umi, g1 * 8);
    ENS(rp_umilen, g1 * 2); ENS(rp_state, g1); ENS(rp_supp, g1 * 4); ENS(rp_nm, g1 * 8); ENS(rp_qsl, g1 * 4); ENS(rp_qsr, g1 * 4);
    w.gen_list = e->gen_list.as<uint32_t>(); w.gen_flag = e->gen_flag.as<uint8_t>(); w.score_list = e->score_list.as<uint32_t>();
    w.rp_left = e->rp_left.as<uint32_t>(); w.rp_right = e->rp_right.as<uint32_t>();
    w.rp_merge = e->rp_merge.as<uint32_t>(); w.rp_rmerge = e->rp_rmerge.as<uint32_t>(); w.rp_umi = e->rp_umi.as<const char *>();
    w.rp_umilen = e->rp_umilen.as<uint16_t>(); w.rp_state = e->rp_state.as<uint8_t>(); w.rp_supp = e->rp_supp.as<int32_t>();
    w.rp_nm = e->rp_nm.as<int32_t>(); w.rp_qsl = e->rp_qsl.as<uint32_t>(); w.rp_qsr = e->rp_qsr.as<uint32_t>();
    if (NG > 0 && e->dev_error == 0) {
        {   // argument blocks of the generic consensus kernel in device memory (gce_kernels.hpp); here, where the host has just waited for the GPU anyway: `w` is complete
            SlowArgs sa; sa.b = b; sa.p = p; sa.w = w;
            HIPCHK(e->slow_args.ensure(sizeof sa));
            HIPCHK(hipMemcpyAsync(e->slow_args.p, &sa, sizeof sa, hipMemcpyHostToDevice, s));      // (pageable source: staged before the call returns)
        }
        // one launch for the five clears (spatch needs none: k_score2 writes the patch word of both reads of every pair it scores, and only those are read)
        fill_many(s, {FillSeg{e->gen_flag.p, g1 * 2, 0u, 0u},
                      FillSeg{e->rp_nm.p, g1 * 8, 0xFFu, 0u},                                  // -1: NM untouched
                      FillSeg{e->rp_left.p, g1 * 4, 0xFFu, 0u}, FillSeg{e->rp_right.p, g1 * 4, 0xFFu, 0u}});   // NONE: a group no kernel voted on emits nothing (instead of stale read indices)
#ifdef VB_XCD
        const unsigned nbatch = (((unsigned)(e->h_si.vote_weight / VB_W) + 1u) + 7u) & ~7u;      // (vb_start is 0xFF-filled up to vb_cap: the extra blocks find no batch)
#else
        const unsigned nbatch = (unsigned)(e->h_si.vote_weight / VB_W) + 1u;
#endif
#ifdef VB_STOP
        {   // experiment builds (tools/vote_stop.sh): time the truncated k_vote alone and stop -- it leaves garbage behind
            hipEvent_t a_, b_; (void)hipEventCreate(&a_); (void)hipEventCreate(&b_);
            (void)hipEventRecord(a_, s); hipLaunchKernelGGL(k_vote, dim3(nbatch), dim3(VB_T), 0, s, b, p, w, NG); (void)hipEventRecord(b_, s); (void)hipEventSynchronize(b_);
            float ms_ = 0; (void)hipEventElapsedTime(&ms_, a_, b_); fprintf(stderr, "k_vote up to tick %d: %.3f ms\n", VB_STOP, ms_);
            return fail(e, GCE_ERR_INVALID, "experiment build");
        }
#endif
        LAUNCH_EV(k_vote, dim3(nbatch), dim3(VB_T), s, e->ev[EV_SCORE], b, p, w, NG);
        CANARY("EV_SCORE");
        // A stream of deep groups (mean depth beyond 24 pairs: the deep kernels carry the consensus phase, cfg5) runs Pair::computeScore for the
        // handed-on groups (k_score2: bandwidth) on a second HIP stream BESIDE the compaction, the hand-on of the deep sides and their template /
        // voter preparation (k_deep_prepare: a wave per side, latency) -- none of those reads a score or a quality; the votes wait for both.
        const bool deep_stream = ((double)N > 48.0 * (double)NG || getenv("GCE_FORCE_AUX_STREAM")) && !getenv("GCE_NO_AUX_STREAM");          // (reads per group / 2 = mean pairs per group)
        // both kernels stride over lists whose length only the device knows: on a stream of ordinary depth k_vote hands on a few percent of the groups, and a grid sized for
        // "everything" spends its time starting empty blocks (k_consensus_fast: 32 k blocks for ~10 k sides were 20 of its 26 us at cfg3)
        const unsigned cf_cap = deep_stream ? 32768u : 4096u, sc2_cap = deep_stream ? 16384u : 2048u;
        const unsigned cf_grid = std::min<unsigned>(cdiv(2ull * NG, WAVES_PER_BLOCK), cf_cap);
        const unsigned sc2_grid = std::min<unsigned>(cdiv(N, 2 * WAVES_PER_BLOCK * 64), sc2_cap);      // k_score2: waves stride over the list of handed-on pair slots (<= N)
        if (deep_stream) {
            if ((rc = aux_ready(e)) != GCE_OK) return rc;
            HIPCHK(hipEventRecord(e->aux_ev[0], s));                                       // k_vote is done: its slot flags stand, the deep sides are on slow_list
            HIPCHK(hipStreamWaitEvent(e->aux_stream, e->aux_ev[0], 0));
            hipLaunchKernelGGL(k_score2, dim3(sc2_grid), dim3(256), 0, e->aux_stream, b, p, w);
            HIPCHK(hipEventRecord(e->aux_ev[1], e->aux_stream));
            hipLaunchKernelGGL(k_deep_prepare, dim3(1024), dim3(256), 0, s, b, p, w);      // (template + voter list of the deep sides: reads no score, no quality)
            HIPCHK(hipStreamWaitEvent(s, e->aux_ev[1], 0));                                // the scores (and the rewritten qualities) stand
            hipLaunchKernelGGL(k_consensus_fast, dim3(cf_grid), dim3(256), 0, s, b, p, w);           // everything else on gen_list
        } else {
            hipLaunchKernelGGL(k_score2, dim3(sc2_grid), dim3(256), 0, s, b, p, w);       // the handed-on groups only
            hipLaunchKernelGGL(k_deep_prepare, dim3(1024), dim3(256), 0, s, b, p, w);      // (before k_consensus_fast appends the sides IT cannot take: those are not deep)
            hipLaunchKernelGGL(k_consensus_fast, dim3(cf_grid), dim3(256), 0, s, b, p, w);
        }
        hipLaunchKernelGGL(k_vote_deep, dim3(1024), dim3(DV_T), 0, s, b, p, w);                 // deep sides, one block each; leaves what it cannot take
        LAUNCH_EV(k_consensus_slow, dim3(512), dim3(256), s, e->ev[EV_CONSENSUS], (const SlowArgs *)e->slow_args.p);
        CANARY("EV_CONSENSUS");
        hipLaunchKernelGGL(k_group_tail, dim3(cdiv(NG, 256)), dim3(256), 0, s, b, p, w, NG);
        if (!p.disable_duplex) {                                                                  // duplex stage: flagged clusters, compacted, a wave each
            const unsigned nbc = cdiv(C, SCAN_TILE);
            hipLaunchKernelGGL(k_finish_screen, dim3(cdiv(C, 256)), dim3(256), 0, s, w, C, w.pf_flag);
            hipLaunchKernelGGL(k_flag_reduce, dim3(nbc), dim3(256), 0, s, (const uint8_t *)w.pf_flag, (uint64_t)C, w.scan_part);
            hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, w.scan_part, (uint64_t)nbc, &w.si->n_pf_items, (unsigned long long *)nullptr);
            hipLaunchKernelGGL(k_flag_apply, dim3(nbc), dim3(256), 0, s, (const uint8_t *)w.pf_flag, (uint64_t)C, (const uint64_t *)w.scan_part, w.pf_list);
            hipLaunchKernelGGL(k_finish, dim3(4096), dim3(256), 0, s, b, p, w, (const uint32_t *)w.pf_list, (const unsigned long long *)&w.si->n_pf_items);
        }
    } else { HIPCHK(hipEventRecord(e->ev[EV_SCORE], s)); HIPCHK(hipEventRecord(e->ev[EV_CONSENSUS], s)); }
#ifdef VB_PROF
    if (NG > 0) {
        StreamInfo hs; (void)hipStreamSynchronize(s); (void)hipMemcpy(&hs, e->si.p, sizeof hs, hipMemcpyDeviceToHost);
        static const char *nm[11] = {"P0 groups", "P1 pairs", "P2 classes", "P3 overlap", "P4 pass A", "P5 lists", "P5 items", "P5 decide", "P5 tail", "P6 results", "P7 write"};
        const double blocks = (double)hs.prof[15];
        fprintf(stderr, "k_vote phases, mean per block (100 MHz ticks -> us), %.0f blocks:", blocks);
        for (int k = 0; k < 11; k++) fprintf(stderr, " %s %.2f;", nm[k], blocks ? hs.prof[k] / blocks / 100.0 : 0.0);
#ifndef VB_COUNT
        fprintf(stderr, " [of P2: (a) reads vs class %.2f; (b) sides' states %.2f; the rest = order, prefixes, barrier]", blocks ? hs.prof[11] / blocks / 100.0 : 0.0, blocks ? hs.prof[12] / blocks / 100.0 : 0.0);
#endif
        fprintf(stderr, "\n");
#ifdef VB_COUNT
        fprintf(stderr, "k_vote columns in the full vote: %.0f, one base only: %.0f, of those unchanged: %.0f\n", (double)hs.prof[11], (double)hs.prof[12], (double)hs.prof[13]);
#endif
#ifdef DV_PROF
        fprintf(stderr, "k_deep_prepare: %.0f sides, mean %.1f us per side, mean %.0f pairs per side\n", (double)hs.prof[14], hs.prof[14] ? hs.prof[6] / (double)hs.prof[14] / 100.0 : 0.0, hs.prof[14] ? hs.prof[7] / (double)hs.prof[14] / 100.0 : 0.0);
#endif
        static const char *pn[8] = {"windows", "name sort", "pairs", "umi words", "umi sort", "grouping", "layout sort", "layout write"};
        const double cl = (double)hs.prof[30];
        fprintf(stderr, "k_pairing_deep phases, mean per cluster (us), %.0f clusters:", cl);
        for (int k = 0; k < 8; k++) fprintf(stderr, " %s %.1f;", pn[k], cl ? hs.prof[16 + k] / cl / 100.0 : 0.0);
        fprintf(stderr, "\n");
    }
#endif
    HIPCHK(hipEventRecord(e->ev[EV_FINISH], s));
    CANARY("EV_FINISH");
    // ---- the output set: emitted reads in bamComp order (gencore.h:19-47) as one compact table.  Capacities are worst case
    //      (every read emitted): nothing here needs a host round trip.
    OutTable o{};
    if (N > 0 && e->dev_error == 0) {
        // worst case: every read emitted.  The sum of the reads' bytes is at most the blobs' size -- and at most reads x longest read (k_describe's lq_max, on the
        // host since the first look at StreamInfo): on the raw-stream path the "blobs" are the whole inflated file (names, CIGARs and tags included, and the SAME
        // 2.4 GB for bases and qualities), and sizing both outputs by it was 5 of the 9 GB an engine allocated for 8 M reads
        size_t seq_cap = hb.seq_bytes + 16 * n1 + 64, qual_cap = hb.qual_bytes + 16 * n1 + 64;
        if (e->h_si.lq_max >= 0) {
            const size_t lqm = (size_t)e->h_si.lq_max;
            seq_cap = std::min(seq_cap, n1 * ((lqm + 1) / 2 + 16) + 64); qual_cap = std::min(qual_cap, n1 * (lqm + 16) + 64);
        }
        ENS(o_src, n1 * 4); ENS(o_kind, n1); ENS(o_qsrc, n1 * 4); ENS(o_nm, n1 * 4); ENS(o_fr, n1 * 2); ENS(o_rr, n1 * 2); ENS(o_mate, n1 * 4);
        ENS(o_soff, n1 * 8); ENS(o_qoff, n1 * 8); ENS(o_seq, seq_cap); ENS(o_qual, qual_cap);
        ENS(o_key, n1 * sizeof(OutKey)); ENS(o_rec, n1 * sizeof(OutRec)); ENS(o_ksoff, n1 * 8); ENS(o_kqoff, n1 * 8); ENS(o_krow, n1 * 4); ENS(o_rank64, (n1 / 64 + 2) * 4); const unsigned nblk_O = cdiv(n1, OUT_TILE); ENS(o_part3, (size_t)nblk_O * 24 + 64);
        o.src = e->o_src.as<uint32_t>(); o.kind = e->o_kind.as<uint8_t>(); o.qname_src = e->o_qsrc.as<uint32_t>(); o.nm_new = e->o_nm.as<int32_t>();
        o.fr = e->o_fr.as<int16_t>(); o.rr = e->o_rr.as<int16_t>(); o.mate = e->o_mate.as<uint32_t>();
        o.seq_off = e->o_soff.as<uint64_t>(); o.qual_off = e->o_qoff.as<uint64_t>(); o.seq = e->o_seq.as<uint8_t>(); o.qual = e->o_qual.as<uint8_t>();
        o.key = e->o_key.as<OutKey>(); o.rec = e->o_rec.as<OutRec>(); o.ksoff = e->o_ksoff.as<uint64_t>(); o.kqoff = e->o_kqoff.as<uint64_t>(); o.krow = e->o_krow.as<uint32_t>(); o.rank64 = e->o_rank64.as<uint32_t>(); o.part3 = e->o_part3.as<uint64_t>();
        hipLaunchKernelGGL(k_stats, dim3(1024), dim3(256), 0, s, b, w, (NG > 0 ? C : 0u), NG);     // Stats: clusters, groups  (round 5: on the second stream beside the output kernels it hid its 50 us and stretched k_out_reduce / k_out_partials by as much: not kept)
        CANARY("k_stats");
        hipLaunchKernelGGL(k_out_reduce, dim3(nblk_O), dim3(OUT_T), 0, s, b, w, o);
        CANARY("k_out_reduce");
        hipLaunchKernelGGL(k_out_partials, dim3(1), dim3(1024), 0, s, o, (uint64_t)nblk_O, w);
        CANARY("k_out_partials");
        // every emitted read of one length (k_describe's range, final on the host once the look behind the pairing phase has happened): no offset lists in LDS
        const bool lq_uniform = si_behind_describe && e->h_si.lq_max >= 0 && e->h_si.lq_min == e->h_si.lq_max;
        if (lq_uniform) hipLaunchKernelGGL(k_out_meta<true>, dim3(nblk_O), dim3(OUT_T), 0, s, b, w, o, (int)e->h_si.lq_max);
        else hipLaunchKernelGGL(k_out_meta<false>, dim3(nblk_O), dim3(OUT_T), 0, s, b, w, o, -1);
        CANARY("k_out_meta");
        const unsigned og = std::min<unsigned>(cdiv(n1, 256), 8192u);
        hipLaunchKernelGGL(k_out_rows, dim3(og), dim3(256), 0, s, w, o);
        CANARY("k_out_rows");
        hipLaunchKernelGGL(k_out_mate, dim3(og), dim3(256), 0, s, w, o);
        CANARY("k_out_mate");
        hipLaunchKernelGGL(k_out_gather, dim3(std::min<unsigned>(cdiv(n1, 16), 16384u)), dim3(256), 0, s, b, w, o);
        CANARY("k_out_gather");
    }
    HIPCHK(hipEventRecord(e->ev[EV_OUTPUT], s));
    CANARY("EV_OUTPUT");
    hipLaunchKernelGGL(k_fold_stats, dim3(1), dim3(64), 0, s, w.si, pre_extra);          // (outside the timed step: a convenience of gce_stats_device)
    if ((rc = read_si(e)) != GCE_OK) return rc;
    HIPCHK(hipGetLastError());
#ifdef GCE_SI_CHECK       // debugging build: the post block's histogram only ever has entry 1 (outputPair: addMolecule(1, PE)) -- anything else is a stray write
    {
        bool badw = false;
        for (int k = 14; k < GCE_STATS_WORDS; k++) if ((k != 15 && e->h_si.post[k] != 0) || e->h_si.pre[k] < 0) badw = true;
        if (badw) {
            StreamInfo again; (void)hipMemcpy(&again, e->si.p, sizeof again, hipMemcpyDeviceToHost);
            fprintf(stderr, "SI_CHECK: N %lld NG %u C %u n_out %llu\n post:", (long long)N, NG, C, (unsigned long long)e->h_si.n_out);
            for (int k = 0; k < 30; k++) fprintf(stderr, " %lld", e->h_si.post[k]);
            fprintf(stderr, "\n post again:"); for (int k = 0; k < 30; k++) fprintf(stderr, " %lld", again.post[k]);
            fprintf(stderr, "\n pre:"); for (int k = 0; k < 30; k++) fprintf(stderr, " %lld", e->h_si.pre[k]);
            fprintf(stderr, "\n other nonzero post words:"); for (int k = 30; k < GCE_STATS_WORDS; k++) if (e->h_si.post[k]) fprintf(stderr, " [%d]=%lld", k, e->h_si.post[k]);
            fprintf(stderr, "\n post_slot col 0..7 sums:"); for (int k = 0; k < 8; k++) { long long a = 0; for (int q = 0; q < GCE_PRE_SLOTS; q++) a += e->h_si.post_slot[q][k]; fprintf(stderr, " %lld", a); }
            fprintf(stderr, "\n scalars: n_clustered %llu n_slow %u n_deep %u n_slow_pair %u n_gen %llu n_pf %llu n_pq %llu vote_weight %llu\n", e->h_si.n_clustered, e->h_si.n_slow, e->h_si.n_deep, e->h_si.n_slow_pair, (unsigned long long)(e->h_si.hand_on >> 32), e->h_si.n_pf_items, e->h_si.n_pq_items, e->h_si.vote_weight);
        }
    }
#endif
    e->h_si.n_groups = NG;
    float ms = 0;
    auto el = [&](int a, int c) { ms = 0; (void)hipEventElapsedTime(&ms, e->ev[a], e->ev[c]); return (double)ms; };
    e->timing.total_ms = el(EV_START, EV_OUTPUT);
    e->timing.cluster_ms = el(EV_START, EV_CLUSTER); e->timing.csr_ms = el(EV_CLUSTER, EV_CSR); e->timing.describe_ms = el(EV_CSR, EV_DESCRIBE);
    e->timing.pairing_ms = el(EV_DESCRIBE, EV_PAIRING); e->timing.score_ms = el(EV_PAIRING, EV_SCORE); e->timing.consensus_ms = el(EV_SCORE, EV_CONSENSUS);
    e->timing.finish_ms = el(EV_CONSENSUS, EV_FINISH); e->timing.output_ms = el(EV_FINISH, EV_OUTPUT);
    e->timing.n_clusters = C; e->timing.n_groups = NG;
    if (e->dev_error != 0) {
        char buf[160]; snprintf(buf, sizeof buf, "%s (read %u)", gce_status_message(e->dev_error), e->dev_error_read);
        return fail(e, e->dev_error, buf);
    }
    e->n_out = (int64_t)e->h_si.n_out;
    e->out_seq_bytes = (size_t)(e->h_si.out_units >> 32) * 16; e->out_qual_bytes = (size_t)(e->h_si.out_units & 0xFFFFFFFFull) * 16;
    e->timing.n_pairs = (int64_t)e->h_si.n_pairs_total; e->timing.n_leaders = (int64_t)e->h_si.n_leaders;
    return GCE_OK;
}

static void fill_stats(gce_stats *dst, const long long *src) { memcpy(dst, src, sizeof(gce_stats)); }

int gce_result_device(gce_engine *e, gce_result *out) {
    if (!e || !out || !e->processed || e->dev_error) return GCE_ERR_INVALID;
    memset(out, 0, sizeof *out);
    out->n_reads = e->n; out->n_out = e->n_out;
    out->src = e->o_src.as<uint32_t>(); out->kind = e->o_kind.as<uint8_t>(); out->qname_src = e->o_qsrc.as<uint32_t>(); out->nm_new = e->o_nm.as<int32_t>();
    out->fr = e->o_fr.as<int16_t>(); out->rr = e->o_rr.as<int16_t>(); out->mate = e->o_mate.as<uint32_t>();
    out->seq_off = e->o_soff.as<uint64_t>(); out->qual_off = e->o_qoff.as<uint64_t>(); out->seq = e->o_seq.as<uint8_t>(); out->qual = e->o_qual.as<uint8_t>();
    out->seq_bytes = e->out_seq_bytes; out->qual_bytes = e->out_qual_bytes;
    fill_stats(&out->pre, e->h_si.pre); fill_stats(&out->post, e->h_si.post);
    return GCE_OK;
}

// Draining csPairs into Gencore::outputPair (src/gencore.cpp:356-360,410-414): the table of emitted records copied to host memory.
int gce_drain(gce_engine *e, gce_result *out) {
    if (!e || !out || !e->processed || e->dev_error) return GCE_ERR_INVALID;
    (void)hipSetDevice(e->prm.device);
    const size_t n = (size_t)e->n_out;
    e->r_src.resize(n); e->r_kind.resize(n); e->r_qsrc.resize(n); e->r_nm.resize(n); e->r_fr.resize(n); e->r_rr.resize(n); e->r_mate.resize(n);
    e->r_soff.resize(n); e->r_qoff.resize(n); e->r_seq.resize(e->out_seq_bytes); e->r_qual.resize(e->out_qual_bytes);
    if (e->r_seq.size() != e->out_seq_bytes || e->r_qual.size() != e->out_qual_bytes) return GCE_ERR_OOM;
    hipStream_t s = e->stream;
    if (n) {
        HIPCHK(hipMemcpyAsync(e->r_src.data(), e->o_src.p, n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(e->r_kind.data(), e->o_kind.p, n, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(e->r_qsrc.data(), e->o_qsrc.p, n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(e->r_nm.data(), e->o_nm.p, n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(e->r_fr.data(), e->o_fr.p, n * 2, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(e->r_rr.data(), e->o_rr.p, n * 2, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(e->r_mate.data(), e->o_mate.p, n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(e->r_soff.data(), e->o_soff.p, n * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(e->r_qoff.data(), e->o_qoff.p, n * 8, hipMemcpyDeviceToHost, s));
        if (!e->r_seq.empty()) HIPCHK(hipMemcpyAsync(e->r_seq.data(), e->o_seq.p, e->r_seq.size(), hipMemcpyDeviceToHost, s));
        if (!e->r_qual.empty()) HIPCHK(hipMemcpyAsync(e->r_qual.data(), e->o_qual.p, e->r_qual.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    memset(out, 0, sizeof *out);
    out->n_reads = e->n; out->n_out = e->n_out;
    out->src = e->r_src.data(); out->kind = e->r_kind.data(); out->qname_src = e->r_qsrc.data(); out->nm_new = e->r_nm.data();
    out->fr = e->r_fr.data(); out->rr = e->r_rr.data(); out->mate = e->r_mate.data();
    out->seq_off = e->r_soff.data(); out->qual_off = e->r_qoff.data(); out->seq = e->r_seq.data(); out->qual = e->r_qual.data();
    out->seq_bytes = e->out_seq_bytes; out->qual_bytes = e->out_qual_bytes;
#ifdef GCE_SI_CHECK
    for (int k = 14; k < GCE_STATS_WORDS; k++) if (k != 15 && e->h_si.post[k] != 0) fprintf(stderr, "SI_CHECK(gce_drain): the engine's host copy has post[%d] = %lld\n", k, e->h_si.post[k]);
#endif
    fill_stats(&out->pre, e->h_si.pre); fill_stats(&out->post, e->h_si.post);
    return GCE_OK;
}

// Stats::statDepth / Bed::statDepth for the processed stream: see gce_depth.hpp and include/gencore_amd.h.
// BED counts from the kernel's order (regions grouped by contig) to the order of the file, and the two Stats blocks in front of the payload
__global__ void k_payload_finish(const StreamInfo *si, const int32_t *where, int32_t n_regions, const unsigned long long *bed_grouped, int32_t nreg, long long *payload, int64_t nbins) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * GCE_STATS_WORDS) payload[i] = i < GCE_STATS_WORDS ? si->pre[i] : si->post[i - GCE_STATS_WORDS];
    if (i < n_regions) {
        const int w = where[i];
        long long *bed = payload + 2 * GCE_STATS_WORDS + 2 * nbins;
        bed[i] = w >= 0 ? (long long)bed_grouped[w] : 0; bed[n_regions + i] = w >= 0 ? (long long)bed_grouped[nreg + w] : 0;
    }
}

// Everything the final Stats merge of a multi-GPU run adds up (stats.cpp:39-46,56-83, bed.cpp:64-79, gencore.cpp:181-184) as ONE buffer of int64 in
// device memory: [pre Stats][post Stats][pre depth bins][post depth bins][pre BED counts][post BED counts] -- all additive over key-range shards, so a
// multi-GPU run merges it with one all-reduce(sum) (bench.py) or one add per engine (gce_raw_merge_outputs).  See include/gencore_amd.h.
int gce_stats_payload_device(gce_engine *e, int32_t step, int32_t n_regions, const int32_t *r_tid, const int32_t *r_start, const int32_t *r_end, const int64_t **payload, gce_payload_layout *layout) {
    if (!e || !payload || !layout || step <= 0 || n_regions < 0 || (n_regions > 0 && (!r_tid || !r_start || !r_end))) return GCE_ERR_INVALID;
    if (!e->processed || e->dev_error) return fail(e, GCE_ERR_INVALID, "gce_stats_payload_device before a successful gce_process");
    (void)hipSetDevice(e->prm.device);
    hipStream_t s = e->stream;
    const int nt = (int)e->target_len.size();
    e->h_binoff.assign(nt + 1, 0);
    for (int t = 0; t < nt; t++) e->h_binoff[t + 1] = e->h_binoff[t] + 1 + (int64_t)e->target_len[t] / step;           // stats.cpp:41-47
    const int64_t nbins = e->h_binoff[nt];
    // regions grouped by contig, file order kept (Bed::loadFromFile pushes them per contig, bed.cpp:165-166)
    std::vector<int32_t> off(nt + 1, 0), rs(n_regions), re(n_regions), pm(n_regions), where(n_regions);
    for (int k = 0; k < n_regions; k++) if (r_tid[k] >= 0 && r_tid[k] < nt) off[r_tid[k] + 1]++;
    for (int t = 0; t < nt; t++) off[t + 1] += off[t];
    std::vector<int32_t> fill(off.begin(), off.end() - 1);
    std::vector<uint8_t> sorted(std::max(nt, 1), 1);
    std::fill(where.begin(), where.end(), -1);
    for (int k = 0; k < n_regions; k++) {
        if (r_tid[k] < 0 || r_tid[k] >= nt) continue;                                                                  // contig not in the BAM header: dropped (bed.cpp:151-166)
        const int t = r_tid[k], at = fill[t]++;
        rs[at] = r_start[k]; re[at] = r_end[k]; where[k] = at;
        pm[at] = at > off[t] ? std::max(pm[at - 1], r_end[k]) : r_end[k];
        if (at > off[t] && rs[at] < rs[at - 1]) sorted[t] = 0;
    }
    const int nreg = off[nt];
    auto up = [&](DevBuf &d, const void *src, size_t bytes) -> int { HIPCHK(d.ensure(bytes + 64)); if (bytes) HIPCHK(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, s)); return GCE_OK; };
    int rc;
    if ((rc = up(e->dp_binoff, e->h_binoff.data(), (nt + 1) * 8)) || (rc = up(e->dp_regoff, off.data(), (nt + 1) * 4)) || (rc = up(e->dp_rs, rs.data(), (size_t)nreg * 4)) ||
        (rc = up(e->dp_re, re.data(), (size_t)nreg * 4)) || (rc = up(e->dp_pmax, pm.data(), (size_t)nreg * 4)) || (rc = up(e->dp_sorted, sorted.data(), sorted.size())) ||
        (rc = up(e->dp_where, where.data(), (size_t)n_regions * 4))) return rc;
    const int64_t words = 2 * GCE_STATS_WORDS + 2 * nbins + 2 * (int64_t)n_regions;
    HIPCHK(e->dp_depth.ensure((size_t)words * 8 + 64)); HIPCHK(e->dp_bed.ensure((size_t)nreg * 16 + 64));
    HIPCHK(hipMemsetAsync(e->dp_depth.p, 0, (size_t)words * 8, s)); HIPCHK(hipMemsetAsync(e->dp_bed.p, 0, (size_t)nreg * 16 + 16, s));
    DepthCtx c; c.bin_off = e->dp_binoff.as<int64_t>(); c.n_targets = nt; c.step = step; c.reg_off = e->dp_regoff.as<int32_t>();
    c.r_start = e->dp_rs.as<int32_t>(); c.r_end = e->dp_re.as<int32_t>(); c.r_pmax = e->dp_pmax.as<int32_t>(); c.contig_sorted = e->dp_sorted.as<uint8_t>();
    long long *pay = e->dp_depth.as<long long>();
    unsigned long long *dpre = (unsigned long long *)pay + 2 * GCE_STATS_WORDS, *dpost = dpre + nbins, *bpre = e->dp_bed.as<unsigned long long>(), *bpost = bpre + nreg;
    const uint64_t n = (uint64_t)e->n_pre, no = (uint64_t)e->n_out;
    if (n) hipLaunchKernelGGL(k_depth, dim3(cdiv(n, DP_T * DP_RPT)), dim3(DP_T), 0, s, e->dev_batch.core, (const uint32_t *)nullptr, n, c, dpre, bpre);
    if (no) hipLaunchKernelGGL(k_depth, dim3(cdiv(no, DP_T * DP_RPT)), dim3(DP_T), 0, s, e->dev_batch.core, (const uint32_t *)e->o_src.p, no, c, dpost, bpost);
    const int fin = std::max(2 * GCE_STATS_WORDS, (int)n_regions);
    hipLaunchKernelGGL(k_payload_finish, dim3(cdiv(fin, 256)), dim3(256), 0, s, (const StreamInfo *)e->si.p, (const int32_t *)e->dp_where.p, n_regions, (const unsigned long long *)e->dp_bed.p, nreg, pay, nbins);
    HIPCHK(hipGetLastError());
    // the payload is complete when the call returns: a caller that reads it on another stream -- RCCL on its own stream, a torch side stream -- needs no
    // knowledge of this engine's stream.  One host wait per call (the kernels above take ~0.4 ms at cfg3); gce_stats_payload_sum / _read order themselves as before.
    HIPCHK(hipStreamSynchronize(s));
    layout->stats_words = 2 * GCE_STATS_WORDS; layout->n_targets = nt; layout->n_bins = nbins; layout->n_regions = n_regions; layout->total_words = words; layout->bin_off = e->h_binoff.data();
    e->payload_words = words;
    *payload = (const int64_t *)pay;
    return GCE_OK;
}

// the payload (or the sum gce_stats_payload_sum left on this engine's device) on the host
int gce_stats_payload_read(gce_engine *e, const int64_t *payload, int64_t n_words, int64_t *host) {
    if (!e || !payload || !host || n_words < 0) return GCE_ERR_INVALID;
    (void)hipSetDevice(e->prm.device);
    if (n_words) HIPCHK(hipMemcpyAsync(host, payload, (size_t)n_words * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return GCE_OK;
}

int gce_depth_stats(gce_engine *e, int32_t step, int32_t n_regions, const int32_t *r_tid, const int32_t *r_start, const int32_t *r_end, gce_depth *out) {
    if (!e || !out) return GCE_ERR_INVALID;
    const int64_t *pay = nullptr; gce_payload_layout lay;
    int rc = gce_stats_payload_device(e, step, n_regions, r_tid, r_start, r_end, &pay, &lay);
    if (rc != GCE_OK) return rc;
    hipStream_t s = e->stream;
    const int64_t nbins = lay.n_bins;
    e->h_depth_pre.resize(nbins); e->h_depth_post.resize(nbins); e->h_bed_pre.assign(n_regions, 0); e->h_bed_post.assign(n_regions, 0);
    const int64_t *d0 = pay + lay.stats_words;
    if (nbins) { HIPCHK(hipMemcpyAsync(e->h_depth_pre.data(), d0, nbins * 8, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(e->h_depth_post.data(), d0 + nbins, nbins * 8, hipMemcpyDeviceToHost, s)); }
    if (n_regions) { HIPCHK(hipMemcpyAsync(e->h_bed_pre.data(), d0 + 2 * nbins, (size_t)n_regions * 8, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(e->h_bed_post.data(), d0 + 2 * nbins + n_regions, (size_t)n_regions * 8, hipMemcpyDeviceToHost, s)); }
    HIPCHK(hipStreamSynchronize(s));
    out->n_targets = lay.n_targets; out->bin_off = e->h_binoff.data(); out->pre_depth = e->h_depth_pre.data(); out->post_depth = e->h_depth_post.data();
    out->n_regions = n_regions; out->pre_bed = e->h_bed_pre.data(); out->post_bed = e->h_bed_post.data();
    return GCE_OK;
}

// the Stats blocks of the last gce_process on the DEVICE: 2 x GCE_STATS_WORDS int64 (pre, then post), complete (the spread addRead
// counters are folded in), for a collective that should not bounce through the host (one RCCL all-reduce)
int gce_stats_device(gce_engine *e, const int64_t **pre_then_post) {
    if (!e || !pre_then_post || !e->processed || e->dev_error) return GCE_ERR_INVALID;
    *pre_then_post = (const int64_t *)e->si.as<StreamInfo>()->pre;
    return GCE_OK;
}

int gce_get_timing(gce_engine *e, gce_timing *out) {
    if (!e || !out) return GCE_ERR_INVALID;
    *out = e->timing;
    return GCE_OK;
}

}  // extern "C"

#include "gce_plan.hpp"
#include "gce_bamdev.hpp"
