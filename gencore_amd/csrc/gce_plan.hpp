// gce_plan.hpp — the multi-GPU planner in the C-ABI: what a key-range shard of ONE sorted stream has to know about the
// rest of the stream, computed on a GPU from the 32-byte key records alone.
//   gce_stream_context   per read the value of the reference's `tick` right after the read was added (gencore.cpp:319-320), and the
//                        flush events: the reads on which tick % period == 0 (gencore.cpp:321-322) with their (tid, pos)
//   gce_plan_shards      the shard of every read.  All reads of one cluster key (tid, left) get the same shard -- a right mate follows
//                        its mate's position (gencore.cpp:301-303), so a shard's reads interleave with its neighbours' in stream order.
//                        mode 0: contiguous key ranges balanced by read count (one radix sort of the keys, the cuts are its quantiles);
//                        mode 1: clusters dealt to the least loaded shard, heaviest first, weight = reads^2 (ultra-deep hotspots)
// The spec is gencore_amd/shard.py (stream_context, plan_shards); tests compare the two.  Scans, sort, run-length encoding and
// compaction are rocPRIM's (through hipCUB): plain library primitives; the keys and flags come from the kernels below.
#pragma once
#include <hipcub/hipcub.hpp>

namespace {

__global__ __launch_bounds__(256) void k_plan_keys(const gce_core *core, int64_t n, unsigned long long *tick, unsigned long long *key, unsigned int *first_unm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    union { gce_core c; uint4 q[2]; } t; const uint4 *src = reinterpret_cast<const uint4 *>(core + i); t.q[0] = src[0]; t.q[1] = src[1];
    const gce_core c = t.c;
    if (tick) tick[i] = d_classify(c) == CLS_CLUSTERED ? 1ull : 0ull;
    if (first_unm && (c.tid < 0 || c.pos < 0) && (unsigned int)i < *(volatile unsigned int *)first_unm) atomicMin(first_unm, (unsigned int)i);
    if (key) {                                                      // (tid, left) of the cluster key, unmapped reads last (shard.py: plan_shards)
        long long d = (long long)c.mpos - (long long)c.pos; if (d < 0) d = -d;
        const bool near = c.mtid == c.tid && d < 100000;
        long long left = (near && c.isize < 0) ? c.mpos : c.pos; if (left < 0) left = 0;
        const unsigned long long tq = c.tid < 0 ? (1ull << 30) : (unsigned long long)c.tid;
        key[i] = (tq << 32) | (unsigned long long)left;
    }
}
// event flag of every read: clustered, tick a multiple of the period, in front of the first unmapped read; `bad`: a clustered read behind it
__global__ __launch_bounds__(256) void k_plan_events(const gce_core *core, int64_t n, const unsigned long long *tick, unsigned long long period, const unsigned int *first_unm, uint8_t *flag, int *bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool cm = tick[i] != (i ? tick[i - 1] : 0ull);
    const bool front = (unsigned int)i < *first_unm;
    flag[i] = cm && front && tick[i] % period == 0;
    if (cm && !front) *bad = 1;
}
__global__ __launch_bounds__(256) void k_plan_event_pos(const gce_core *core, const int64_t *idx, int n_ev, int32_t *ev_tid, int32_t *ev_pos) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_ev) { ev_tid[j] = core[idx[j]].tid; ev_pos[j] = core[idx[j]].pos; }
}
struct PlanCuts { unsigned long long c[63]; int n; };
__global__ __launch_bounds__(256) void k_plan_range(const unsigned long long *key, int64_t n, PlanCuts cuts, int32_t *shard) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = key[i];
    int s = 0;
    for (int q = 0; q < cuts.n; q++) s += cuts.c[q] <= k;           // searchsorted(cuts, key, side = "right")
    shard[i] = s;
}
__global__ __launch_bounds__(256) void k_plan_owner(const unsigned long long *key, int64_t n, const unsigned long long *uniq, const int32_t *owner, int64_t n_uniq, int32_t *shard) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = key[i];
    int64_t lo = 0, hi = n_uniq;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (uniq[mid] < k) lo = mid + 1; else hi = mid; }
    shard[i] = owner[lo];
}

struct PlanBuf {                                                    // a device buffer for the life of one call
    void *p = nullptr;
    ~PlanBuf() { if (p) (void)hipFree(p); }
    hipError_t get(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T> T *as() { return (T *)p; }
};
inline bool plan_is_device(const void *p) {
    hipPointerAttribute_t attr;
    const bool dev = hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    return dev;
}
#define PLCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return _e == hipErrorOutOfMemory ? GCE_ERR_OOM : GCE_ERR_HIP; } while (0)

}  // namespace

extern "C" {

void gce_free(void *p) { free(p); }

int gce_stream_context(int32_t device, const gce_core *core, int64_t n, int32_t flush_period, uint64_t *tick_out, int32_t *n_events, int32_t **ev_tid, int32_t **ev_pos) {
    if (n < 0 || (n > 0 && (!core || !tick_out)) || flush_period <= 0 || !n_events || !ev_tid || !ev_pos || n >= (int64_t)0x7FFFFFF0ll) return GCE_ERR_INVALID;   // (the library primitives take 32-bit counts)
    *n_events = 0; *ev_tid = nullptr; *ev_pos = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return GCE_ERR_NO_DEVICE;
    if (n == 0) return GCE_OK;
    PLCHK(hipSetDevice(device));
    PlanBuf b_core, b_tick, b_flag, b_idx, b_tmp, b_misc, b_et, b_ep;
    const gce_core *dcore = core;
    if (!plan_is_device(core)) { PLCHK(b_core.get((size_t)n * sizeof(gce_core))); PLCHK(hipMemcpy(b_core.p, core, (size_t)n * sizeof(gce_core), hipMemcpyHostToDevice)); dcore = b_core.as<gce_core>(); }
    const bool tick_dev = plan_is_device(tick_out);
    unsigned long long *dtick = (unsigned long long *)tick_out;
    if (!tick_dev) { PLCHK(b_tick.get((size_t)n * 8)); dtick = b_tick.as<unsigned long long>(); }
    PLCHK(b_misc.get(64));                                           // [0] first unmapped read, [1] bad flag, [2] number of events (int64 at +8)
    unsigned int init[4] = {NONE32, 0u, 0u, 0u};
    PLCHK(hipMemcpy(b_misc.p, init, sizeof init, hipMemcpyHostToDevice));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_plan_keys, dim3(nb), dim3(256), 0, 0, dcore, n, dtick, (unsigned long long *)nullptr, b_misc.as<unsigned int>());
    size_t tmp_bytes = 0, tmp2 = 0;
    PLCHK(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, dtick, dtick, (int)n));
    PLCHK(b_flag.get((size_t)n)); PLCHK(b_idx.get((size_t)n * 8));
    PLCHK(hipcub::DeviceSelect::Flagged(nullptr, tmp2, hipcub::CountingInputIterator<int64_t>(0), b_flag.as<uint8_t>(), b_idx.as<int64_t>(), (int64_t *)((char *)b_misc.p + 8), (int)n));
    PLCHK(b_tmp.get(std::max(tmp_bytes, tmp2)));
    PLCHK(hipcub::DeviceScan::InclusiveSum(b_tmp.p, tmp_bytes, dtick, dtick, (int)n));
    hipLaunchKernelGGL(k_plan_events, dim3(nb), dim3(256), 0, 0, dcore, n, (const unsigned long long *)dtick, (unsigned long long)flush_period, (const unsigned int *)b_misc.p, b_flag.as<uint8_t>(), (int *)b_misc.p + 1);
    PLCHK(hipcub::DeviceSelect::Flagged(b_tmp.p, tmp2, hipcub::CountingInputIterator<int64_t>(0), b_flag.as<uint8_t>(), b_idx.as<int64_t>(), (int64_t *)((char *)b_misc.p + 8), (int)n));
    struct { unsigned int first_unm; int bad; int64_t n_ev; } h;
    PLCHK(hipMemcpy(&h, b_misc.p, sizeof h, hipMemcpyDeviceToHost));
    if (h.bad) return GCE_ERR_INVALID;                               // key-range shards need every mapped read in front of the first unmapped read (shard.py)
    if (!tick_dev) PLCHK(hipMemcpy(tick_out, dtick, (size_t)n * 8, hipMemcpyDeviceToHost));
    const int ne = (int)h.n_ev;
    if (ne > 0) {
        PLCHK(b_et.get((size_t)ne * 4)); PLCHK(b_ep.get((size_t)ne * 4));
        hipLaunchKernelGGL(k_plan_event_pos, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, 0, dcore, (const int64_t *)b_idx.p, ne, b_et.as<int32_t>(), b_ep.as<int32_t>());
        *ev_tid = (int32_t *)malloc((size_t)ne * 4); *ev_pos = (int32_t *)malloc((size_t)ne * 4);
        if (!*ev_tid || !*ev_pos) { free(*ev_tid); free(*ev_pos); *ev_tid = *ev_pos = nullptr; return GCE_ERR_OOM; }
        PLCHK(hipMemcpy(*ev_tid, b_et.p, (size_t)ne * 4, hipMemcpyDeviceToHost)); PLCHK(hipMemcpy(*ev_pos, b_ep.p, (size_t)ne * 4, hipMemcpyDeviceToHost));
    }
    *n_events = ne;
    PLCHK(hipDeviceSynchronize());
    return GCE_OK;
}

int gce_plan_shards(int32_t device, const gce_core *core, int64_t n, int32_t world, int32_t mode, int32_t *shard_out) {
    if (n < 0 || (n > 0 && (!core || !shard_out)) || world < 1 || world > 64 || (mode != 0 && mode != 1) || n >= (int64_t)0x7FFFFFF0ll) return GCE_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return GCE_ERR_NO_DEVICE;
    if (n == 0) return GCE_OK;
    PLCHK(hipSetDevice(device));
    PlanBuf b_core, b_key, b_srt, b_tmp, b_shard, b_uniq, b_cnt, b_nrun, b_owner;
    const gce_core *dcore = core;
    if (!plan_is_device(core)) { PLCHK(b_core.get((size_t)n * sizeof(gce_core))); PLCHK(hipMemcpy(b_core.p, core, (size_t)n * sizeof(gce_core), hipMemcpyHostToDevice)); dcore = b_core.as<gce_core>(); }
    const bool out_dev = plan_is_device(shard_out);
    int32_t *dshard = shard_out;
    if (!out_dev) { PLCHK(b_shard.get((size_t)n * 4)); dshard = b_shard.as<int32_t>(); }
    PLCHK(b_key.get((size_t)n * 8)); PLCHK(b_srt.get((size_t)n * 8));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_plan_keys, dim3(nb), dim3(256), 0, 0, dcore, n, (unsigned long long *)nullptr, b_key.as<unsigned long long>(), (unsigned int *)nullptr);
    size_t tmp_bytes = 0;
    PLCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, b_key.as<unsigned long long>(), b_srt.as<unsigned long long>(), (int)n, 0, 63));
    PLCHK(b_tmp.get(tmp_bytes));
    PLCHK(hipcub::DeviceRadixSort::SortKeys(b_tmp.p, tmp_bytes, b_key.as<unsigned long long>(), b_srt.as<unsigned long long>(), (int)n, 0, 63));
    if (mode == 0) {
        PlanCuts cuts; cuts.n = world - 1;
        for (int r = 1; r < world; r++) {                            // srt[min(n - 1, n * r / world)]
            const int64_t at = std::min<int64_t>(n - 1, (int64_t)((__int128)n * r / world));
            PLCHK(hipMemcpy(&cuts.c[r - 1], b_srt.as<unsigned long long>() + at, 8, hipMemcpyDeviceToHost));
        }
        hipLaunchKernelGGL(k_plan_range, dim3(nb), dim3(256), 0, 0, (const unsigned long long *)b_key.p, n, cuts, dshard);
    } else {
        PLCHK(b_uniq.get((size_t)n * 8)); PLCHK(b_cnt.get((size_t)n * 4)); PLCHK(b_nrun.get(16));
        size_t t2 = 0;
        PLCHK(hipcub::DeviceRunLengthEncode::Encode(nullptr, t2, b_srt.as<unsigned long long>(), b_uniq.as<unsigned long long>(), b_cnt.as<int>(), b_nrun.as<int>(), (int)n));
        PlanBuf b_t2; PLCHK(b_t2.get(t2));
        PLCHK(hipcub::DeviceRunLengthEncode::Encode(b_t2.p, t2, b_srt.as<unsigned long long>(), b_uniq.as<unsigned long long>(), b_cnt.as<int>(), b_nrun.as<int>(), (int)n));
        int nrun = 0;
        PLCHK(hipMemcpy(&nrun, b_nrun.p, 4, hipMemcpyDeviceToHost));
        std::vector<int> cnt((size_t)nrun); std::vector<int32_t> owner((size_t)nrun), order((size_t)nrun);
        PLCHK(hipMemcpy(cnt.data(), b_cnt.p, (size_t)nrun * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < nrun; k++) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t c) { return (double)cnt[a] * cnt[a] > (double)cnt[c] * cnt[c]; });   // argsort(-cnt^2, stable)
        std::vector<double> load((size_t)world, 0.0);
        for (int32_t c : order) {                                    // longest processing time first: the least loaded shard, lowest number on ties (np.argmin)
            int r = 0;
            for (int q = 1; q < world; q++) if (load[q] < load[r]) r = q;
            owner[c] = r; load[r] += (double)cnt[c] * (double)cnt[c];
        }
        PLCHK(b_owner.get((size_t)nrun * 4));
        PLCHK(hipMemcpy(b_owner.p, owner.data(), (size_t)nrun * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_plan_owner, dim3(nb), dim3(256), 0, 0, (const unsigned long long *)b_key.p, n, (const unsigned long long *)b_uniq.p, (const int32_t *)b_owner.p, (int64_t)nrun, dshard);
    }
    if (!out_dev) PLCHK(hipMemcpy(shard_out, dshard, (size_t)n * 4, hipMemcpyDeviceToHost));
    PLCHK(hipDeviceSynchronize());
    return GCE_OK;
}

}  // extern "C"
