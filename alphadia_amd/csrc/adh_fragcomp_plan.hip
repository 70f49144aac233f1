// adh_fragcomp_plan.hip - what FragmentCompetition.__call__ derives before it competes, on the device (round 5).
//
// The reference prepares the competition with pandas (fragcomp.py:170-229,268-289; fragcomp/utils.py:11-58): a candidate
// key per PSM and per fragment row, the fragment range of every PSM (first row with its key to last such row + 1; PSMs
// without fragment rows leave), the DIA window of every PSM (first cycle row whose [lowest, highest) isolation limit holds
// its observed m/z, row 0 when none does), and the processing order: window by window, lowest `proba` first, ties by
// precursor_idx, then by input position (a stable multi-column sort).  alphadia_amd/fragcomp.py::competition_plan does
// the same in NumPy: 360-420 ms per 1e6 PSMs with 12 M fragment rows on the GPU boxes' host, 150 x the competition
// kernels (2.4 ms).  Here the columns go up as they are and the plan is a handful of kernels, scans and radix sorts:
//   * runs of equal keys in the fragment table (a candidate's rows follow each other in the table collect_fragments
//     writes): run starts by a scan, the runs' keys sorted (1 key per candidate, not per row).  A key that starts two
//     runs means the table is not grouped: the call says so (`grouped` = 0) and the caller takes the NumPy plan
//   * per PSM: binary search of its key among the runs -> fragment range; the window by a walk over the cycle rows
//   * the order by two stable radix sorts over the PSMs that have fragments: by precursor_idx, then by
//     (window << 32 | proba as an order-preserving 32-bit pattern)
//   * window row ranges from the sorted window column, then fragcomp::compete on device arrays
// Only `rows` (input position of every processed PSM, in processing order) and `valid` come back.
#include <hipcub/hipcub.hpp>

namespace fcplan {

__device__ __forceinline__ uint64_t ckey(uint32_t p, uint8_t r) { return ((uint64_t)r << 32) | (uint64_t)p; }

// float -> unsigned pattern with the same order (pandas: ascending, NaN last, -0 == +0)
__device__ __forceinline__ uint32_t sortable(float x) {
    if (x != x) return 0xFFFFFFFFu;
    if (x == 0.0f) x = 0.0f;  // (-0 -> +0)
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void run_flag_kernel(const uint32_t *__restrict__ fp, const uint8_t *__restrict__ fr, int64_t m,
                                uint32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    flag[i] = (i == 0 || ckey(fp[i], fr[i]) != ckey(fp[i - 1], fr[i - 1])) ? 1u : 0u;
}

__global__ void run_emit_kernel(const uint32_t *__restrict__ fp, const uint8_t *__restrict__ fr, int64_t m,
                                const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos,
                                uint64_t *__restrict__ run_key, uint32_t *__restrict__ run_start,
                                uint32_t *__restrict__ run_id) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || !flag[i]) return;
    const uint32_t r = pos[i];
    run_key[r] = ckey(fp[i], fr[i]);
    run_start[r] = (uint32_t)i;
    run_id[r] = r;
}

__global__ void dup_kernel(const uint64_t *__restrict__ sorted_key, int64_t n_runs, int *__restrict__ dup) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < n_runs && sorted_key[i] == sorted_key[i + 1]) *dup = 1;
}

__global__ void psm_kernel(int64_t n, const uint32_t *__restrict__ pp, const uint8_t *__restrict__ pr,
                           const float *__restrict__ pmz, const float *__restrict__ proba,
                           const uint64_t *__restrict__ sorted_key, const uint32_t *__restrict__ sorted_run,
                           const uint32_t *__restrict__ run_start, int64_t n_runs, int64_t m, int n_rows,
                           const double *__restrict__ lower, const double *__restrict__ upper, uint32_t *__restrict__ has,
                           int64_t *__restrict__ fs, int64_t *__restrict__ fe, uint64_t *__restrict__ key64) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = ckey(pp[i], pr[i]);
    int64_t lo = 0, hi = n_runs;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sorted_key[mid] < k) lo = mid + 1;
        else hi = mid;
    }
    const bool found = lo < n_runs && sorted_key[lo] == k;
    has[i] = found ? 1u : 0u;
    int64_t a = 0, b = 0;
    if (found) {
        const uint32_t r = sorted_run[lo];
        a = run_start[r];
        b = (int64_t)r + 1 < n_runs ? (int64_t)run_start[r + 1] : m;
    }
    fs[i] = a;
    fe[i] = b;
    // the reference compares float32 m/z with the float64 limits (numpy promotes): the same here
    const double mz = (double)pmz[i];
    uint32_t w = 0;
    for (int r = 0; r < n_rows; ++r)
        if (mz >= lower[r] && mz < upper[r]) {
            w = (uint32_t)r;
            break;
        }
    key64[i] = ((uint64_t)w << 32) | (uint64_t)sortable(proba[i]);
}

__global__ void compact_kernel(int64_t n, const uint32_t *__restrict__ has, const uint32_t *__restrict__ pos,
                               const uint32_t *__restrict__ pp, uint32_t *__restrict__ c_idx, uint32_t *__restrict__ c_pidx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !has[i]) return;
    c_idx[pos[i]] = (uint32_t)i;
    c_pidx[pos[i]] = pp[i];
}

__global__ void gather_key_kernel(int64_t n, const uint32_t *__restrict__ idx, const uint64_t *__restrict__ key64,
                                  uint64_t *__restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = key64[idx[j]];
}

__global__ void finish_kernel(int64_t n, const uint32_t *__restrict__ order, const uint64_t *__restrict__ sorted_key64,
                              const float *__restrict__ prt, const int64_t *__restrict__ fs, const int64_t *__restrict__ fe,
                              int64_t *__restrict__ rows, float *__restrict__ rt_o, int64_t *__restrict__ fs_o,
                              int64_t *__restrict__ fe_o, uint8_t *__restrict__ valid, uint32_t *__restrict__ wflag) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t i = order[j];
    rows[j] = (int64_t)i;
    rt_o[j] = prt[i];
    fs_o[j] = fs[i];
    fe_o[j] = fe[i];
    valid[j] = 1;
    wflag[j] = (j == 0 || (sorted_key64[j] >> 32) != (sorted_key64[j - 1] >> 32)) ? 1u : 0u;
}

__global__ void window_emit_kernel(int64_t n, const uint32_t *__restrict__ wflag, const uint32_t *__restrict__ wpos,
                                   int64_t *__restrict__ ws) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && wflag[j]) ws[wpos[j]] = j;
}

__global__ void window_stop_kernel(int64_t n_w, int64_t n, const int64_t *__restrict__ ws, int64_t *__restrict__ we) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_w) we[k] = k + 1 < n_w ? ws[k + 1] : n;
}

}  // namespace fcplan

extern "C" {

int adh_fragcomp_frames(adh_handle_t *h, int64_t n_psm, const uint32_t *psm_precursor_idx, const uint8_t *psm_rank,
                        const float *psm_mz_observed, const float *psm_rt_observed, const float *psm_proba, int64_t n_frag,
                        const uint32_t *frag_precursor_idx, const uint8_t *frag_rank, const float *frag_mz_observed,
                        int32_t n_cycle_rows, const double *window_lower, const double *window_upper, double rt_tol_seconds,
                        double mass_tol_ppm, int64_t *rows, uint8_t *valid, int64_t *n_rows, int32_t *grouped) {
    using namespace fcplan;
    if (!h || !n_rows || !grouped || (n_psm > 0 && (!psm_precursor_idx || !psm_rank || !psm_mz_observed || !psm_rt_observed ||
                                                    !psm_proba || !rows || !valid)) ||
        (n_frag > 0 && (!frag_precursor_idx || !frag_rank || !frag_mz_observed)) ||
        (n_cycle_rows > 0 && (!window_lower || !window_upper)))
        return fail(ADH_ERR_INVALID_ARGUMENT, "adh_fragcomp_frames: NULL argument");
    if (n_psm < 0 || n_frag < 0 || n_cycle_rows < 0) return fail(ADH_ERR_INVALID_ARGUMENT, "adh_fragcomp_frames: negative size");
    *n_rows = 0;
    *grouped = 1;
    if (n_psm >= 0x7FFFFFF0ll || n_frag >= 0x7FFFFFF0ll) {  // (the scans and sorts count in int: the caller's plan takes over)
        *grouped = 0;
        return ADH_OK;
    }
    if (n_psm == 0 || n_frag == 0) return ADH_OK;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    DeviceBuffers tmp;
    struct Release {
        DeviceBuffers &b;
        ~Release() { b.release(); }
    } release{tmp};
    int rc;
    const uint32_t *d_pp, *d_fp;
    const uint8_t *d_pr, *d_fr;
    const float *d_pmz, *d_prt, *d_pproba, *d_fmz;
    const double *d_lo, *d_up;
    // device copies of the frames' columns: allocated one by one, filled together through page-locked staging
    // (the columns of a DataFrame are pageable: 125 MB for 1e6 PSMs / 12 M fragment rows)
    std::vector<UpJob> jobs;
    auto dev_copy = [&](auto *host, int64_t n, auto **dev) -> int {
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(host)>>;
        void *p = nullptr;
        HIP_TRY(hipMalloc(&p, (size_t)std::max<int64_t>(n, 1) * sizeof(T)));
        tmp.ptrs.push_back(p);
        *dev = static_cast<const T *>(p);
        if (n > 0) jobs.push_back(UpJob{p, host, (size_t)n * sizeof(T)});
        return ADH_OK;
    };
#define FP_UP(host, n, dev)                        \
    if ((rc = dev_copy(host, n, dev)) != ADH_OK) return rc;
    FP_UP(psm_precursor_idx, n_psm, &d_pp);
    FP_UP(psm_rank, n_psm, &d_pr);
    FP_UP(psm_mz_observed, n_psm, &d_pmz);
    FP_UP(psm_rt_observed, n_psm, &d_prt);
    FP_UP(psm_proba, n_psm, &d_pproba);
    FP_UP(frag_precursor_idx, n_frag, &d_fp);
    FP_UP(frag_rank, n_frag, &d_fr);
    FP_UP(frag_mz_observed, n_frag, &d_fmz);
    FP_UP(window_lower, (int64_t)n_cycle_rows, &d_lo);
    FP_UP(window_upper, (int64_t)n_cycle_rows, &d_up);
#undef FP_UP
    if ((rc = upload_staged(h, jobs)) != ADH_OK) return rc;
    auto dev_alloc = [&](void **p, size_t bytes) -> int {
        HIP_TRY(hipMalloc(p, std::max<size_t>(bytes, 16)));
        tmp.ptrs.push_back(*p);
        return ADH_OK;
    };
#define FP_ALLOC(ptr, count) \
    if ((rc = dev_alloc((void **)&ptr, sizeof(*ptr) * (size_t)(count))) != ADH_OK) return rc;
    const int64_t big = std::max(n_psm, n_frag);
    uint32_t *flag = nullptr, *pos = nullptr, *run_start = nullptr, *run_id = nullptr, *run_sorted = nullptr;
    uint64_t *run_key = nullptr, *run_key_sorted = nullptr;
    FP_ALLOC(flag, big + 1);
    FP_ALLOC(pos, big + 1);
    // ---- runs of equal keys in the fragment table
    const unsigned gm = (unsigned)((n_frag + 255) / 256), gn = (unsigned)((n_psm + 255) / 256);
    hipLaunchKernelGGL(run_flag_kernel, dim3(gm), dim3(256), 0, st, d_fp, d_fr, n_frag, flag);
    size_t cub_bytes = 0, need = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need, flag, pos, (int)(big + 1), st));
    cub_bytes = std::max(cub_bytes, need);
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, run_key, run_key_sorted, run_id, run_sorted, (int)big, 0, 40, st));
    cub_bytes = std::max(cub_bytes, need);
    {
        uint64_t *k64 = nullptr;
        uint32_t *v32 = nullptr;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k64, k64, v32, v32, (int)n_psm, 0, 64, st));
        cub_bytes = std::max(cub_bytes, need);
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, v32, v32, v32, v32, (int)n_psm, 0, 32, st));
        cub_bytes = std::max(cub_bytes, need);
    }
    void *cub_tmp = nullptr;
    if ((rc = dev_alloc(&cub_tmp, cub_bytes)) != ADH_OK) return rc;
    HIP_TRY(hipMemsetAsync(flag + n_frag, 0, 4, st));
    need = cub_bytes;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp, need, flag, pos, (int)(n_frag + 1), st));
    uint32_t n_runs_u = 0;
    HIP_TRY(hipMemcpyAsync(&n_runs_u, pos + n_frag, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int64_t n_runs = n_runs_u;
    FP_ALLOC(run_key, n_runs);
    FP_ALLOC(run_key_sorted, n_runs);
    FP_ALLOC(run_start, n_runs);
    FP_ALLOC(run_id, n_runs);
    FP_ALLOC(run_sorted, n_runs);
    hipLaunchKernelGGL(run_emit_kernel, dim3(gm), dim3(256), 0, st, d_fp, d_fr, n_frag, flag, pos, run_key, run_start, run_id);
    need = cub_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub_tmp, need, run_key, run_key_sorted, run_id, run_sorted, (int)n_runs, 0, 40, st));
    int *d_dup = nullptr;
    FP_ALLOC(d_dup, 4);
    HIP_TRY(hipMemsetAsync(d_dup, 0, 4, st));
    hipLaunchKernelGGL(dup_kernel, dim3((unsigned)((n_runs + 255) / 256)), dim3(256), 0, st, run_key_sorted, n_runs, d_dup);
    int dup = 0;
    HIP_TRY(hipMemcpyAsync(&dup, d_dup, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (dup) {  // a candidate's rows are not contiguous: first-row / last-row semantics need the general (host) plan
        *grouped = 0;
        return ADH_OK;
    }
    // ---- per PSM: fragment range, window, sort key
    uint32_t *has = nullptr, *c_idx = nullptr, *c_pidx = nullptr, *s1_idx = nullptr, *s1_key = nullptr, *order = nullptr;
    int64_t *fs = nullptr, *fe = nullptr;
    uint64_t *key64 = nullptr, *k2 = nullptr, *k2_sorted = nullptr;
    FP_ALLOC(has, n_psm + 1);
    FP_ALLOC(fs, n_psm);
    FP_ALLOC(fe, n_psm);
    FP_ALLOC(key64, n_psm);
    hipLaunchKernelGGL(psm_kernel, dim3(gn), dim3(256), 0, st, n_psm, d_pp, d_pr, d_pmz, d_pproba, run_key_sorted, run_sorted,
                       run_start, n_runs, n_frag, (int)n_cycle_rows, d_lo, d_up, has, fs, fe, key64);
    HIP_TRY(hipMemsetAsync(has + n_psm, 0, 4, st));
    need = cub_bytes;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp, need, has, pos, (int)(n_psm + 1), st));
    uint32_t n_out_u = 0;
    HIP_TRY(hipMemcpyAsync(&n_out_u, pos + n_psm, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int64_t n_out = n_out_u;
    *n_rows = n_out;
    if (n_out == 0) return ADH_OK;
    FP_ALLOC(c_idx, n_out);
    FP_ALLOC(c_pidx, n_out);
    FP_ALLOC(s1_idx, n_out);
    FP_ALLOC(s1_key, n_out);
    FP_ALLOC(order, n_out);
    FP_ALLOC(k2, n_out);
    FP_ALLOC(k2_sorted, n_out);
    hipLaunchKernelGGL(compact_kernel, dim3(gn), dim3(256), 0, st, n_psm, has, pos, d_pp, c_idx, c_pidx);
    const unsigned go = (unsigned)((n_out + 255) / 256);
    need = cub_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub_tmp, need, c_pidx, s1_key, c_idx, s1_idx, (int)n_out, 0, 32, st));
    hipLaunchKernelGGL(gather_key_kernel, dim3(go), dim3(256), 0, st, n_out, s1_idx, key64, k2);
    int wbits = 1;
    while ((1 << wbits) < std::max<int>(n_cycle_rows, 2)) ++wbits;
    need = cub_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub_tmp, need, k2, k2_sorted, s1_idx, order, (int)n_out, 0, 32 + wbits, st));
    // ---- processing-order arrays, window row ranges
    int64_t *d_rows = nullptr, *fs_o = nullptr, *fe_o = nullptr, *d_ws = nullptr, *d_we = nullptr;
    float *rt_o = nullptr;
    uint8_t *d_valid = nullptr;
    FP_ALLOC(d_rows, n_out);
    FP_ALLOC(fs_o, n_out);
    FP_ALLOC(fe_o, n_out);
    FP_ALLOC(rt_o, n_out);
    FP_ALLOC(d_valid, n_out);
    hipLaunchKernelGGL(finish_kernel, dim3(go), dim3(256), 0, st, n_out, order, k2_sorted, d_prt, fs, fe, d_rows, rt_o, fs_o, fe_o,
                       d_valid, flag);
    HIP_TRY(hipMemsetAsync(flag + n_out, 0, 4, st));
    need = cub_bytes;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp, need, flag, pos, (int)(n_out + 1), st));
    uint32_t n_w_u = 0;
    HIP_TRY(hipMemcpyAsync(&n_w_u, pos + n_out, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int64_t n_w = n_w_u;
    FP_ALLOC(d_ws, n_w);
    FP_ALLOC(d_we, n_w);
    hipLaunchKernelGGL(window_emit_kernel, dim3(go), dim3(256), 0, st, n_out, flag, pos, d_ws);
    hipLaunchKernelGGL(window_stop_kernel, dim3((unsigned)((n_w + 255) / 256)), dim3(256), 0, st, n_w, n_out, d_ws, d_we);
    HIP_TRY(hipGetLastError());
#undef FP_ALLOC
    fragcomp::Stats stats;
    hipError_t e = fragcomp::compete(st, n_w, d_ws, d_we, n_out, rt_o, fs_o, fe_o, d_fmz, rt_tol_seconds, mass_tol_ppm, d_valid,
                                     &stats);
    if (e == hipSuccess) e = hipMemcpy(rows, d_rows, (size_t)n_out * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(valid, d_valid, (size_t)n_out, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? ADH_ERR_OUT_OF_MEMORY : ADH_ERR_HIP,
                    std::string("adh_fragcomp_frames: ") + hipGetErrorString(e));
    }
    h->last_fragcomp = stats;
    return ADH_OK;
}

}  // extern "C"
