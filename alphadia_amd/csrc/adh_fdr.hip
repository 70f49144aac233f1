// adh_fdr.hip - target/decoy statistics of the FDR stage on the device:
// q-values (fdr.py:232-297), best row per group (fdr.py:181-213).  The feature table of a run
// is a few million rows: everything here is a handful of stable radix sorts and scans over
// row-sized arrays that never leave HBM (hipCUB), plus three small glue kernels.
// Included by adh_api.hip (shares its error helpers and the handle).

namespace fdr {

// pandas orders -0.0 with 0.0 and every NaN after +inf: make the bit patterns agree
__global__ void canonical_score_kernel(const double *__restrict__ in, int64_t n, double *__restrict__ out,
                                       int64_t *__restrict__ iota) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = in[i];
    if (v != v) v = __longlong_as_double(0x7FF8000000000000ll);
    else if (v == 0.0) v = 0.0;
    out[i] = v;
    iota[i] = i;
}

template <typename T>
__global__ void take_kernel(const T *__restrict__ in, const int64_t *__restrict__ order, int64_t n,
                            T *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[order[i]];
}

__global__ void take_flag_kernel(const uint8_t *__restrict__ decoy, const int64_t *__restrict__ order, int64_t n,
                                 int64_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = decoy[order[i]] ? 1 : 0;
}

// fdr.py:291-295: decoys so far / targets so far, written back to front for the running minimum
__global__ void fdr_reversed_kernel(const int64_t *__restrict__ flag, const int64_t *__restrict__ decoy_cum,
                                    int64_t n, double *__restrict__ fdr_rev) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    (void)flag;
    const double dc = (double)decoy_cum[i];
    const double tc = (double)((i + 1) - decoy_cum[i]);
    fdr_rev[n - 1 - i] = dc / tc;
}

__global__ void reverse_kernel(const double *__restrict__ in, int64_t n, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[n - 1 - i];
}

// np.minimum propagates NaN; fdr values here are never NaN (the first row is a target or a decoy)
struct MinOp {
    __device__ __forceinline__ double operator()(double a, double b) const { return b < a ? b : a; }
};

// first row of every run of equal (a, b) keys in the sorted order
__global__ void group_head_kernel(const int64_t *__restrict__ order, const int64_t *__restrict__ a,
                                  const int64_t *__restrict__ b, int64_t n, uint8_t *__restrict__ keep) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = order[i];
    bool head = true;
    if (i > 0) {
        const int64_t p = order[i - 1];
        head = a[r] != a[p] || (b && b[r] != b[p]);
    }
    keep[r] = head ? 1 : 0;
}

struct Scratch {
    std::vector<void *> ptrs;
    ~Scratch() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    hipError_t alloc(T **p, size_t count) {
        hipError_t e = hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

// stable sort of `order` (and nothing else) by key[order[i]]
template <typename K>
hipError_t stable_sort_by(Scratch &s, const K *d_key, int64_t n, int64_t *&order, int64_t *&order_alt, K *k_in,
                          K *k_out, hipStream_t st) {
    hipLaunchKernelGGL((take_kernel<K>), grid_for(n), dim3(256), 0, st, d_key, order, n, k_in);
    size_t bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k_in, k_out, order, order_alt, (int)n, 0,
                                                      (int)sizeof(K) * 8, st);
    if (e != hipSuccess) return e;
    void *tmp = nullptr;
    if ((e = s.alloc((char **)&tmp, bytes)) != hipSuccess) return e;
    e = hipcub::DeviceRadixSort::SortPairs(tmp, bytes, k_in, k_out, order, order_alt, (int)n, 0, (int)sizeof(K) * 8,
                                           st);
    std::swap(order, order_alt);
    return e;
}

}  // namespace fdr

// ---- device cores: every argument is a device buffer; results stay in HBM --------------------
namespace fdr {

// get_q_values (fdr.py:232-297) + _fdr_to_q_values (fdr.py:215-230).  On return order[i] = input row
// at sorted position i and qval[i] = its q-value (both owned by `s`).
int q_values_core(adh_handle *h, Scratch &s, int64_t n, const double *d_score, const uint8_t *d_decoy,
                  const int64_t *d_tie, int64_t **order_out, double **qval_out) {
    hipStream_t st = h->stream;
    double *d_canon = nullptr, *k_in = nullptr, *k_out = nullptr, *d_fdr = nullptr, *d_q = nullptr;
    int64_t *order = nullptr, *order_alt = nullptr, *d_flag = nullptr, *d_cum = nullptr, *t_in = nullptr, *t_out = nullptr;
    HIP_TRY(s.alloc(&d_canon, n));
    HIP_TRY(s.alloc(&k_in, n));
    HIP_TRY(s.alloc(&k_out, n));
    HIP_TRY(s.alloc(&d_fdr, n));
    HIP_TRY(s.alloc(&d_q, n));
    HIP_TRY(s.alloc(&order, n));
    HIP_TRY(s.alloc(&order_alt, n));
    HIP_TRY(s.alloc(&d_flag, n));
    HIP_TRY(s.alloc(&d_cum, n));
    hipLaunchKernelGGL(canonical_score_kernel, grid_for(n), dim3(256), 0, st, d_score, n, d_canon, order);
    // sort_values([score, decoy, tiebreak]) == stable sorts from the last key to the first
    if (d_tie) {
        HIP_TRY(s.alloc(&t_in, n));
        HIP_TRY(s.alloc(&t_out, n));
        HIP_TRY(stable_sort_by<int64_t>(s, d_tie, n, order, order_alt, t_in, t_out, st));
    }
    {
        hipLaunchKernelGGL(take_flag_kernel, grid_for(n), dim3(256), 0, st, d_decoy, order, n, d_flag);
        size_t bytes = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, d_flag, d_cum, order, order_alt, (int)n, 0, 1, st));
        void *tmp = nullptr;
        HIP_TRY(s.alloc((char **)&tmp, bytes));
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, d_flag, d_cum, order, order_alt, (int)n, 0, 1, st));
        std::swap(order, order_alt);
    }
    HIP_TRY(stable_sort_by<double>(s, d_canon, n, order, order_alt, k_in, k_out, st));
    // cumulative decoys / targets, then the running minimum from the back
    hipLaunchKernelGGL(take_flag_kernel, grid_for(n), dim3(256), 0, st, d_decoy, order, n, d_flag);
    {
        size_t bytes = 0;
        HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, d_flag, d_cum, (int)n, st));
        void *tmp = nullptr;
        HIP_TRY(s.alloc((char **)&tmp, bytes));
        HIP_TRY(hipcub::DeviceScan::InclusiveSum(tmp, bytes, d_flag, d_cum, (int)n, st));
    }
    hipLaunchKernelGGL(fdr_reversed_kernel, grid_for(n), dim3(256), 0, st, d_flag, d_cum, n, d_fdr);
    {
        size_t bytes = 0;
        HIP_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, bytes, d_fdr, d_q, MinOp(), (int)n, st));
        void *tmp = nullptr;
        HIP_TRY(s.alloc((char **)&tmp, bytes));
        HIP_TRY(hipcub::DeviceScan::InclusiveScan(tmp, bytes, d_fdr, d_q, MinOp(), (int)n, st));
    }
    hipLaunchKernelGGL(reverse_kernel, grid_for(n), dim3(256), 0, st, d_q, n, d_fdr);
    HIP_TRY(hipGetLastError());
    *order_out = order;
    *qval_out = d_fdr;
    return ADH_OK;
}

// keep_best (fdr.py:181-213): d_keep[row] = 1 for the lowest-score row of every (a[, b]) group
int keep_best_core(adh_handle *h, Scratch &s, int64_t n, const double *d_score, const int64_t *d_a,
                   const int64_t *d_b, uint8_t *d_keep) {
    hipStream_t st = h->stream;
    double *d_canon = nullptr, *k_in = nullptr, *k_out = nullptr;
    int64_t *order = nullptr, *order_alt = nullptr, *t_in = nullptr, *t_out = nullptr;
    HIP_TRY(s.alloc(&d_canon, n));
    HIP_TRY(s.alloc(&k_in, n));
    HIP_TRY(s.alloc(&k_out, n));
    HIP_TRY(s.alloc(&order, n));
    HIP_TRY(s.alloc(&order_alt, n));
    HIP_TRY(s.alloc(&t_in, n));
    HIP_TRY(s.alloc(&t_out, n));
    hipLaunchKernelGGL(canonical_score_kernel, grid_for(n), dim3(256), 0, st, d_score, n, d_canon, order);
    // rows of one group become adjacent, ordered by (score, input row)
    HIP_TRY(stable_sort_by<double>(s, d_canon, n, order, order_alt, k_in, k_out, st));
    if (d_b) HIP_TRY(stable_sort_by<int64_t>(s, d_b, n, order, order_alt, t_in, t_out, st));
    HIP_TRY(stable_sort_by<int64_t>(s, d_a, n, order, order_alt, t_in, t_out, st));
    hipLaunchKernelGGL(group_head_kernel, grid_for(n), dim3(256), 0, st, order, d_a, d_b, n, d_keep);
    HIP_TRY(hipGetLastError());
    return ADH_OK;
}

}  // namespace fdr

// fdr.py:232-297 (get_q_values) with fdr.py:215-230 (_fdr_to_q_values): host arrays in and out.
int adh_fdr_q_values(adh_handle_t *h, int64_t n, const double *score, const uint8_t *decoy,
                     const int64_t *tiebreak, int64_t *order_out, double *qval_out) {
    if (!h || n < 0 || (n > 0 && (!score || !decoy || !order_out || !qval_out)))
        return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n >= (int64_t)0x7FFFFFFFll) return fail(ADH_ERR_UNSUPPORTED, "2^31 or more rows");
    if (n == 0) return ADH_OK;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    fdr::Scratch s;
    double *d_score = nullptr, *d_qval = nullptr;
    uint8_t *d_decoy = nullptr;
    int64_t *d_tie = nullptr, *d_order = nullptr;
    HIP_TRY(s.alloc(&d_score, n));
    HIP_TRY(s.alloc(&d_decoy, n));
    HIP_TRY(hipMemcpyAsync(d_score, score, n * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_decoy, decoy, n, hipMemcpyHostToDevice, st));
    if (tiebreak) {
        HIP_TRY(s.alloc(&d_tie, n));
        HIP_TRY(hipMemcpyAsync(d_tie, tiebreak, n * sizeof(int64_t), hipMemcpyHostToDevice, st));
    }
    int rc = fdr::q_values_core(h, s, n, d_score, d_decoy, d_tie, &d_order, &d_qval);
    if (rc != ADH_OK) return rc;
    HIP_TRY(hipMemcpyAsync(qval_out, d_qval, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(order_out, d_order, n * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->d2h_bytes += (uint64_t)n * 16;
    return ADH_OK;
}

// fdr.py:181-213 (keep_best): per (group_a[, group_b]) the row with the lowest score, the earliest
// row on ties; `keep` is a mask over the rows in their input order.  Host arrays in and out.
int adh_fdr_keep_best(adh_handle_t *h, int64_t n, const double *score, const int64_t *group_a,
                      const int64_t *group_b, uint8_t *keep) {
    if (!h || n < 0 || (n > 0 && (!score || !group_a || !keep))) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n >= (int64_t)0x7FFFFFFFll) return fail(ADH_ERR_UNSUPPORTED, "2^31 or more rows");
    if (n == 0) return ADH_OK;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    fdr::Scratch s;
    double *d_score = nullptr;
    int64_t *d_a = nullptr, *d_b = nullptr;
    uint8_t *d_keep = nullptr;
    HIP_TRY(s.alloc(&d_score, n));
    HIP_TRY(s.alloc(&d_a, n));
    HIP_TRY(s.alloc(&d_keep, n));
    HIP_TRY(hipMemcpyAsync(d_score, score, n * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_a, group_a, n * sizeof(int64_t), hipMemcpyHostToDevice, st));
    if (group_b) {
        HIP_TRY(s.alloc(&d_b, n));
        HIP_TRY(hipMemcpyAsync(d_b, group_b, n * sizeof(int64_t), hipMemcpyHostToDevice, st));
    }
    int rc = fdr::keep_best_core(h, s, n, d_score, d_a, d_b, d_keep);
    if (rc != ADH_OK) return rc;
    HIP_TRY(hipMemcpyAsync(keep, d_keep, n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->d2h_bytes += (uint64_t)n;
    return ADH_OK;
}
