// adh_fdr_device.hip - the FDR stage fed straight from the scoring tables in HBM (SURVEY section
// 8f row 3: "feature table stays on GPU -> classifier -> q-values -> fragcomp -> keep_best").
//
// The reference hands DataFrames from scoring to perform_fdr on the host
// (alphadia/workflow/peptidecentric/peptidecentric.py:219-243 -> alphadia/fdr/fdr.py:24-178).  Here the
// tables adh_score_candidates left in HBM are the input:
//   adh_mlp_stage_rows_device  fdr.py:86-105  dropna + np.concatenate([targets, decoys]) as a stable
//                                             partition of the usable rows; X gathered on the device
//   adh_mlp_fit                classifiers.py:316-433 (unchanged, trains on the staged rows)
//   adh_mlp_predict_resident   fdr.py:133     probabilities stay in HBM
//   adh_fdr_resident           fdr.py:134-178 q-values -> fragment competition (fragcomp.py:231-299,
//                                             fed by the fragment_mz_observed table) -> best row per
//                                             group -> q-values
// Only row-sized metadata crosses PCIe: decoy flags, group keys and tie-breakers in (one byte + three
// int64 per candidate), the surviving (row, proba, qval) triples out.  The 46-float feature rows and
// the fragment tables never leave the GPU (adh_transfer_counters lets a test check that).
// Included by adh_api.hip after adh_fdr.hip and adh_mlp.hip.

namespace fdrdev {

using fdr::grid_for;
using fdr::Scratch;

// value of classifier column j for candidate row i: a feature, an extra per-candidate column, or
// rt_observed (feature 2) minus an extra column (delta_rt, scoring.py:458)
struct ColumnSpec {
    int32_t d;
    int32_t src[64];
    const float *extra[8];
};

__device__ __forceinline__ float column_value(const ColumnSpec &c, const float *__restrict__ features, int64_t i, int j) {
    const int s = c.src[j];
    if (s >= 0 && s < ADH_NUM_FEATURES) return features[i * ADH_NUM_FEATURES + s];
    if (s >= ADH_NUM_FEATURES) return c.extra[s - ADH_NUM_FEATURES][i];
    return features[i * ADH_NUM_FEATURES + 2] - c.extra[-1 - s][i];
}

// usable rows (valid, no NaN in a classifier column: dropna, fdr.py:86-87), split by decoy flag
__global__ void usable_kernel(ColumnSpec c, const uint8_t *__restrict__ valid, const float *__restrict__ features,
                              const uint8_t *__restrict__ decoy, int64_t n, int32_t *__restrict__ flag_t,
                              int32_t *__restrict__ flag_d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool ok = valid[i] != 0;
    for (int j = 0; ok && j < c.d; ++j) {
        const float v = column_value(c, features, i, j);
        ok = !(v != v);
    }
    flag_t[i] = ok && !decoy[i];
    flag_d[i] = ok && decoy[i];
}

__global__ void stage_kernel(ColumnSpec c, const float *__restrict__ features, const uint8_t *__restrict__ decoy,
                             const int32_t *__restrict__ flag_t, const int32_t *__restrict__ flag_d,
                             const int32_t *__restrict__ pos_t, const int32_t *__restrict__ pos_d, int64_t n,
                             float *__restrict__ X, float *__restrict__ Y, int64_t *__restrict__ rowmap,
                             uint8_t *__restrict__ decoy_rows, int64_t *__restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t n_t = (int64_t)pos_t[n - 1] + flag_t[n - 1];
    if (i == 0) {
        counts[0] = n_t;
        counts[1] = (int64_t)pos_d[n - 1] + flag_d[n - 1];
    }
    int64_t p;
    if (flag_t[i]) p = pos_t[i];
    else if (flag_d[i]) p = n_t + pos_d[i];
    else return;
    for (int j = 0; j < c.d; ++j) X[p * c.d + j] = column_value(c, features, i, j);
    Y[p] = decoy[i] ? 1.0f : 0.0f;
    rowmap[p] = i;
    decoy_rows[p] = decoy[i];
}

__global__ void proba_to_score_kernel(const float *__restrict__ proba, int out_dim, int64_t n, double *__restrict__ score) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) score[i] = (double)proba[i * out_dim + 1];  // psm_df["proba"] = predict_proba(X)[:, 1], fdr.py:133
}

template <typename T>
__global__ void take2_kernel(const T *__restrict__ table, const int64_t *__restrict__ rowmap, const int64_t *__restrict__ ids,
                             int64_t n, T *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = table[rowmap[ids[i]]];
}

// first sorted position whose q-value is >= the heuristic (searchsorted(..., "left"), fdr.py:148)
__global__ void first_at_least_kernel(const double *__restrict__ qval, int64_t n, double bound, int64_t *__restrict__ out) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (qval[mid] < bound) lo = mid + 1; else hi = mid;
    }
    *out = lo;
}

// per PSM of the competition (sorted position i -> staged row ids[i]): DIA window of the observed m/z
// (fragcomp.py:170-202), retention time, fragment range inside the fragment_mz_observed table
__global__ void competition_rows_kernel(const int64_t *__restrict__ ids, const int64_t *__restrict__ rowmap, int64_t n,
                                        const float *__restrict__ features, const float *__restrict__ frag_mz_library,
                                        int32_t top_k, const double *__restrict__ win_lo, const double *__restrict__ win_hi,
                                        int32_t n_win, int64_t *__restrict__ window, float *__restrict__ rt,
                                        int64_t *__restrict__ fstart, int64_t *__restrict__ fstop) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = rowmap[ids[i]];
    const float mz = features[r * ADH_NUM_FEATURES + 10];  // mz_observed
    int w = 0;
    for (int k = 0; k < n_win; ++k)
        if ((double)mz >= win_lo[k] && (double)mz < win_hi[k]) {
            w = k;
            break;
        }
    window[i] = w;
    rt[i] = features[r * ADH_NUM_FEATURES + 2];  // rt_observed
    int cnt = 0;
    for (int k = 0; k < top_k; ++k) cnt += frag_mz_library[r * top_k + k] > 0.0f;  // collect_fragments, scoring.py:526
    fstart[i] = r * (int64_t)top_k;
    fstop[i] = r * (int64_t)top_k + cnt;
}

__global__ void window_bounds_kernel(const int64_t *__restrict__ sorted_window, int64_t n, int32_t n_win,
                                     int64_t *__restrict__ start, int64_t *__restrict__ stop) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t w = sorted_window[i];
    if (i == 0 || sorted_window[i - 1] != w) start[w] = i;
    if (i == n - 1 || sorted_window[i + 1] != w) stop[w] = i + 1;
    (void)n_win;
}

__global__ void iota_kernel(int64_t *x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = i;
}

__global__ void fill_u8_kernel(uint8_t *x, int64_t n, uint8_t v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}

template <typename T>
__global__ void gather_kernel(const T *__restrict__ in, const int64_t *__restrict__ idx, int64_t n, T *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}

// keep the entries of `ids` whose flag is set, order preserved; returns the new count
int compact_ids(adh_handle *h, Scratch &s, const int64_t *ids, const uint8_t *flags, int64_t n, int64_t **out,
                int64_t *n_out) {
    hipStream_t st = h->stream;
    int64_t *d_out = nullptr, *d_count = nullptr;
    HIP_TRY(s.alloc(&d_out, n));
    HIP_TRY(s.alloc(&d_count, 1));
    size_t bytes = 0;
    HIP_TRY(hipcub::DeviceSelect::Flagged(nullptr, bytes, ids, flags, d_out, d_count, (int)n, st));
    void *tmp = nullptr;
    HIP_TRY(s.alloc((char **)&tmp, bytes));
    HIP_TRY(hipcub::DeviceSelect::Flagged(tmp, bytes, ids, flags, d_out, d_count, (int)n, st));
    int64_t cnt = 0;
    HIP_TRY(hipMemcpyAsync(&cnt, d_count, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->d2h_bytes += 8;
    *out = d_out;
    *n_out = cnt;
    return ADH_OK;
}

}  // namespace fdrdev

int adh_mlp_stage_rows_device(adh_mlp_t *m, const int32_t *src_cols, int32_t d, const float *const *extra_cols,
                              int32_t n_extra, const uint8_t *decoy, int64_t n_rows, int64_t *n_targets,
                              int64_t *n_decoys) {
    using namespace fdrdev;
    if (!m || !src_cols || !decoy || !n_targets || !n_decoys) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    adh_handle *h = m->h;
    if (h->last_tables < 0 || h->last_rows != n_rows)
        return fail(ADH_ERR_NOT_STAGED, "the device tables of the last adh_score_candidates call do not have n_rows rows");
    if (d != m->A.dims[0] || d < 1 || d > 64) return fail(ADH_ERR_INVALID_ARGUMENT, "column count must equal the network input (<= 64)");
    if (n_extra < 0 || n_extra > 8 || (n_extra > 0 && !extra_cols)) return fail(ADH_ERR_INVALID_ARGUMENT, "at most 8 extra columns");
    if (n_rows >= 0x7FFFFFFFll) return fail(ADH_ERR_UNSUPPORTED, "2^31 or more rows");
    ColumnSpec spec;
    memset(&spec, 0, sizeof(spec));
    spec.d = d;
    for (int j = 0; j < d; ++j) {
        const int sc = src_cols[j];
        if (sc >= ADH_NUM_FEATURES + n_extra || sc < -n_extra) return fail(ADH_ERR_INVALID_ARGUMENT, "column source out of range");
        spec.src[j] = sc;
    }
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const int64_t n = n_rows;
    *n_targets = *n_decoys = 0;
    for (void **p : {(void **)&m->d_X, (void **)&m->d_Y, (void **)&m->d_rowmap, (void **)&m->d_decoy, (void **)&m->d_proba}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    m->n_rows = 0;
    m->has_y = false;
    m->proba_ready = false;
    m->n_table = n;
    if (n == 0) {  // an empty table is staged, with no rows (adh_mlp_staged_rows then succeeds and returns none)
        HIP_TRY(hipMalloc((void **)&m->d_rowmap, 8));
        return ADH_OK;
    }
    const adh_output_t &tab = h->tables[h->last_tables].view;
    Scratch s;
    uint8_t *d_decoy_all = nullptr;
    int32_t *flag_t = nullptr, *flag_d = nullptr, *pos_t = nullptr, *pos_d = nullptr;
    int64_t *d_counts = nullptr;
    HIP_TRY(s.alloc(&d_decoy_all, n));
    HIP_TRY(s.alloc(&flag_t, n));
    HIP_TRY(s.alloc(&flag_d, n));
    HIP_TRY(s.alloc(&pos_t, n));
    HIP_TRY(s.alloc(&pos_d, n));
    HIP_TRY(s.alloc(&d_counts, 2));
    HIP_TRY(hipMemcpyAsync(d_decoy_all, decoy, n, hipMemcpyHostToDevice, st));
    for (int e = 0; e < n_extra; ++e) {
        float *d_e = nullptr;
        HIP_TRY(s.alloc(&d_e, n));
        HIP_TRY(hipMemcpyAsync(d_e, extra_cols[e], n * 4, hipMemcpyHostToDevice, st));
        spec.extra[e] = d_e;
    }
    hipLaunchKernelGGL(usable_kernel, grid_for(n), dim3(256), 0, st, spec, tab.valid, tab.features, d_decoy_all, n, flag_t, flag_d);
    {
        size_t bytes = 0;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, flag_t, pos_t, (int)n, st));
        void *tmp = nullptr;
        HIP_TRY(s.alloc((char **)&tmp, bytes));
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, bytes, flag_t, pos_t, (int)n, st));
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, bytes, flag_d, pos_d, (int)n, st));
    }
    // upper bound of the staged rows is n: allocate once, the live count comes back with the scatter
    // (at least one element each: an empty table must still count as staged, adh_mlp_staged_rows)
    HIP_TRY(hipMalloc((void **)&m->d_X, std::max<size_t>((size_t)n * d * 4, 4)));
    HIP_TRY(hipMalloc((void **)&m->d_Y, std::max<size_t>((size_t)n * 4, 4)));
    HIP_TRY(hipMalloc((void **)&m->d_rowmap, std::max<size_t>((size_t)n * 8, 8)));
    HIP_TRY(hipMalloc((void **)&m->d_decoy, std::max<size_t>((size_t)n, 1)));
    hipLaunchKernelGGL(stage_kernel, grid_for(n), dim3(256), 0, st, spec, tab.features, d_decoy_all, flag_t, flag_d, pos_t,
                       pos_d, n, m->d_X, m->d_Y, m->d_rowmap, m->d_decoy, d_counts);
    HIP_TRY(hipGetLastError());
    int64_t counts[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(counts, d_counts, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->d2h_bytes += 16;
    *n_targets = counts[0];
    *n_decoys = counts[1];
    m->n_rows = counts[0] + counts[1];
    m->has_y = true;
    return ADH_OK;
}

int adh_mlp_staged_rows(adh_mlp_t *m, int64_t *rows_out, int64_t capacity) {
    if (!m || !rows_out) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!m->d_rowmap) return fail(ADH_ERR_NOT_STAGED, "rows were not staged from the device tables");
    if (capacity < m->n_rows) return fail(ADH_ERR_INVALID_ARGUMENT, "rows_out too small");
    HIP_TRY(hipSetDevice(m->h->device));
    HIP_TRY(hipMemcpy(rows_out, m->d_rowmap, (size_t)m->n_rows * 8, hipMemcpyDeviceToHost));
    m->h->d2h_bytes += (uint64_t)m->n_rows * 8;
    return ADH_OK;
}

int adh_mlp_predict_resident(adh_mlp_t *m) {
    if (!m) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!m->d_X) return fail(ADH_ERR_NOT_STAGED, "stage the rows first");
    HIP_TRY(hipSetDevice(m->h->device));
    hipStream_t st = m->h->stream;
    const int64_t n = m->n_rows;
    const int out_dim = m->A.dims[m->A.n_linear];
    if (m->d_proba) (void)hipFree(m->d_proba);
    m->d_proba = nullptr;
    m->proba_ready = false;
    HIP_TRY(hipMalloc((void **)&m->d_proba, std::max<size_t>((size_t)n * out_dim * 4, 4)));
    if (n > 0) {
        const int64_t n_tiles = (n + ADH_MLP_TR - 1) / ADH_MLP_TR;
        const unsigned grid = (unsigned)std::min<int64_t>(n_tiles, 256 * 8);
        hipLaunchKernelGGL(adh_mlp_predict_kernel, dim3(grid), dim3(ADH_MLP_THREADS), m->lds_bytes, st, m->A, m->d_P, m->d_rm,
                           m->d_rv, m->d_X, (const int64_t *)nullptr, n, m->d_proba);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
    }
    m->proba_ready = true;
    return ADH_OK;
}

int adh_fdr_resident(adh_handle_t *h, adh_mlp_t *m, const int64_t *group_a, const int64_t *group_b,
                     const int64_t *tiebreak, const double *cycle, int32_t cycle_len, int32_t cycle_scans,
                     double rt_tol_seconds, double mass_tol_ppm, double fdr_heuristic, int64_t *n_out,
                     int64_t *row_out, float *proba_out, double *qval_out) {
    using namespace fdrdev;
    if (!h || !m || !group_a || !n_out || !row_out || !proba_out || !qval_out)
        return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (m->h != h || !m->d_rowmap || !m->proba_ready)
        return fail(ADH_ERR_NOT_STAGED, "stage the rows from the device tables and run adh_mlp_predict_resident first");
    if (h->last_tables < 0 || h->last_rows != m->n_table)
        return fail(ADH_ERR_NOT_STAGED, "the device tables were replaced since the rows were staged");
    if (cycle && (cycle_len < 1 || cycle_scans < 1)) return fail(ADH_ERR_INVALID_ARGUMENT, "invalid cycle shape");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const int64_t n_table = m->n_table, n0 = m->n_rows;
    const int out_dim = m->A.dims[m->A.n_linear];
    *n_out = 0;
    if (n0 == 0) return ADH_OK;
    {
        const int rc_m = materialise_tables(h);  // (fragment competition reads fragment_mz_library)
        if (rc_m != ADH_OK) return rc_m;
    }
    const adh_output_t &tab = h->tables[h->last_tables].view;
    Scratch s;
    // per-candidate keys (host, one int64 each) -> device
    int64_t *d_ga = nullptr, *d_gb = nullptr, *d_tb = nullptr;
    HIP_TRY(s.alloc(&d_ga, n_table));
    HIP_TRY(hipMemcpyAsync(d_ga, group_a, n_table * 8, hipMemcpyHostToDevice, st));
    if (group_b) {
        HIP_TRY(s.alloc(&d_gb, n_table));
        HIP_TRY(hipMemcpyAsync(d_gb, group_b, n_table * 8, hipMemcpyHostToDevice, st));
    }
    if (tiebreak) {
        HIP_TRY(s.alloc(&d_tb, n_table));
        HIP_TRY(hipMemcpyAsync(d_tb, tiebreak, n_table * 8, hipMemcpyHostToDevice, st));
    }
    // staged-row columns
    double *score = nullptr;
    int64_t *ids = nullptr, *tie_rows = nullptr;
    HIP_TRY(s.alloc(&score, n0));
    HIP_TRY(s.alloc(&ids, n0));
    hipLaunchKernelGGL(proba_to_score_kernel, grid_for(n0), dim3(256), 0, st, m->d_proba, out_dim, n0, score);
    hipLaunchKernelGGL(iota_kernel, grid_for(n0), dim3(256), 0, st, ids, n0);
    if (d_tb) {
        HIP_TRY(s.alloc(&tie_rows, n0));
        hipLaunchKernelGGL((take2_kernel<int64_t>), grid_for(n0), dim3(256), 0, st, d_tb, m->d_rowmap, ids, n0, tie_rows);
    }
    // ---- q-values of all staged rows (fdr.py:134)
    int64_t *order = nullptr;
    double *qval = nullptr;
    int rc = fdr::q_values_core(h, s, n0, score, m->d_decoy, tie_rows, &order, &qval);
    if (rc != ADH_OK) return rc;
    // `order` lists the staged rows in the sorted frame; that is the current PSM list
    int64_t *cur = order;
    int64_t n_cur = n0;
    if (cycle && cycle_scans <= 2) {  // fdr.py:146 (MAX_DIA_CYCLE_SHAPE)
        // ---- rows below the heuristic FDR compete for fragments (fdr.py:146-163)
        int64_t *d_first = nullptr;
        HIP_TRY(s.alloc(&d_first, 1));
        hipLaunchKernelGGL(first_at_least_kernel, dim3(1), dim3(1), 0, st, qval, n0, fdr_heuristic, d_first);
        int64_t start_idx = 0;
        HIP_TRY(hipMemcpyAsync(&start_idx, d_first, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        h->d2h_bytes += 8;
        if (start_idx == 0) start_idx = n0;
        const int64_t nc = start_idx;
        // DIA windows: extreme isolation limits of every cycle row over its scans (fragcomp.py:186-189)
        std::vector<double> lo((size_t)cycle_len), hi((size_t)cycle_len);
        for (int r = 0; r < cycle_len; ++r) {
            double a = cycle[2 * ((size_t)r * cycle_scans)], b = cycle[2 * ((size_t)r * cycle_scans) + 1];
            for (int sc = 1; sc < cycle_scans; ++sc) {
                a = std::min(a, cycle[2 * ((size_t)r * cycle_scans + sc)]);
                b = std::max(b, cycle[2 * ((size_t)r * cycle_scans + sc) + 1]);
            }
            lo[(size_t)r] = a;
            hi[(size_t)r] = b;
        }
        double *d_lo = nullptr, *d_hi = nullptr;
        HIP_TRY(s.alloc(&d_lo, cycle_len));
        HIP_TRY(s.alloc(&d_hi, cycle_len));
        HIP_TRY(hipMemcpyAsync(d_lo, lo.data(), (size_t)cycle_len * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_hi, hi.data(), (size_t)cycle_len * 8, hipMemcpyHostToDevice, st));
        int64_t *window = nullptr, *fstart = nullptr, *fstop = nullptr, *pos = nullptr, *pos_alt = nullptr;
        float *rt = nullptr;
        HIP_TRY(s.alloc(&window, nc));
        HIP_TRY(s.alloc(&fstart, nc));
        HIP_TRY(s.alloc(&fstop, nc));
        HIP_TRY(s.alloc(&rt, nc));
        HIP_TRY(s.alloc(&pos, nc));
        HIP_TRY(s.alloc(&pos_alt, nc));
        hipLaunchKernelGGL(competition_rows_kernel, grid_for(nc), dim3(256), 0, st, cur, m->d_rowmap, nc, tab.features,
                           tab.fragment_mz_library, (int32_t)tab.top_k, d_lo, d_hi, cycle_len, window, rt, fstart, fstop);
        // processing order: window, then proba, then precursor_idx, stable (fragcomp.py:268-270)
        hipLaunchKernelGGL(iota_kernel, grid_for(nc), dim3(256), 0, st, pos, nc);
        int64_t *k64 = nullptr, *k64b = nullptr;
        double *kd = nullptr, *kdb = nullptr, *score_cur = nullptr, *canon = nullptr;
        int64_t *tie_cur = nullptr, *scratch_iota = nullptr;
        HIP_TRY(s.alloc(&k64, nc));
        HIP_TRY(s.alloc(&k64b, nc));
        HIP_TRY(s.alloc(&kd, nc));
        HIP_TRY(s.alloc(&kdb, nc));
        HIP_TRY(s.alloc(&score_cur, nc));
        HIP_TRY(s.alloc(&canon, nc));
        HIP_TRY(s.alloc(&scratch_iota, nc));
        hipLaunchKernelGGL((fdrdev::gather_kernel<double>), grid_for(nc), dim3(256), 0, st, score, cur, nc, score_cur);
        hipLaunchKernelGGL(fdr::canonical_score_kernel, grid_for(nc), dim3(256), 0, st, score_cur, nc, canon, scratch_iota);
        if (tie_rows) {
            HIP_TRY(s.alloc(&tie_cur, nc));
            hipLaunchKernelGGL((fdrdev::gather_kernel<int64_t>), grid_for(nc), dim3(256), 0, st, tie_rows, cur, nc, tie_cur);
            HIP_TRY(fdr::stable_sort_by<int64_t>(s, tie_cur, nc, pos, pos_alt, k64, k64b, st));
        }
        HIP_TRY(fdr::stable_sort_by<double>(s, canon, nc, pos, pos_alt, kd, kdb, st));
        HIP_TRY(fdr::stable_sort_by<int64_t>(s, window, nc, pos, pos_alt, k64, k64b, st));
        // columns in processing order
        int64_t *w_sorted = nullptr, *fs_sorted = nullptr, *fe_sorted = nullptr, *ws = nullptr, *we = nullptr, *ids_sorted = nullptr;
        float *rt_sorted = nullptr;
        uint8_t *alive = nullptr;
        HIP_TRY(s.alloc(&w_sorted, nc));
        HIP_TRY(s.alloc(&fs_sorted, nc));
        HIP_TRY(s.alloc(&fe_sorted, nc));
        HIP_TRY(s.alloc(&rt_sorted, nc));
        HIP_TRY(s.alloc(&ids_sorted, nc));
        HIP_TRY(s.alloc(&alive, nc));
        HIP_TRY(s.alloc(&ws, cycle_len));
        HIP_TRY(s.alloc(&we, cycle_len));
        hipLaunchKernelGGL((fdrdev::gather_kernel<int64_t>), grid_for(nc), dim3(256), 0, st, window, pos, nc, w_sorted);
        hipLaunchKernelGGL((fdrdev::gather_kernel<int64_t>), grid_for(nc), dim3(256), 0, st, fstart, pos, nc, fs_sorted);
        hipLaunchKernelGGL((fdrdev::gather_kernel<int64_t>), grid_for(nc), dim3(256), 0, st, fstop, pos, nc, fe_sorted);
        hipLaunchKernelGGL((fdrdev::gather_kernel<float>), grid_for(nc), dim3(256), 0, st, rt, pos, nc, rt_sorted);
        hipLaunchKernelGGL((fdrdev::gather_kernel<int64_t>), grid_for(nc), dim3(256), 0, st, cur, pos, nc, ids_sorted);
        HIP_TRY(hipMemsetAsync(ws, 0, (size_t)cycle_len * 8, st));
        HIP_TRY(hipMemsetAsync(we, 0, (size_t)cycle_len * 8, st));
        hipLaunchKernelGGL(window_bounds_kernel, grid_for(nc), dim3(256), 0, st, w_sorted, nc, cycle_len, ws, we);
        hipLaunchKernelGGL(fill_u8_kernel, grid_for(nc), dim3(256), 0, st, alive, nc, (uint8_t)1);
        HIP_TRY(hipGetLastError());
        HIP_TRY(fragcomp::compete(st, (int64_t)cycle_len, ws, we, nc, rt_sorted, fs_sorted, fe_sorted, tab.fragment_mz_observed,
                                  rt_tol_seconds, mass_tol_ppm, alive, &h->last_fragcomp));
        // survivors, in processing order (FragmentCompetition.__call__ returns psm_df[valid])
        rc = compact_ids(h, s, ids_sorted, alive, nc, &cur, &n_cur);
        if (rc != ADH_OK) return rc;
    }
    if (n_cur == 0) return ADH_OK;
    // ---- best row per group (fdr.py:165-166), then q-values again (fdr.py:167)
    {
        double *score_cur = nullptr;
        int64_t *a_cur = nullptr, *b_cur = nullptr, *iota = nullptr;
        uint8_t *keep = nullptr;
        HIP_TRY(s.alloc(&score_cur, n_cur));
        HIP_TRY(s.alloc(&a_cur, n_cur));
        HIP_TRY(s.alloc(&keep, n_cur));
        HIP_TRY(s.alloc(&iota, n_cur));
        hipLaunchKernelGGL((fdrdev::gather_kernel<double>), grid_for(n_cur), dim3(256), 0, st, score, cur, n_cur, score_cur);
        hipLaunchKernelGGL(iota_kernel, grid_for(n_cur), dim3(256), 0, st, iota, n_cur);
        hipLaunchKernelGGL((take2_kernel<int64_t>), grid_for(n_cur), dim3(256), 0, st, d_ga, m->d_rowmap, cur, n_cur, a_cur);
        if (d_gb) {
            HIP_TRY(s.alloc(&b_cur, n_cur));
            hipLaunchKernelGGL((take2_kernel<int64_t>), grid_for(n_cur), dim3(256), 0, st, d_gb, m->d_rowmap, cur, n_cur, b_cur);
        }
        rc = fdr::keep_best_core(h, s, n_cur, score_cur, a_cur, b_cur, keep);
        if (rc != ADH_OK) return rc;
        rc = compact_ids(h, s, cur, keep, n_cur, &cur, &n_cur);
        if (rc != ADH_OK) return rc;
    }
    if (n_cur == 0) return ADH_OK;
    {
        double *score_cur = nullptr, *qv2 = nullptr;
        uint8_t *decoy_cur = nullptr;
        int64_t *tie_cur = nullptr, *order2 = nullptr, *rows_final = nullptr, *ids_final = nullptr;
        float *proba_final = nullptr;
        HIP_TRY(s.alloc(&score_cur, n_cur));
        HIP_TRY(s.alloc(&decoy_cur, n_cur));
        hipLaunchKernelGGL((fdrdev::gather_kernel<double>), grid_for(n_cur), dim3(256), 0, st, score, cur, n_cur, score_cur);
        hipLaunchKernelGGL((fdrdev::gather_kernel<uint8_t>), grid_for(n_cur), dim3(256), 0, st, m->d_decoy, cur, n_cur, decoy_cur);
        if (tie_rows) {
            HIP_TRY(s.alloc(&tie_cur, n_cur));
            hipLaunchKernelGGL((fdrdev::gather_kernel<int64_t>), grid_for(n_cur), dim3(256), 0, st, tie_rows, cur, n_cur, tie_cur);
        }
        rc = fdr::q_values_core(h, s, n_cur, score_cur, decoy_cur, tie_cur, &order2, &qv2);
        if (rc != ADH_OK) return rc;
        HIP_TRY(s.alloc(&ids_final, n_cur));
        HIP_TRY(s.alloc(&rows_final, n_cur));
        HIP_TRY(s.alloc(&proba_final, n_cur));
        hipLaunchKernelGGL((fdrdev::gather_kernel<int64_t>), grid_for(n_cur), dim3(256), 0, st, cur, order2, n_cur, ids_final);
        hipLaunchKernelGGL((fdrdev::gather_kernel<int64_t>), grid_for(n_cur), dim3(256), 0, st, m->d_rowmap, ids_final, n_cur, rows_final);
        double *score_final = nullptr;
        HIP_TRY(s.alloc(&score_final, n_cur));
        hipLaunchKernelGGL((fdrdev::gather_kernel<double>), grid_for(n_cur), dim3(256), 0, st, score, ids_final, n_cur, score_final);
        HIP_TRY(hipGetLastError());
        // the only table that leaves the GPU: (candidate row, proba, qval) of the surviving PSMs
        std::vector<double> sc((size_t)n_cur);
        HIP_TRY(hipMemcpyAsync(row_out, rows_final, (size_t)n_cur * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(qval_out, qv2, (size_t)n_cur * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(sc.data(), score_final, (size_t)n_cur * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int64_t i = 0; i < n_cur; ++i) proba_out[i] = (float)sc[(size_t)i];
        (void)proba_final;
        h->d2h_bytes += (uint64_t)n_cur * 24;
    }
    *n_out = n_cur;
    return ADH_OK;
}
