// adh_score_host.hip - host side of the scoring entry points (included by adh_api.hip inside
// its extern "C" block): candidate upload, the device-built plan (adh_plan.hip), kernel
// launches and the chunked host -> host pipeline of adh_score_candidates.
//
// Replaces the harness around the reference's pjit loop:
//   CandidateScoring.__call__            alphadia/search/scoring/scoring.py:582-661
//   assemble_score_group_container       alphadia/search/scoring/scoring.py:273-353
//   _process_score_groups                alphadia/search/scoring/scoring.py:114-137
// Timeline of one adh_score_candidates call (three HIP streams, two plan slots):
//   copy-in : H2D columns(c+1) | plan(c+1)                    (while the kernels of chunk c run)
//   compute : zero tables | gather(c) features(c) | gather(c+1) ...
//   copy-out:                  D2H tables(c) | D2H tables(c+1) ...
// The host only waits for the 128-byte plan record of a chunk (class sizes, LDS capacities,
// validation result) before it launches that chunk's kernels.

namespace {

// adh_comm.hip
int comm_wait_slot(adh_handle *h, int slot);
int comm_gather_slot(adh_handle *h, int slot);

struct OutFieldDesc {
    const char *name;  // the OutputPsmDF column (adh_table_layout)
    size_t member;   // offsetof the pointer inside adh_output_t
    int per_row;     // 1, ADH_NUM_FEATURES, or -1 = top_k
    int elem;        // bytes per element
    bool wire;       // travels in the all-gather and over PCIe (computed tables); the others are rebuilt locally
    bool optional;   // the caller's host pointer may be NULL
};
// The columns that are not `wire` are copies of the candidate table (precursor_idx, rank) and of the library
// (per fragment slot): adh_score_candidates does not copy them back, it rebuilds them in the caller's host
// buffers from fragment_lib_slot with host threads while the D2H copies of later chunks are in flight
// (216 of the 646 bytes per candidate; the link is what bounds the host -> host step).
// stat_matched_peaks is the exception: computed, optional, copied when asked for.

// computed tables first: they form the contiguous "wire" prefix of the packed device buffer
const OutFieldDesc kOutFields[] = {
    {"valid", offsetof(adh_output_t, valid), 1, 1, true, false},
    {"features", offsetof(adh_output_t, features), ADH_NUM_FEATURES, 4, true, false},
    {"fragment_mz_observed", offsetof(adh_output_t, fragment_mz_observed), -1, 4, true, false},
    {"fragment_height", offsetof(adh_output_t, fragment_height), -1, 4, true, false},
    {"fragment_intensity", offsetof(adh_output_t, fragment_intensity), -1, 4, true, false},
    {"fragment_mass_error", offsetof(adh_output_t, fragment_mass_error), -1, 4, true, false},
    {"fragment_correlation", offsetof(adh_output_t, fragment_correlation), -1, 4, true, false},
    {"fragment_lib_slot", offsetof(adh_output_t, fragment_lib_slot), -1, 2, true, true},
    {"precursor_idx", offsetof(adh_output_t, precursor_idx), 1, 4, false, false},
    {"rank", offsetof(adh_output_t, rank), 1, 1, false, false},
    {"fragment_precursor_idx", offsetof(adh_output_t, fragment_precursor_idx), -1, 4, false, false},
    {"fragment_rank", offsetof(adh_output_t, fragment_rank), -1, 1, false, false},
    {"fragment_mz_library", offsetof(adh_output_t, fragment_mz_library), -1, 4, false, false},
    {"fragment_mz", offsetof(adh_output_t, fragment_mz), -1, 4, false, false},
    {"fragment_position", offsetof(adh_output_t, fragment_position), -1, 1, false, false},
    {"fragment_number", offsetof(adh_output_t, fragment_number), -1, 1, false, false},
    {"fragment_type", offsetof(adh_output_t, fragment_type), -1, 1, false, false},
    {"fragment_charge", offsetof(adh_output_t, fragment_charge), -1, 1, false, false},
    {"fragment_loss_type", offsetof(adh_output_t, fragment_loss_type), -1, 1, false, false},
    {"stat_matched_peaks", offsetof(adh_output_t, stat_matched_peaks), 1, 4, false, true},
};
constexpr int kNumOutFields = (int)(sizeof(kOutFields) / sizeof(kOutFields[0]));

inline void **out_member(adh_output_t *o, const OutFieldDesc &f) {
    return reinterpret_cast<void **>(reinterpret_cast<unsigned char *>(o) + f.member);
}
inline size_t out_row_bytes(const OutFieldDesc &f, int top_k) {
    return (size_t)(f.per_row < 0 ? top_k : f.per_row) * (size_t)f.elem;
}

// layout of the packed device tables for `rows` rows: view->... = base + offset; returns total bytes
size_t layout_tables(unsigned char *base, int64_t rows, int top_k, adh_output_t *view, size_t *wire_bytes) {
    size_t off = 0, wire = 0;
    for (int i = 0; i < kNumOutFields; ++i) {
        const OutFieldDesc &f = kOutFields[i];
        if (view) *out_member(view, f) = base ? (void *)(base + off) : nullptr;
        off += ((size_t)rows * out_row_bytes(f, top_k) + 255) / 256 * 256;
        if (f.wire) wire = off;
    }
    if (wire_bytes) *wire_bytes = wire;
    if (view) {
        view->n = rows;
        view->top_k = top_k;
    }
    return std::max<size_t>(off, 256);
}

int ensure_tables(adh_handle *h, int slot, int64_t rows, int top_k) {
    DevTables &t = h->tables[slot];
    const size_t need = layout_tables(nullptr, rows, top_k, nullptr, nullptr);
    if (t.bytes < need) {
        if (t.base) (void)hipFree(t.base);
        t.base = nullptr;
        t.bytes = 0;
        hipError_t e = hipMalloc(&t.base, need);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(ADH_ERR_OUT_OF_MEMORY, std::string("hipMalloc(output tables): ") + hipGetErrorString(e));
        }
        t.bytes = need;
    }
    t.used = layout_tables(static_cast<unsigned char *>(t.base), rows, top_k, &t.view, &t.wire_bytes);
    t.rows = rows;
    t.top_k = top_k;
    return ADH_OK;
}

// ---------------------------------------------------------------- candidate columns in HBM
struct CandColumn {
    const void *host;
    void **dev;
    size_t elem;
};

int cand_reserve(adh_handle *h, int64_t n, int32_t n_iso) {
    CandSlab &s = h->cs;
    const int64_t rows = std::max<int64_t>(n, 1);
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t need = 3 * al((size_t)rows * 4) + 3 * al((size_t)rows) + 6 * al((size_t)rows * 8) +
                        al((size_t)rows * 4) + al((size_t)rows * (size_t)n_iso * 4);
    if (s.bytes < need) {
        HIP_TRY(hipDeviceSynchronize());  // nothing may still read the old slab
        if (s.base) (void)hipFree(s.base);
        s.base = nullptr;
        s.bytes = 0;
        HIP_TRY(hipMalloc(&s.base, need + need / 8));
        s.bytes = need + need / 8;
    }
    unsigned char *p = static_cast<unsigned char *>(s.base);
    auto take = [&](size_t b) {
        unsigned char *q = p;
        p += al(b);
        return q;
    };
    s.d.precursor_idx = (const uint32_t *)take((size_t)rows * 4);
    s.d.frag_start = (const uint32_t *)take((size_t)rows * 4);
    s.d.frag_stop = (const uint32_t *)take((size_t)rows * 4);
    s.d.rank = (const uint8_t *)take((size_t)rows);
    s.flags = (uint8_t *)take((size_t)rows);
    s.d.charge = (const uint8_t *)take((size_t)rows);
    s.d.scan_start = (const int64_t *)take((size_t)rows * 8);
    s.d.scan_stop = (const int64_t *)take((size_t)rows * 8);
    s.d.scan_center = (const int64_t *)take((size_t)rows * 8);
    s.d.frame_start = (const int64_t *)take((size_t)rows * 8);
    s.d.frame_stop = (const int64_t *)take((size_t)rows * 8);
    s.d.frame_center = (const int64_t *)take((size_t)rows * 8);
    s.d.precursor_mz = (const float *)take((size_t)rows * 4);
    s.iso = (float *)take((size_t)rows * (size_t)n_iso * 4);
    s.d.flags = nullptr;
    s.n = n;
    s.n_iso_cols = n_iso;
    return ADH_OK;
}

// H2D of rows [a, b) of every candidate column, asynchronous on `st`
// Copy jobs done by a kernel: the candidate columns of a SMALL range (<= 131 072 rows: the batches of the
// optimisation loop, the short first chunk of a large call) when they sit in page-locked host memory - one
// launch that reads the 14 host columns over PCIe instead of 14 DMA copies.  Measured: 5-10 % on small
// batches (8 000 precursors host -> host 1.45 -> 1.33 ms: fewer driver calls).  Large ranges stay with the
// DMA engine: a copy kernel sits on the CUs for the 4-5 ms the 0.2 GB of a 3 M-candidate step take over PCIe
// and the gather kernel next to it ran 7 % slower, with nothing gained on the step (the host-bound stream
// is the limit either way; round 4 again, for the burst behind the plan of chunk 1 and for per-chunk uploads: the
// scoring kernels 14.8 -> 17.7 ... 18.4 ms, the step 33 -> 37 ms).  ADH_H2D_KERNEL=0 switches the kernel off; pageable columns always use
// hipMemcpyAsync.
struct CopyJobs {
    const unsigned char *src[16];
    unsigned char *dst[16];
    uint64_t bytes[16];
    int n;
};
__global__ __launch_bounds__(256) void adh_copy_jobs_kernel(CopyJobs jobs) {
    const int j = blockIdx.y;
    if (j >= jobs.n) return;
    const unsigned char *src = jobs.src[j];
    unsigned char *dst = jobs.dst[j];
    const uint64_t bytes = jobs.bytes[j];
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15u) == 0) {
        const uint64_t words = bytes / 16;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        for (uint64_t w = tid; w < words; w += stride) d4[w] = s4[w];
        for (uint64_t q = words * 16 + tid; q < bytes; q += stride) dst[q] = src[q];
    } else {
        for (uint64_t q = tid; q < bytes; q += stride) dst[q] = src[q];
    }
}

int cand_upload_range(adh_handle *h, const adh_candidates_t *c, int64_t a, int64_t b, hipStream_t st) {
    CandSlab &s = h->cs;
    if (b <= a) return ADH_OK;
    s.d.flags = c->flags ? s.flags : nullptr;
    const size_t iso_w = (size_t)c->n_isotope_cols * 4;
    const CandColumn cols[] = {
        {c->precursor_idx, (void **)&s.d.precursor_idx, 4}, {c->frag_start_idx, (void **)&s.d.frag_start, 4},
        {c->frag_stop_idx, (void **)&s.d.frag_stop, 4},     {c->rank, (void **)&s.d.rank, 1},
        {c->flags, (void **)&s.flags, 1},                   {c->charge, (void **)&s.d.charge, 1},
        {c->scan_start, (void **)&s.d.scan_start, 8},       {c->scan_stop, (void **)&s.d.scan_stop, 8},
        {c->scan_center, (void **)&s.d.scan_center, 8},     {c->frame_start, (void **)&s.d.frame_start, 8},
        {c->frame_stop, (void **)&s.d.frame_stop, 8},       {c->frame_center, (void **)&s.d.frame_center, 8},
        {c->precursor_mz, (void **)&s.d.precursor_mz, 4},   {c->isotope_intensity, (void **)&s.iso, iso_w},
    };
    // page-locked columns are read by a kernel, pageable ones go through hipMemcpyAsync
    static const bool by_kernel = [] {
        const char *env = getenv("ADH_H2D_KERNEL");
        return !(env && atoi(env) == 0);
    }();
    CopyJobs jobs;
    jobs.n = 0;
    uint64_t most = 0;
    for (const CandColumn &col : cols) {
        if (!col.host) continue;
        unsigned char *dst = static_cast<unsigned char *>(*col.dev) + (size_t)a * col.elem;
        const unsigned char *src = static_cast<const unsigned char *>(col.host) + (size_t)a * col.elem;
        const size_t bytes = (size_t)(b - a) * col.elem;
        void *dev_view = nullptr;
        if (by_kernel && b - a <= 131072) {  // (small ranges only: see above)
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost &&
                hipHostGetDevicePointer(&dev_view, const_cast<unsigned char *>(src), 0) == hipSuccess && dev_view) {
                jobs.src[jobs.n] = static_cast<const unsigned char *>(dev_view);
                jobs.dst[jobs.n] = dst;
                jobs.bytes[jobs.n] = bytes;
                most = std::max<uint64_t>(most, bytes);
                ++jobs.n;
                continue;
            }
            (void)hipGetLastError();
        }
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    }
    if (jobs.n > 0) {
        const unsigned blocks = (unsigned)std::min<uint64_t>(std::max<uint64_t>(most / 16 / 256, 1), 64);
        hipLaunchKernelGGL(adh_copy_jobs_kernel, dim3(blocks, (unsigned)jobs.n), dim3(256), 0, st, jobs);
        HIP_TRY(hipGetLastError());
    }
    return ADH_OK;
}

int check_candidate_args(adh_handle *h, const adh_candidates_t *c) {
    if (!(h->run_staged || h->tims_staged) || !h->lib_staged)
        return fail(ADH_ERR_NOT_STAGED, "stage the run and the fragment library first");
    if (c->n < 0 || c->n_isotope_cols < 1)
        return fail(ADH_ERR_INVALID_ARGUMENT, "invalid candidate table dimensions");
    if (c->n > 0x7FFFFFFFll) return fail(ADH_ERR_UNSUPPORTED, "more than 2^31 candidates in one batch");
    if (c->n > 0 && (!c->precursor_idx || !c->rank || !c->frag_start_idx || !c->frag_stop_idx || !c->scan_start ||
                     !c->scan_stop || !c->scan_center || !c->frame_start || !c->frame_stop || !c->frame_center ||
                     !c->charge || !c->precursor_mz || !c->isotope_intensity))
        return fail(ADH_ERR_INVALID_ARGUMENT, "candidate column is NULL");
    return ADH_OK;
}

// ---------------------------------------------------------------- the plan (adh_plan.hip)
__global__ void adh_plan_init_kernel(PlanMeta *meta, int32_t I) {
    PlanMeta m;
    memset(&m, 0, sizeof(m));
    m.all_k = m.all_o = m.all_f = m.all_n_lib = m.all_s = m.all_op = 1;
    m.gen_k = m.gen_o = m.gen_f = m.gen_n_lib = 1;
    (void)I;
    *meta = m;
}

int plan_reserve(adh_handle *h, PlanSlot &s, int64_t n, bool im) {
    if (!s.done) HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    if (!s.h_meta) HIP_TRY(hipHostMalloc((void **)&s.h_meta, sizeof(PlanMeta), hipHostMallocDefault));
    const size_t rec = im ? sizeof(CandRecIM) : sizeof(CandRec);
    if (s.cap >= n && s.rec_bytes >= rec) return ADH_OK;
    HIP_TRY(hipDeviceSynchronize());
    s.buf.release();
    s.cap = 0;
    const int64_t cap = n + n / 8 + 64;
    auto dev_alloc = [&](void **p, size_t bytes) -> int {
        HIP_TRY(hipMalloc(p, std::max<size_t>(bytes, 256)));
        s.buf.ptrs.push_back(*p);
        return ADH_OK;
    };
    int rc;
    if ((rc = dev_alloc(&s.recs, (size_t)cap * rec)) != ADH_OK) return rc;
    if ((rc = dev_alloc(&s.ordered, (size_t)cap * rec)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&s.keys_in, (size_t)cap * 4)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&s.keys_out, (size_t)cap * 4)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&s.idx_in, (size_t)cap * 4)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&s.idx_out, (size_t)cap * 4)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&s.bytes, (size_t)cap * 8)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&s.sorted_bytes, (size_t)cap * 8)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&s.offs, (size_t)cap * 8)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&s.d_meta, sizeof(PlanMeta))) != ADH_OK) return rc;
    size_t b_sort = 0, b_scan = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, b_sort, s.keys_in, s.keys_out, s.idx_in, s.idx_out, (int)cap, 0, 32,
                                               h->stream));
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, b_scan, s.sorted_bytes, s.offs, (int)cap, h->stream));
    s.cub_bytes = std::max(b_sort, b_scan);
    if ((rc = dev_alloc(&s.cub_tmp, s.cub_bytes)) != ADH_OK) return rc;
    s.cap = cap;
    s.rec_bytes = rec;
    return ADH_OK;
}

struct PlanKey {
    uint32_t top_k_fragments, top_k_isotopes;
    bool fast_cfg, quant_all, fused_cfg, wide_cfg;
};

PlanKey plan_key(const adh_scoring_config_t *cfg) {
    const bool fast = cfg->experimental_xic != 0 && !getenv("ADH_DEBUG_NO_FAST");
    return PlanKey{cfg->top_k_fragments, cfg->top_k_isotopes, fast, cfg->quant_all != 0,
                   fast && !getenv("ADH_DEBUG_NO_FUSED"),   // developer switches: two-kernel path only
                   fast && !getenv("ADH_DEBUG_NO_WIDE")};   // ... more than 16 kept fragments through the generic kernel
}

// enqueue the plan of rows [row0, row0 + n) on `st`; plan_finish() waits for it
int plan_enqueue(adh_handle *h, PlanSlot &s, const adh_scoring_config_t *cfg, int64_t row0, int64_t n, hipStream_t st) {
    const bool im = h->tims_staged;
    int rc = plan_reserve(h, s, n, im);
    if (rc != ADH_OK) return rc;
    const PlanKey key = plan_key(cfg);
    PlanArgs p{};
    p.row0 = row0;
    p.n = n;
    p.n_lib = h->n_lib;
    p.I = (int32_t)std::min<uint32_t>(cfg->top_k_isotopes, (uint32_t)h->cs.n_iso_cols);
    p.top_k = cfg->top_k_fragments;
    p.fast_cfg = (key.fast_cfg ? 1 : 0) | (key.wide_cfg ? 2 : 0);
    p.quant_all = key.quant_all ? 1 : 0;
    p.fused_cfg = (key.fused_cfg && !im && h->run.n_ms1_obs == 1 && p.I <= 4) ? (getenv("ADH_DEBUG_NO_FUSED2") ? 1 : 3) : 0;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(adh_plan_init_kernel, dim3(1), dim3(1), 0, st, s.d_meta, p.I);
    const int64_t n_frames = im ? h->tims.n_frames : h->run.n_spectra;
    const int L = im ? h->tims.cycle_len : h->run.cycle_len;
    p.L = L;
    p.n_frames = n_frames;
    p.n_cyc_bins = (int32_t)std::min<int64_t>(n_frames / L + 2, 1 << 26);
    // counting sort when the key space is small next to the batch (the usual case: 18 classes x
    // cycles of the run); a stable radix sort otherwise
    const int64_t n_keys = (int64_t)ADH_N_CLASSES * p.n_cyc_bins;
    const bool counting = n_keys <= (1 << 22) && !getenv("ADH_DEBUG_PLAN_RADIX");
    if (counting) {
        if (s.hist_cap < n_keys + 1) {
            HIP_TRY(hipDeviceSynchronize());
            if (s.hist) (void)hipFree(s.hist);
            s.hist = nullptr;
            s.hist_cap = 0;
            HIP_TRY(hipMalloc((void **)&s.hist, (size_t)(n_keys + 1) * 4));
            s.hist_cap = n_keys + 1;
            size_t b = 0;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, b, s.hist, s.hist, (int)(n_keys + 1), st));
            if (b > s.cub_bytes) {  // (the scratch was sized for the per-candidate scans)
                void *tmp = nullptr;
                HIP_TRY(hipMalloc(&tmp, b));
                s.buf.ptrs.push_back(tmp);
                s.cub_tmp = tmp;
                s.cub_bytes = b;
            }
        }
        HIP_TRY(hipMemsetAsync(s.hist, 0, (size_t)(n_keys + 1) * 4, st));
    }
    uint32_t *hist = counting ? s.hist : nullptr;
    if (im) {
        p.rows = h->tims.cycle_len;
        p.scan_max = h->tims.scan_max;
        p.zeroth = h->tims.zeroth;
        hipLaunchKernelGGL(adh_plan_rec_im_kernel, dim3(blocks), dim3(256), 0, st, h->cs.d, h->tims.cycle, h->tims.dpc, p,
                           static_cast<CandRecIM *>(s.recs), s.keys_in, s.idx_in, s.bytes, hist, s.d_meta);
    } else {
        p.rows = h->run.cycle_len * h->run.cycle_scans;
        hipLaunchKernelGGL(adh_plan_rec_kernel, dim3(blocks), dim3(256), 0, st, h->cs.d, h->run.cycle, p,
                           static_cast<CandRec *>(s.recs), s.keys_in, s.idx_in, s.bytes, hist, s.d_meta);
    }
    HIP_TRY(hipGetLastError());
    size_t tb = s.cub_bytes;
    if (counting) {
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(s.cub_tmp, tb, s.hist, s.hist, (int)(n_keys + 1), st));
        hipLaunchKernelGGL(adh_plan_scatter_kernel, dim3(blocks), dim3(256), 0, st, s.keys_in, s.bytes, n, s.hist, s.keys_out,
                           s.idx_out, s.sorted_bytes);
    } else {
        int end_bit = 1;
        while (end_bit < 32 && (1ull << end_bit) < (uint64_t)n_keys) ++end_bit;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(s.cub_tmp, tb, s.keys_in, s.keys_out, s.idx_in, s.idx_out, (int)n, 0,
                                                   end_bit, st));
        hipLaunchKernelGGL(adh_plan_take_bytes_kernel, dim3(blocks), dim3(256), 0, st, s.bytes, s.idx_out, n, s.sorted_bytes);
    }
    tb = s.cub_bytes;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(s.cub_tmp, tb, s.sorted_bytes, s.offs, (int)n, st));
    if (im)
        hipLaunchKernelGGL((adh_plan_order_kernel<CandRecIM>), dim3(blocks), dim3(256), 0, st,
                           static_cast<const CandRecIM *>(s.recs), s.keys_out, s.idx_out, s.offs, s.sorted_bytes, n,
                           (uint32_t)p.n_cyc_bins, static_cast<CandRecIM *>(s.ordered), s.d_meta);
    else
        hipLaunchKernelGGL((adh_plan_order_kernel<CandRec>), dim3(blocks), dim3(256), 0, st,
                           static_cast<const CandRec *>(s.recs), s.keys_out, s.idx_out, s.offs, s.sorted_bytes, n,
                           (uint32_t)p.n_cyc_bins, static_cast<CandRec *>(s.ordered), s.d_meta);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(s.h_meta, s.d_meta, sizeof(PlanMeta), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(s.done, st));
    return ADH_OK;
}

int plan_finish(adh_handle *h, PlanSlot &s, const adh_scoring_config_t *cfg, int64_t row0, int64_t n, Plan &p) {
    HIP_TRY(hipEventSynchronize(s.done));
    const PlanMeta &m = *s.h_meta;
    switch (m.err) {
        case ADH_PLAN_OK: break;
        case ADH_PLAN_ERR_FRAG_SLICE: return fail(ADH_ERR_INVALID_ARGUMENT, "fragment slice outside the staged library");
        case ADH_PLAN_ERR_FRAME_LIMITS: return fail(ADH_ERR_INVALID_ARGUMENT, "frame limits outside the staged run");
        case ADH_PLAN_ERR_CYCLE_BOUNDARY: return fail(ADH_ERR_INVALID_ARGUMENT, "frame_start must sit on a cycle boundary");
        case ADH_PLAN_ERR_SCAN_LIMITS: return fail(ADH_ERR_INVALID_ARGUMENT, "scan limits outside the staged run");
        case ADH_PLAN_ERR_ALPHARAW_SCANS:
            return fail(ADH_ERR_UNSUPPORTED, "AlphaRaw candidates must have scan_start=0, scan_stop=1, scan_center=0");
        case ADH_PLAN_ERR_CHARGE: return fail(ADH_ERR_INVALID_ARGUMENT, "precursor charge is 0");
        case ADH_PLAN_ERR_TOO_MANY_OBS:
            return fail(ADH_ERR_UNSUPPORTED, h->tims_staged ? "a precursor overlaps more than 8 cycle rows"
                                                            : "a precursor overlaps more than 8 isolation windows");
        default: return fail(ADH_ERR_UNSUPPORTED, "more than 16 unfragmented cycle rows in the scan range");
    }
    const PlanKey key = plan_key(cfg);
    const int I = (int)std::min<uint32_t>(cfg->top_k_isotopes, (uint32_t)h->cs.n_iso_cols);
    p = Plan();
    p.row0 = row0;
    p.n = n;
    p.d_recs = h->tims_staged ? nullptr : static_cast<CandRec *>(s.ordered);
    p.d_recs_im = h->tims_staged ? static_cast<CandRecIM *>(s.ordered) : nullptr;
    for (int c = 0; c < ADH_N_CLASSES; ++c) p.n_class[c] = (int64_t)m.class_first[c + 1] - (int64_t)m.class_first[c];
    p.caps_all = Caps{m.all_k, m.all_o, m.all_f, std::max(I, 1), m.all_n_lib, 0, m.all_s, m.all_op};
    p.caps_generic = h->tims_staged ? p.caps_all : Caps{m.gen_k, m.gen_o, m.gen_f, std::max(I, 1), m.gen_n_lib, 0, 1, 1};
    p.scratch_bytes = std::max<uint64_t>(m.scratch_bytes, 32);
    p.top_k_fragments = key.top_k_fragments;
    p.top_k_isotopes = key.top_k_isotopes;
    p.fast_ok = key.fast_cfg;
    p.fused_ok = key.fused_cfg;
    p.wide_ok = key.wide_cfg;
    p.quant_all = key.quant_all;
    p.ready = true;
    return ADH_OK;
}

// the scratch slab is grow-only and shared by all chunks (their kernels are serialised on one stream)
// (see DevBlockCache in adh_api.hip)
void settle_copy_path(adh_handle *h) {
    // Only on request (ADH_COPY_PATH_RESET=1) and only after a block of 1 GB or more really went back to the runtime
    // (DevBlockCache in adh_api.hip parks them, so that needs a cache overflow or adh_trim_device_cache): page-locking
    // 2 GB of host memory once brings the DMA copies back to the full link rate on this ROCm (tools/probes/d2h_pattern.hip)
    (void)h;
    static const bool on = [] {
        const char *env = getenv("ADH_COPY_PATH_RESET");
        return env && atoi(env) != 0;
    }();
    if (!on || !g_big_free.exchange(false)) return;
    void *p = nullptr;
    if (hipHostMalloc(&p, (size_t)2 << 30, hipHostMallocPortable) == hipSuccess && p) (void)hipHostFree(p);
    (void)hipGetLastError();  // (no room for it: the copies stay slow, nothing else changes)
}

int ensure_scratch(adh_handle *h, uint64_t bytes) {
    if (h->scratch_slab_bytes >= bytes) return ADH_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (h->scratch_slab) (void)hipFree(h->scratch_slab);
    h->scratch_slab = nullptr;
    h->scratch_slab_bytes = 0;
    const uint64_t want = bytes + bytes / 4;
    HIP_TRY(hipMalloc(&h->scratch_slab, want));
    h->scratch_slab_bytes = want;
    return ADH_OK;
}

// fold finished timing triples into the running sums so that the event list stays short
void fold_timed(adh_handle *h, bool wait) {
    size_t keep = 0;
    for (size_t i = 0; i < h->timed.size(); ++i) {
        adh_handle::Timed &t = h->timed[i];
        bool done = wait ? (hipEventSynchronize(t.e2) == hipSuccess) : (hipEventQuery(t.e2) == hipSuccess);
        float a = 0, b = 0;
        if (done && hipEventElapsedTime(&a, t.e0, t.e1) == hipSuccess && hipEventElapsedTime(&b, t.e1, t.e2) == hipSuccess) {
            h->sum_gather_ms += a;
            h->sum_feature_ms += b;
            ++h->n_timed;
            h->free_events.push_back(t.e0);
            h->free_events.push_back(t.e1);
            h->free_events.push_back(t.e2);
        } else {
            h->timed[keep++] = t;
        }
    }
    (void)hipGetLastError();
    h->timed.resize(keep);
}

int launch_scoring_im(adh_handle *h, Plan &p, const adh_scoring_config_t *cfg, adh_output_t *out, hipStream_t st) {
    if (cfg->collect_fragments && p.caps_all.k > out->top_k)
        return fail(ADH_ERR_INVALID_ARGUMENT, "output top_k smaller than config.top_k_fragments");
    p.caps_all.stop_phase = 0;
    if (const char *dbg = getenv("ADH_DEBUG_IM")) p.caps_all.stop_phase = atoi(dbg);
    p.caps_all.dbg_drop_dense = 0;
    if (const char *dbg = getenv("ADH_DEBUG_IM_DROP_DENSE")) p.caps_all.dbg_drop_dense = atoi(dbg);
    size_t g_pad = 0, f_pad = 0;  // developer switches: extra LDS per block, to see what occupancy is worth
    if (const char *dbg = getenv("ADH_DEBUG_IM_GATHER_LDS_PAD")) g_pad = (size_t)atoi(dbg);
    if (const char *dbg = getenv("ADH_DEBUG_IM_FEATURE_LDS_PAD")) f_pad = (size_t)atoi(dbg);
    const size_t g_lds = adh_gather_im_lds_bytes(p.caps_all) + g_pad;
    const size_t f_lds = adh_feature_im_lds_bytes(p.caps_all) + f_pad;
    if (f_lds > 160 * 1024 - ADH_IM_STATIC_LDS || g_lds > 160 * 1024) {
        char buf[256];
        snprintf(buf, sizeof(buf),
                 "ion-mobility tile needs %zu bytes of LDS (K=%d O=%d S=%d F=%d): exceeds 160 KiB", f_lds,
                 p.caps_all.k, p.caps_all.o, p.caps_all.s, p.caps_all.f);
        return fail(ADH_ERR_UNSUPPORTED, buf);
    }
    // Split feature path (adh_features_im2.hip): the one-observation classes with a fixed layout leave a profile
    // record per candidate behind the scratch blocks, and a second kernel with four candidates per wavefront
    // finishes them.  ADH_DEBUG_IM_NO_SPLIT=1: the one-kernel path for everything (the GPU suite holds both to
    // identical bits); the developer ablations (ADH_DEBUG_IM stops, layout switch) belong to the one-kernel path.
    const int32_t n_iso = h->cs.n_iso_cols;
    const bool fixed_layouts = !getenv("ADH_DEBUG_IM_DYNAMIC_LAYOUT");
    const bool split_cfg = fixed_layouts && cfg->experimental_xic && (p.caps_all.stop_phase == 0 || p.caps_all.stop_phase == 8) &&
                           !getenv("ADH_DEBUG_IM_NO_SPLIT") &&  // (8: the gather's dense-tiles switch, no feature-kernel stop)
                           std::min<uint32_t>(cfg->top_k_isotopes, (uint32_t)n_iso) <= 3;
    auto class_caps = [&](int c) {
        Caps cc = p.caps_all;
        if (c == 0 || c == ADH_CLASS_IM_SMALL) cc.o = 1;
        if (c == 1) cc.o = std::min(cc.o, 2);
        if (c == ADH_CLASS_IM_SMALL) {  // (what the plan admitted to the class: adh_plan_rec_im_kernel)
            cc.k = std::min(cc.k, ADH_IM_SMALL_K);
            cc.s = std::min(cc.s, ADH_IM_SMALL_S);
            cc.f = std::min(cc.f, ADH_IM_SMALL_F);
        }
        return cc;
    };
    typedef ImProfRec<featim::DimsCommon::Fc, featim::DimsCommon::Sc> ProfCommon;
    typedef ImProfRec<featim::DimsSmall::Fc, featim::DimsSmall::Sc> ProfSmall;
    const bool split_small = split_cfg && p.n_class[ADH_CLASS_IM_SMALL] > 0 && featim::DimsSmall::holds_axes(class_caps(ADH_CLASS_IM_SMALL));
    const bool split_common = split_cfg && p.n_class[0] > 0 && featim::DimsCommon::holds(class_caps(0)) && class_caps(0).f >= 3;
    typedef ImProfRec<featim::DimsCommon2::Fc, featim::DimsCommon2::Sc, 2> ProfCommon2;
    const bool split_two = split_cfg && p.n_class[1] > 0 && featim::DimsCommon2::holds(class_caps(1)) && class_caps(1).f >= 3 &&
                           !getenv("ADH_DEBUG_IM_NO_SPLIT2");
    const uint64_t prof_base = (p.scratch_bytes + 255) / 256 * 256;
    const uint64_t prof_small_off = prof_base + (split_common ? (uint64_t)p.n_class[0] * sizeof(ProfCommon) : 0);
    const uint64_t prof_two_off = prof_small_off + (split_small ? (uint64_t)p.n_class[ADH_CLASS_IM_SMALL] * sizeof(ProfSmall) : 0);
    const uint64_t prof_end = prof_two_off + (split_two ? (uint64_t)p.n_class[1] * sizeof(ProfCommon2) : 0);
    // Round 6: the one-observation split classes go through adh_feature_im_tile4_kernel (four candidates per
    // wavefront, adh_features_im4.hip); the candidates it leaves aside (materialised tiles) are listed behind the
    // records - per class a counter and the plan positions - and taken by the first blocks of the same grid with the old body.
    // ADH_DEBUG_IM_TILE1=1: the one-candidate-per-wavefront tile kernel everywhere.
    const bool tile4 = !getenv("ADH_DEBUG_IM_TILE1");
    const int32_t stop4 = getenv("ADH_DEBUG_IM4") ? atoi(getenv("ADH_DEBUG_IM4")) : 0;  // developer ablation (tile4_phase)
    const uint64_t list_common_off = (prof_end + 255) / 256 * 256;
    const uint64_t list_small_off = list_common_off + (tile4 && split_common ? featim4::side_bytes(p.n_class[0]) : 0);
    const uint64_t list_two_off = list_small_off + (tile4 && split_small ? featim4::side_bytes(p.n_class[ADH_CLASS_IM_SMALL]) : 0);
    const bool fuse4 = tile4 && !getenv("ADH_DEBUG_IM_NO_FUSE4") && stop4 == 0;  // (tile + profile phase in one kernel, one observation)
    const bool tile4_two = tile4 && !getenv("ADH_DEBUG_IM_TILE1_TWO");  // (A/B: the two-observation class through the one-candidate kernel)
    const uint64_t prof_bytes = list_two_off + (tile4_two && split_two ? featim4::side_bytes(p.n_class[1]) : 0);
    int rc = ensure_scratch(h, prof_bytes);
    if (rc != ADH_OK) return rc;
    unsigned char *d_scratch = static_cast<unsigned char *>(h->scratch_slab);
    adh_handle::Timed t;
    rc = get_event(h, &t.e0);
    if (rc == ADH_OK) rc = get_event(h, &t.e1);
    if (rc == ADH_OK) rc = get_event(h, &t.e2);
    if (rc != ADH_OK) return rc;
    HIP_TRY(hipEventRecord(t.e0, st));
    DevTims run_view = h->tims;
    if (getenv("ADH_DEBUG_IM_NO_TILES")) run_view.tile_ev = nullptr;  // A/B: the (window, TOF bin) ranges of the TOF-major events
    hipLaunchKernelGGL(adh_gather_im_kernel, dim3((unsigned)p.n), dim3(ADH_WAVE), g_lds, st, run_view, h->d_lib,
                       p.d_recs_im, *cfg, n_iso, d_scratch, *out, p.caps_all);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(t.e1, st));
    {
        // one feature launch per observation class (adh_plan_rec_im_kernel): 1, 2, more
        int64_t first = 0;
        for (int c = 0; c < ADH_N_CLASSES; ++c) {
            const int64_t cnt = p.n_class[c];
            if (cnt > 0) {
                const Caps cc = class_caps(c);
                // (capacities fixed at compile time for the common shape: adh_features_im.hip, DimsFix)
                const bool fixed_ok = fixed_layouts;
                const bool common = featim::DimsCommon::holds(cc) && fixed_ok;
                const unsigned groups = (unsigned)((cnt + ADH_WAVE / 16 - 1) / (ADH_WAVE / 16));
                const unsigned list_blocks = (unsigned)std::min<int64_t>(cnt, 1024);  // (blocks that take the materialised tiles in turn)
                if (c == ADH_CLASS_IM_SMALL && split_small) {
                    unsigned char *prof = d_scratch + prof_small_off;
                    if (tile4) {
                        uint32_t *side = reinterpret_cast<uint32_t *>(d_scratch + list_small_off);
                        const unsigned ob = (unsigned)((cnt + 255) / 256);
                        HIP_TRY(hipMemsetAsync(side, 0, featim4::SIDE_HEAD * 4, st));
                        hipLaunchKernelGGL(adh_im_order_hist_kernel, dim3(ob), dim3(256), 0, st, p.d_recs_im + first, (int32_t)cnt,
                                           d_scratch, side);
                        hipLaunchKernelGGL(adh_im_order_scatter_kernel, dim3(ob), dim3(256), 0, st, p.d_recs_im + first, (int32_t)cnt,
                                           d_scratch, side);
                        if (fuse4) {
                            const size_t f4_lds = std::max({sizeof(featim4::WaveTile<featim::DimsSmall::Fc, featim::DimsSmall::Sc, 1>),
                                                            sizeof(featim2::GroupLds<featim::DimsSmall::Fc, featim::DimsSmall::Sc, 1>) * (ADH_WAVE / 16),
                                                            featim::LayoutSmall(cc).bytes() + f_pad + ADH_IM_STATIC_LDS + 16});
                            hipLaunchKernelGGL((adh_feature_im_fused4_kernel<featim::DimsSmall::Fc, featim::DimsSmall::Sc, featim::LayoutSmall>),
                                               dim3(groups + list_blocks), dim3(ADH_WAVE), f4_lds, st, h->tims, p.d_recs_im + first,
                                               (int32_t)cnt, h->cs.iso, n_iso, *cfg, d_scratch, *out, side, cc, (int32_t)list_blocks);
                            HIP_TRY(hipGetLastError());
                            first += cnt;
                            continue;
                        }
                        const size_t t4_lds = std::max(sizeof(featim4::WaveTile<featim::DimsSmall::Fc, featim::DimsSmall::Sc, 1>), featim::LayoutSmall(cc).bytes() + f_pad + ADH_IM_STATIC_LDS + 16);
                        hipLaunchKernelGGL((adh_feature_im_tile4_kernel<featim::DimsSmall::Fc, featim::DimsSmall::Sc, 1, featim::LayoutSmall>),
                                           dim3(groups + list_blocks), dim3(ADH_WAVE), t4_lds, st, h->tims, p.d_recs_im + first,
                                           (int32_t)cnt, h->cs.iso, n_iso, *cfg, d_scratch, *out, prof, side, cc,
                                           (int32_t)list_blocks, stop4);
                    } else
                    hipLaunchKernelGGL((adh_feature_im_kernel<featim::LayoutSmall, true>), dim3((unsigned)cnt), dim3(ADH_WAVE),
                                       featim::LayoutSmall(cc).bytes() + f_pad, st, h->tims, p.d_recs_im + first, h->cs.iso, n_iso,
                                       *cfg, d_scratch, *out, cc, prof);
                    hipLaunchKernelGGL((adh_feature_im_profiles_kernel<featim::DimsSmall::Fc, featim::DimsSmall::Sc, 1>), dim3(groups),
                                       dim3(ADH_WAVE), 0, st, h->tims, p.d_recs_im + first, (int32_t)cnt, *cfg, n_iso, d_scratch,
                                       prof, *out);
                } else if (c == 1 && split_two) {
                    unsigned char *prof = d_scratch + prof_two_off;
                    if (tile4_two) {
                        uint32_t *side = reinterpret_cast<uint32_t *>(d_scratch + list_two_off);
                        const unsigned ob = (unsigned)((cnt + 255) / 256);
                        HIP_TRY(hipMemsetAsync(side, 0, featim4::SIDE_HEAD * 4, st));
                        hipLaunchKernelGGL(adh_im_order_hist_kernel, dim3(ob), dim3(256), 0, st, p.d_recs_im + first, (int32_t)cnt,
                                           d_scratch, side);
                        hipLaunchKernelGGL(adh_im_order_scatter_kernel, dim3(ob), dim3(256), 0, st, p.d_recs_im + first, (int32_t)cnt,
                                           d_scratch, side);
                        // (ADH_DEBUG_IM_TILE4_TWO_LDS_PAD: extra LDS for this launch alone, to see what its occupancy is worth)
                        const size_t t42_pad = getenv("ADH_DEBUG_IM_TILE4_TWO_LDS_PAD") ? (size_t)atoi(getenv("ADH_DEBUG_IM_TILE4_TWO_LDS_PAD")) : 0;
                        const size_t t4_lds = std::max(sizeof(featim4::WaveTile<featim::DimsCommon2::Fc, featim::DimsCommon2::Sc, 2>),
                                                       featim::LayoutCommon2(cc).bytes() + f_pad + ADH_IM_STATIC_LDS + 16) + t42_pad;
                        hipLaunchKernelGGL((adh_feature_im_tile4_kernel<featim::DimsCommon2::Fc, featim::DimsCommon2::Sc, 2, featim::LayoutCommon2>),
                                           dim3(groups + list_blocks), dim3(ADH_WAVE), t4_lds, st, h->tims, p.d_recs_im + first,
                                           (int32_t)cnt, h->cs.iso, n_iso, *cfg, d_scratch, *out, prof, side, cc,
                                           (int32_t)list_blocks, stop4);
                    } else
                    hipLaunchKernelGGL((adh_feature_im_kernel<featim::LayoutCommon2, true>), dim3((unsigned)cnt), dim3(ADH_WAVE),
                                       featim::LayoutCommon2(cc).bytes() + f_pad, st, h->tims, p.d_recs_im + first, h->cs.iso, n_iso,
                                       *cfg, d_scratch, *out, cc, prof);
                    hipLaunchKernelGGL((adh_feature_im_profiles_kernel<featim::DimsCommon2::Fc, featim::DimsCommon2::Sc, 2>),
                                       dim3(groups), dim3(ADH_WAVE), 0, st, h->tims, p.d_recs_im + first, (int32_t)cnt, *cfg, n_iso,
                                       d_scratch, prof, *out);
                } else if (c == 0 && split_common) {
                    unsigned char *prof = d_scratch + prof_base;
                    if (tile4) {
                        uint32_t *side = reinterpret_cast<uint32_t *>(d_scratch + list_common_off);
                        const unsigned ob = (unsigned)((cnt + 255) / 256);
                        HIP_TRY(hipMemsetAsync(side, 0, featim4::SIDE_HEAD * 4, st));
                        hipLaunchKernelGGL(adh_im_order_hist_kernel, dim3(ob), dim3(256), 0, st, p.d_recs_im + first, (int32_t)cnt,
                                           d_scratch, side);
                        hipLaunchKernelGGL(adh_im_order_scatter_kernel, dim3(ob), dim3(256), 0, st, p.d_recs_im + first, (int32_t)cnt,
                                           d_scratch, side);
                        if (fuse4) {
                            const size_t f4_lds = std::max({sizeof(featim4::WaveTile<featim::DimsCommon::Fc, featim::DimsCommon::Sc, 1>),
                                                            sizeof(featim2::GroupLds<featim::DimsCommon::Fc, featim::DimsCommon::Sc, 1>) * (ADH_WAVE / 16),
                                                            featim::LayoutCommon(cc).bytes() + f_pad + ADH_IM_STATIC_LDS + 16});
                            hipLaunchKernelGGL((adh_feature_im_fused4_kernel<featim::DimsCommon::Fc, featim::DimsCommon::Sc, featim::LayoutCommon>),
                                               dim3(groups + list_blocks), dim3(ADH_WAVE), f4_lds, st, h->tims, p.d_recs_im + first,
                                               (int32_t)cnt, h->cs.iso, n_iso, *cfg, d_scratch, *out, side, cc, (int32_t)list_blocks);
                            HIP_TRY(hipGetLastError());
                            first += cnt;
                            continue;
                        }
                        const size_t t4_lds = std::max(sizeof(featim4::WaveTile<featim::DimsCommon::Fc, featim::DimsCommon::Sc, 1>), featim::LayoutCommon(cc).bytes() + f_pad + ADH_IM_STATIC_LDS + 16);
                        hipLaunchKernelGGL((adh_feature_im_tile4_kernel<featim::DimsCommon::Fc, featim::DimsCommon::Sc, 1, featim::LayoutCommon>),
                                           dim3(groups + list_blocks), dim3(ADH_WAVE), t4_lds, st, h->tims, p.d_recs_im + first,
                                           (int32_t)cnt, h->cs.iso, n_iso, *cfg, d_scratch, *out, prof, side, cc,
                                           (int32_t)list_blocks, stop4);
                    } else
                    hipLaunchKernelGGL((adh_feature_im_kernel<featim::LayoutCommon, true>), dim3((unsigned)cnt), dim3(ADH_WAVE),
                                       featim::LayoutCommon(cc).bytes() + f_pad, st, h->tims, p.d_recs_im + first, h->cs.iso, n_iso,
                                       *cfg, d_scratch, *out, cc, prof);
                    hipLaunchKernelGGL((adh_feature_im_profiles_kernel<featim::DimsCommon::Fc, featim::DimsCommon::Sc, 1>), dim3(groups),
                                       dim3(ADH_WAVE), 0, st, h->tims, p.d_recs_im + first, (int32_t)cnt, *cfg, n_iso, d_scratch,
                                       prof, *out);
                } else if (c == ADH_CLASS_IM_SMALL && featim::DimsSmall::holds_axes(cc) && fixed_ok)
                    hipLaunchKernelGGL(adh_feature_im_kernel<featim::LayoutSmall>, dim3((unsigned)cnt), dim3(ADH_WAVE),
                                       featim::LayoutSmall(cc).bytes() + f_pad, st, h->tims, p.d_recs_im + first, h->cs.iso, n_iso,
                                       *cfg, d_scratch, *out, cc);
                else if (common)
                    hipLaunchKernelGGL(adh_feature_im_kernel<featim::LayoutCommon>, dim3((unsigned)cnt), dim3(ADH_WAVE),
                                       featim::LayoutCommon(cc).bytes() + f_pad, st, h->tims, p.d_recs_im + first, h->cs.iso, n_iso,
                                       *cfg, d_scratch, *out, cc);
                else
                    hipLaunchKernelGGL(adh_feature_im_kernel<featim::Layout>, dim3((unsigned)cnt), dim3(ADH_WAVE),
                                       featim::Layout(cc).bytes() + f_pad, st, h->tims, p.d_recs_im + first, h->cs.iso, n_iso, *cfg,
                                       d_scratch, *out, cc);
                HIP_TRY(hipGetLastError());
            }
            first += cnt;
        }
    }
    HIP_TRY(hipEventRecord(t.e2, st));
    h->timed.push_back(t);
    return ADH_OK;
}

// enqueue gather + feature kernels of one planned batch on `st`
int launch_scoring(adh_handle *h, Plan &p, const adh_scoring_config_t *cfg, adh_output_t *out, hipStream_t st) {
    if (h->timed.size() > 48) fold_timed(h, false);
    if (p.n == 0) return ADH_OK;
    if (h->tims_staged) return launch_scoring_im(h, p, cfg, out, st);
    if (cfg->collect_fragments && p.caps_all.k > out->top_k)
        return fail(ADH_ERR_INVALID_ARGUMENT, "output top_k smaller than config.top_k_fragments");
    int stop_phase = 0;
    if (const char *dbg = getenv("ADH_DEBUG_STOP_PHASE")) stop_phase = atoi(dbg);  // developer switch

    Caps gcaps = p.caps_all;
    if (const char *dbg = getenv("ADH_DEBUG_GATHER")) gcaps.stop_phase = atoi(dbg);
    const size_t g_lds = adh_gather_lds_bytes(gcaps, h->run.n_ms1_obs);
    p.caps_generic.stop_phase = stop_phase;
    const size_t f_lds = adh_feature_lds_bytes(p.caps_generic);
    if (p.n_class[ADH_CLASS_GENERIC] > 0 && f_lds > 160 * 1024) {
        char buf[256];
        snprintf(buf, sizeof(buf), "candidate tile needs %zu bytes of LDS (K=%d O=%d F=%d): exceeds 160 KiB", f_lds,
                 p.caps_generic.k, p.caps_generic.o, p.caps_generic.f);
        return fail(ADH_ERR_UNSUPPORTED, buf);
    }
    if (g_lds > 160 * 1024) return fail(ADH_ERR_UNSUPPORTED, "library slice / MS1 tile too large for the gather kernel");
    int rc = ensure_scratch(h, p.scratch_bytes);
    if (rc != ADH_OK) return rc;
    unsigned char *d_scratch = static_cast<unsigned char *>(h->scratch_slab);

    adh_handle::Timed t;
    rc = get_event(h, &t.e0);
    if (rc == ADH_OK) rc = get_event(h, &t.e1);
    if (rc == ADH_OK) rc = get_event(h, &t.e2);
    if (rc != ADH_OK) return rc;
    const int32_t n_iso = h->cs.n_iso_cols;
    int64_t n_fused = 0;  // classes of the fused kernel come first in processing order
    for (int c = ADH_CLASS_FUSED0; c < ADH_CLASS_FAST2; ++c) n_fused += p.n_class[c];
    const char *only = getenv("ADH_DEBUG_ONLY");  // developer switch: "fast" / "generic" / "u" = fused kernels only
    const bool run_generic = !(only && (only[0] == 'f' || only[0] == 'u')), run_fast = !(only && only[0] == 'g');
    const bool fused_only = only && only[0] == 'u';
    HIP_TRY(hipEventRecord(t.e0, st));
    if (p.n > n_fused) {
        hipLaunchKernelGGL(adh_gather_kernel, dim3((unsigned)(p.n - n_fused)), dim3(ADH_WAVE), g_lds, st, h->run, h->d_lib,
                           p.d_recs + n_fused, *cfg, n_iso, d_scratch, *out, gcaps);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(t.e1, st));
    const unsigned per_block = ADH_WAVE / ADH_GS;
    if (n_fused > 0 && run_fast) {
        // per observation count: the shorter rows in one launch, the longest (more LDS; with one observation also
        // more registers than three wavefronts per SIMD leave: adh_fused.hip) in a second one
        FusedClasses fcs[4] = {};  // [2 * (observations - 1) + (long rows)]
        int64_t fblocks[4] = {0, 0, 0, 0};
        int64_t first = 0;
        for (int c = ADH_CLASS_FUSED0; c < ADH_CLASS_FAST2; ++c) {
            const int64_t nc = p.n_class[c];
            const int obs = (c - ADH_CLASS_FUSED0) / 7, fm = 8 + 4 * ((c - ADH_CLASS_FUSED0) % 7);
            // one observation: rows up to ADH_FUSED_FM3 cycles at three wavefronts per SIMD, longer ones at two;
            // two observations: ONE launch (round 5: rows of 32 cycles used to have their own - 190 wavefronts per
            // 47 000-row chunk that kept the GPU for a whole wavefront lifetime, 73 us; at two wavefronts per SIMD
            // the larger LDS block of FM = 32 costs no occupancy)
            const int which = obs == 0 ? (fm > ADH_FUSED_FM3 ? 1 : 0) : 2;
            FusedClasses &fc = fcs[which];
            if (nc > 0) {
                fc.first_block[fc.n] = (int32_t)fblocks[which];
                fc.first_cand[fc.n] = (int32_t)first;
                fc.n_cand[fc.n] = (int32_t)nc;
                fc.kind[fc.n] = c - ADH_CLASS_FUSED0;
                ++fc.n;
                fblocks[which] += (nc + per_block - 1) / per_block;
                fc.first_block[fc.n] = (int32_t)fblocks[which];
            }
            first += nc;
        }
        // tile width: 15 columns while lane 15 carries no isotope, 16 with four isotopes (adh_fused.hip)
        const bool wide = std::min<uint32_t>(cfg->top_k_isotopes, (uint32_t)n_iso) > 3;
        // The launches of a chunk write disjoint rows and read nothing of each other, and the smaller ones (long rows,
        // two observations: 10-20 % of a chunk each) keep the GPU for a wavefront lifetime or two (70-100 us) that
        // they do not fill.  Running them side by side was measured in round 5 and is OFF: on side streams that fork
        // from the caller's and join it again (ADH_FUSED_STREAMS=2 / 3) one side stream changes nothing (3 M rows:
        // 14.77 against 14.79 ms of kernels, 375 000 rows in 9 chunks 3.46 against 3.61) and a second one doubles the
        // kernel time (27.7 ms: its hardware queue is shared with the copy streams and its launches start ~250 us
        // late, `rocprofv3 --kernel-trace`); without the barrier bit on one stream (hipExtAnyOrderLaunch) nothing
        // changes on gfx950.  What did help: fewer launches (the two-observation classes in one, above).
        int n_streams = 1;
        if (const char *env = getenv("ADH_FUSED_STREAMS")) n_streams = std::min(std::max(atoi(env), 1), 3);
        int n_launch = 0;
        for (int w = 0; w < 4; ++w) n_launch += fblocks[w] > 0;
        if (n_launch < 2) n_streams = 1;
        for (int a = 0; a + 1 < n_streams; ++a) {
            if (!h->stream_aux[a]) HIP_TRY(hipStreamCreateWithFlags(&h->stream_aux[a], hipStreamNonBlocking));
            if (!h->ev_aux[a]) HIP_TRY(hipEventCreateWithFlags(&h->ev_aux[a], hipEventDisableTiming));
        }
        bool used[2] = {false, false};
        int launched = 0;
        hipStream_t lst = st;
        if (n_streams > 1) HIP_TRY(hipEventRecord(h->ev_fork, st));
#define ADH_LAUNCH_FUSED_TW(W, FM_MIN, FM_MAX, NO, TW)                                                                    \
    hipLaunchKernelGGL((adh_fused_kernel<FM_MIN, FM_MAX, NO, TW>), dim3((unsigned)fblocks[W]), dim3(ADH_WAVE), 0, lst,   \
                       (FusedArgs{h->run, h->d_lib, p.d_recs, fcs[W], h->cs.iso, (int32_t)n_iso, *cfg, h->d_wtp, *out, (int32_t)stop_phase}))
#define ADH_LAUNCH_FUSED(W, FM_MIN, FM_MAX, NO)                                                 \
    if (fblocks[W] > 0) {                                                                       \
        lst = st;                                                                               \
        if (n_streams > 1 && launched > 0) {                                                    \
            const int a = std::min(launched - 1, n_streams - 2);                                \
            lst = h->stream_aux[a];                                                             \
            if (!used[a]) HIP_TRY(hipStreamWaitEvent(lst, h->ev_fork, 0));                      \
            used[a] = true;                                                                     \
        }                                                                                       \
        if (wide) ADH_LAUNCH_FUSED_TW(W, FM_MIN, FM_MAX, NO, 16);                               \
        else ADH_LAUNCH_FUSED_TW(W, FM_MIN, FM_MAX, NO, ADH_FUSED_TW3);                         \
        HIP_TRY(hipGetLastError());                                                             \
        ++launched;                                                                             \
    }
        ADH_LAUNCH_FUSED(0, 8, ADH_FUSED_FM3, 1)
        ADH_LAUNCH_FUSED(1, ADH_FUSED_FM3 + 4, 32, 1)
        ADH_LAUNCH_FUSED(2, 8, 32, 2)
        for (int a = 0; a < 2; ++a) {
            if (!used[a]) continue;
            HIP_TRY(hipEventRecord(h->ev_aux[a], h->stream_aux[a]));
            HIP_TRY(hipStreamWaitEvent(st, h->ev_aux[a], 0));
        }
#undef ADH_LAUNCH_FUSED
#undef ADH_LAUNCH_FUSED_TW
    }
    if (stop_phase != 2 && !fused_only) {
        int64_t first = n_fused;
        for (int c = ADH_CLASS_FAST2; c <= ADH_CLASS_GENERIC; ++c) {
            const bool wide_class = c >= ADH_CLASS_WIDE2 && c < ADH_CLASS_GENERIC;
            if (wide_class && run_fast && c == ADH_CLASS_WIDE2) {
                // the wide classes in two launches (adh_features_fast.hip): classes WIDE2 .. WIDE2 + 2 / WIDE1 .. WIDE1 + 2
                // (one candidate per wavefront) and their 32-lane twins MID2 .. / MID1 .., which follow each other in
                // the plan's order: the heavy bodies first, then the ones that fit three wavefronts per SIMD
                int64_t class_first[ADH_N_CLASSES];
                {
                    int64_t at = first;
                    for (int q = c; q < ADH_CLASS_GENERIC; ++q) class_first[q] = at, at += p.n_class[q];
                }
                for (int part = 0; part < 2; ++part) {
                    const int kinds = part == 0 ? ADH_WIDE_HEAVY : ADH_WIDE_LIGHT;
                    WideClasses wcs{};
                    int64_t blocks = 0;
                    for (int kind = 0; kind < 12; ++kind) {
                        const int j = kind % 6, two = kind / 6;
                        const int q = (j < 3 ? (two ? ADH_CLASS_WIDE2 : ADH_CLASS_WIDE1) : (two ? ADH_CLASS_MID2 : ADH_CLASS_MID1)) + j % 3;
                        const int64_t nq = p.n_class[q];
                        if (nq > 0 && ((kinds >> kind) & 1)) {
                            const int per = j < 3 ? 1 : 2;
                            wcs.first_block[wcs.n] = (int32_t)blocks;
                            wcs.first_cand[wcs.n] = (int32_t)(class_first[q] - n_fused);
                            wcs.n_cand[wcs.n] = (int32_t)nq;
                            wcs.kind[wcs.n] = kind;
                            ++wcs.n;
                            blocks += (nq + per - 1) / per;
                        }
                    }
                    wcs.first_block[wcs.n] = (int32_t)blocks;
                    if (blocks == 0) continue;
                    const CandRec *base = p.d_recs + n_fused;
                    const WideArgs wa{h->run, base, wcs, h->cs.iso, n_iso, *cfg, d_scratch, h->d_wtp, *out, (int32_t)stop_phase};
                    const dim3 grid((unsigned)blocks), wave(ADH_WAVE);
                    if (part == 0) hipLaunchKernelGGL((adh_feature_wide_kernel<ADH_WIDE_HEAVY>), grid, wave, 0, st, wa);
                    else hipLaunchKernelGGL((adh_feature_wide_kernel<ADH_WIDE_LIGHT>), grid, wave, 0, st, wa);
                    HIP_TRY(hipGetLastError());
                }
            }
            if (!wide_class && p.n_class[c] > 0 && (c == ADH_CLASS_GENERIC ? run_generic : run_fast)) {
                const CandRec *recs = p.d_recs + first;
                if (c == ADH_CLASS_GENERIC) {
                    hipLaunchKernelGGL(adh_feature_kernel, dim3((unsigned)p.n_class[c]), dim3(ADH_WAVE), f_lds, st, h->run, recs,
                                       h->cs.iso, n_iso, *cfg, d_scratch, *out, p.caps_generic);
                } else {
                    const unsigned blocks = (unsigned)((p.n_class[c] + per_block - 1) / per_block);
                    const int32_t nc = (int32_t)p.n_class[c];
#define ADH_LAUNCH_FAST(FM, NO)                                                                              \
    hipLaunchKernelGGL((adh_feature_fast_kernel<FM, NO>), dim3(blocks), dim3(ADH_WAVE), 0, st, h->run, recs, \
                       nc, h->cs.iso, n_iso, *cfg, d_scratch, h->d_wtp, *out, (int32_t)stop_phase)
                    switch (c) {
                        case ADH_CLASS_FAST2 + 0: ADH_LAUNCH_FAST(16, 2); break;
                        case ADH_CLASS_FAST2 + 1: ADH_LAUNCH_FAST(24, 2); break;
                        case ADH_CLASS_FAST2 + 2: ADH_LAUNCH_FAST(32, 2); break;
                        case ADH_CLASS_FAST1 + 0: ADH_LAUNCH_FAST(8, 1); break;
                        case ADH_CLASS_FAST1 + 1: ADH_LAUNCH_FAST(12, 1); break;
                        case ADH_CLASS_FAST1 + 2: ADH_LAUNCH_FAST(16, 1); break;
                        case ADH_CLASS_FAST1 + 3: ADH_LAUNCH_FAST(20, 1); break;
                        case ADH_CLASS_FAST1 + 4: ADH_LAUNCH_FAST(24, 1); break;
                        case ADH_CLASS_FAST1 + 5: ADH_LAUNCH_FAST(28, 1); break;
                        default: ADH_LAUNCH_FAST(32, 1); break;
                    }
#undef ADH_LAUNCH_FAST
                }
                HIP_TRY(hipGetLastError());
            }
            first += p.n_class[c];
        }
    }
    HIP_TRY(hipEventRecord(t.e2, st));
    h->timed.push_back(t);
    return ADH_OK;
}

int check_score_args(adh_handle *h, const adh_scoring_config_t *cfg, const adh_output_t *out) {
    if (cfg->top_k_fragments == 0 || cfg->top_k_isotopes == 0)
        return fail(ADH_ERR_INVALID_ARGUMENT, "top_k_fragments / top_k_isotopes must be > 0");
    if (out->top_k <= 0) return fail(ADH_ERR_INVALID_ARGUMENT, "output top_k must be > 0");
    if (h->tims_staged && h->tims.cycle_len > 1024)
        return fail(ADH_ERR_UNSUPPORTED, "ion-mobility cycles of more than 1024 frames are not supported");
    return ADH_OK;
}

int64_t pick_chunk(int64_t n, bool fine) {
    // rows per pipeline chunk.  Two regimes (round 5):
    //  * tables of >= 2 M rows (the one-GPU headline): chunks of 524288 rows - fewer, larger launches suit the fused
    //    kernel, and the un-overlapped first H2D / last D2H are a small share of the call.  3 M candidates, host -> host /
    //    kernels (round 3, two sweeps on one box): 196608 rows 37.9, 36.6 / 16.9 ms, 262144 35.0, 36.0 / 16.1, 393216
    //    36.7, 35.9 / 15.2, 524288 35.5, 36.0 / 14.7; 1048576 cost 2 ms of host -> host time (first and last copies)
    //  * smaller tables (the 375 000-row shards of an eight-GPU run, the batches of the optimisation loop): the copy-out
    //    is the longest stage (8.3 ns per row at 54 GB/s against 0.18 ms + 4.7 ns per row of kernels per chunk), so the call
    //    is as long as the wait for the FIRST copy-out plus every gap of the copy-out stream: about five chunks (sweeps of
    //    round 5, 375 000 rows: 3 chunks 5.1-5.5 ms, 4 parts 4.68, 5 4.57, 6 4.62, 7 4.76, 8 4.88, 10 5.36; 750 000 rows: 9.4 / 8.11 /
    //    8.11 / 8.4 / 8.9), none below ADH_CHUNK_MIN rows (below ~50 000 rows a chunk's kernels take longer than its copy-out and the stream
    //    waits for them).  Round 4 cut such a table in two or three (one short first chunk): the 375 000-row shard took
    //    5.1 ms = 0.65 ramp + 3.25 copies + 0.63 gap + 0.55 tail; see tools/bench_shard.py
    const int64_t big = 524288;
    if (const char *env = getenv("ADH_CHUNK")) {
        const int64_t target = std::max<int64_t>(atoll(env), 1024);
        const int64_t parts = (n + target - 1) / target;
        if (parts <= 1) return std::max<int64_t>(n, 1);
        return (n + parts - 1) / parts;
    }
    if (n >= 4 * big) {
        const int64_t parts = (n + big - 1) / big;
        return (n + parts - 1) / parts;
    }
    if (!fine) {
        // ion-mobility tables: a chunk is ~7 launches and the gather's pair lists, so fewer chunks win (configs[3],
        // 600 000 candidates: 18.4 ms host -> host in 3 chunks, 21.3 in 7): the rule of round 4 - chunks of 524288
        // rows, a table of 196608 rows and more in two at least
        int64_t parts = (n + big - 1) / big;
        if (parts < 2 && n >= 196608) parts = 2;
        if (parts <= 1) return std::max<int64_t>(n, 1);
        return (n + parts - 1) / parts;
    }
    int64_t want_parts = 5, min_rows = 40960;
    if (const char *env = getenv("ADH_CHUNK_PARTS")) want_parts = std::max<int64_t>(atoll(env), 1);
    if (const char *env = getenv("ADH_CHUNK_MIN")) min_rows = std::max<int64_t>(atoll(env), 1024);
    int64_t c = std::max(min_rows, (n + want_parts - 1) / want_parts);
    c = std::min(c, big);
    const int64_t parts = (n + c - 1) / c;
    if (parts <= 1) return std::max<int64_t>(n, 1);
    return (n + parts - 1) / parts;
}

}  // namespace

}  // extern "C"

// The columns of the device tables that repeat the candidate table (precursor_idx, rank) or the library (per
// fragment slot) from fragment_lib_slot: the device-side twin of rebuild_host_rows, one thread per slot.
__global__ void adh_rebuild_columns_kernel(DevCands c, const LibRec *__restrict__ lib, DevOut out, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int top_k = out.top_k;
    if (t >= n * top_k) return;
    const int64_t i = t / top_k;
    const int j = (int)(t - i * top_k);
    const bool skip = c.flags && (c.flags[i] & ADH_FLAG_SKIP);
    const uint32_t p = skip ? 0u : c.precursor_idx[i];
    const uint8_t r = skip ? (uint8_t)0 : c.rank[i];
    if (j == 0) {
        out.precursor_idx[i] = p;
        out.rank[i] = r;
    }
    const uint16_t s = out.fragment_lib_slot[t];
    if (!s) return;  // (the tables were zeroed before the kernels ran)
    const LibRec l = lib[c.frag_start[i] + s - 1];
    out.fragment_precursor_idx[t] = p;
    out.fragment_rank[t] = r;
    out.fragment_mz_library[t] = l.mz_library;
    out.fragment_mz[t] = l.mz;
    out.fragment_position[t] = l.position;
    out.fragment_number[t] = l.number;
    out.fragment_type[t] = l.type;
    out.fragment_charge[t] = l.charge;
    out.fragment_loss_type[t] = l.loss_type;
}

// ---- compacted copy-out of the fragment tables (round 4).  A candidate fills the first K of its top_k fragment slots
// (K = fragments with signal: 4.6 of 12 on the headline) and 58 % of the 264 bytes per candidate that the five
// computed fragment tables + fragment_lib_slot put on PCIe were zeros.  Per chunk of the pipeline: K per row, an
// exclusive scan, the filled slots of the six columns packed behind each other; the offsets travel first (the host
// needs the total to size the six copies), the host team expands them into the caller's padded tables.
__global__ void adh_slot_count_kernel(const uint16_t *__restrict__ lib_slot, int64_t row0, int64_t n, int top_k,
                                      uint32_t *__restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t k = 0;
    if (i < n) {
        const uint16_t *s = lib_slot + (row0 + i) * (int64_t)top_k;
        while (k < (uint32_t)top_k && s[k]) ++k;  // (filled slots are the leading ones: candidate.py:403-442)
    }
    cnt[i] = k;  // (entry n: 0, so that the scan's last entry is the total)
}

struct CompactCols {
    float *f[5];
    uint16_t *slot;
};

__global__ void adh_compact_kernel(DevOut t, int64_t row0, int64_t n, int top_k, const uint32_t *__restrict__ off,
                                   CompactCols c) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n * top_k) return;
    const int64_t i = id / top_k;
    const int j = (int)(id - i * top_k);
    const uint32_t a = off[i], k = off[i + 1] - a;
    if ((uint32_t)j >= k) return;
    const int64_t src = (row0 + i) * (int64_t)top_k + j;
    const size_t dst = (size_t)a + (size_t)j;
    c.f[0][dst] = t.fragment_mz_observed[src];
    c.f[1][dst] = t.fragment_height[src];
    c.f[2][dst] = t.fragment_intensity[src];
    c.f[3][dst] = t.fragment_mass_error[src];
    c.f[4][dst] = t.fragment_correlation[src];
    c.slot[dst] = t.fragment_lib_slot[src];
}

// ---- compacted, column-major copy-out of the operator path (round 5, adh_score_candidates_compact).  What the
// DataFrames of collect_candidates / collect_fragments keep of the padded tables is 91 % of the rows and 38 % of the
// fragment slots (headline).  Per chunk, behind its scoring kernels: (valid, filled slots) per row as one 64-bit count,
// an exclusive scan, and a pack kernel that writes every column of the chunk's valid rows and filled slots - features
// transposed to [feature][row], the library columns of a slot read from the staged library, ids from the candidate
// table - DENSELY into the chunk's block of a device staging buffer (CopBlock: where a column starts follows from the
// chunk's two totals).  The totals reach the host first (8 bytes, stored by the pack kernel straight into page-locked
// memory); the host then moves the block with ONE DMA copy of exactly its used bytes into a page-locked twin, and host
// threads unpack finished blocks into the caller's arrays while later chunks are scored.
// (Measured and dropped: the pack kernel storing through PCIe straight into host memory.  Kernel stores reach the link
// rate - tools/probes/kcopy_probe.hip: 54-55 GB/s from 64 workgroups - but a copy-out kernel does not run BESIDE the
// scoring kernels: on a stream of its own its workgroups wait until the scoring stream's backlog has drained (first
// chunk on the host 15.7 ms into a 36 ms call), on a high-priority stream the launches of the scoring stream stall
// instead (52 ms), and a compute unit backed up with PCIe stores stalls every wavefront on it: scoring kernels of a step
// 14.5 ms alone, 16.8 beside 32 copying workgroups, 23.5 beside 64, 27.2 beside 128.  An 8-byte hipMemcpyAsync is such
// a kernel, too: with the totals fetched that way the first block arrived 57 ms into the call.)
__global__ void adh_cop_count_kernel(const uint8_t *__restrict__ valid, const uint16_t *__restrict__ lib_slot, int64_t row0,
                                     int64_t n, int top_k, uint64_t *__restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint64_t v = 0;
    if (i < n && valid[row0 + i]) {
        const uint16_t *s = lib_slot + (row0 + i) * (int64_t)top_k;
        uint32_t k = 0;
        while (k < (uint32_t)top_k && s[k]) ++k;  // (filled slots are the leading ones: candidate.py:403-442)
        v = (1ull << 32) | k;
    }
    cnt[i] = v;  // (entry n: 0, so that the scan's last entry is the total)
}

// a chunk's block, dense: [row u32 | filled slots u8 | features f32 [46][R]] for its R valid rows, then
// [fragment_lib_slot u16 | 5 computed float columns] for its S filled slots; every column starts on a multiple of 16
// bytes.  Round 6: what repeats the candidate table (precursor_idx, rank, and fragment_row = the row of a slot's
// candidate) and the seven library columns of a slot stay off the wire - 26 instead of 42 bytes per slot, 188 instead of
// 193 per row - and are rebuilt by the unpack team from the caller's candidate columns and the host copy of the
// library, as the padded path does (rebuild_host_rows).
struct CopBlock {
    size_t row, cnt, feat, s_slot, s_f[5], total;
    __host__ __device__ CopBlock(uint64_t R, uint64_t S) {
        size_t o = 0;
        row = o, o += (R * 4 + 15) & ~(size_t)15;
        cnt = o, o += (R + 15) & ~(size_t)15;
        feat = o, o += (R * 4 * ADH_NUM_FEATURES + 15) & ~(size_t)15;
        s_slot = o, o += (S * 2 + 15) & ~(size_t)15;
        for (int j = 0; j < 5; ++j) s_f[j] = o, o += (S * 4 + 15) & ~(size_t)15;
        total = o;
    }
};
// where the block of the chunk that starts at row a sits in the staging buffers (device and host alike): room for
// every row valid and every slot filled
struct CopLayout {
    size_t per_row, total;
    CopLayout(int64_t n, int top_k, int64_t n_chunks)
        : per_row(5 + 4 * ADH_NUM_FEATURES + (size_t)top_k * 22), total((size_t)n * per_row + (size_t)(n_chunks + 1) * 1024) {}
    size_t base(int64_t a, int64_t ci) const { return ((size_t)a * per_row + (size_t)ci * 1024 + 255) & ~(size_t)255; }
};

__global__ void adh_cop_pack_kernel(DevOut t, DevCands c, const LibRec *__restrict__ lib, int64_t row0, int64_t n, int top_k,
                                    const uint64_t *__restrict__ off, unsigned char *__restrict__ block,
                                    uint64_t *__restrict__ totals) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t tot = off[n];
    const uint64_t R = tot >> 32, S = tot & 0xFFFFFFFFull;
    if (tid == 0) totals[0] = tot;  // (page-locked host memory)
    const CopBlock L(R, S);
    uint32_t *const o_row = reinterpret_cast<uint32_t *>(block + L.row);
    uint8_t *const o_cnt = block + L.cnt;
    float *const o_feat = reinterpret_cast<float *>(block + L.feat);
    // valid rows: row id, number of filled slots, the feature row transposed (consecutive lanes = consecutive output
    // rows of one column)
    for (int64_t i = tid; i < n; i += stride) {
        const uint64_t o = off[i], o1 = off[i + 1];
        if ((o1 >> 32) == (o >> 32)) continue;
        const int64_t j = (int64_t)(o >> 32), r = row0 + i;
        o_row[j] = (uint32_t)r;
        o_cnt[j] = (uint8_t)((uint32_t)o1 - (uint32_t)o);
        const float *f = t.features + r * ADH_NUM_FEATURES;
#pragma unroll
        for (int k = 0; k < ADH_NUM_FEATURES; ++k) o_feat[(size_t)k * R + (size_t)j] = f[k];
    }
    // filled slots
    uint16_t *const s_slot = reinterpret_cast<uint16_t *>(block + L.s_slot);
    const int64_t n_slots = n * (int64_t)top_k;
    for (int64_t id = tid; id < n_slots; id += stride) {
        const int64_t i = id / top_k;
        const int s = (int)(id - i * top_k);
        const uint64_t o = off[i], o1 = off[i + 1];
        const uint32_t a = (uint32_t)o, k = (uint32_t)o1 - a;
        if ((uint32_t)s >= k) continue;
        const int64_t r = row0 + i, src = r * (int64_t)top_k + s;
        const size_t dst = (size_t)a + (size_t)s;
        s_slot[dst] = t.fragment_lib_slot[src];
        reinterpret_cast<float *>(block + L.s_f[0])[dst] = t.fragment_mz_observed[src];
        reinterpret_cast<float *>(block + L.s_f[1])[dst] = t.fragment_height[src];
        reinterpret_cast<float *>(block + L.s_f[2])[dst] = t.fragment_intensity[src];
        reinterpret_cast<float *>(block + L.s_f[3])[dst] = t.fragment_mass_error[src];
        reinterpret_cast<float *>(block + L.s_f[4])[dst] = t.fragment_correlation[src];
    }
}

namespace {
// stripe w of T of a finished block (page-locked host copy) -> the caller's arrays (rows at base_r, slots at base_s): the
// stripe is a range of the block's valid rows together with their slots; what the block leaves out - ids, the row of a
// slot's candidate, the library columns - comes from the candidate columns `c` and the library `lib`
void cop_copy_stripe(const unsigned char *block, int64_t cnt_r, int64_t cnt_s, int64_t base_r, int64_t base_s,
                     adh_compact_output_t *out, int w, int T, const adh_candidates_t *c, const LibRec *lib) {
    const CopBlock L((uint64_t)cnt_r, (uint64_t)cnt_s);
    const int64_t lo = cnt_r * w / T, hi = cnt_r * (w + 1) / T;
    if (hi <= lo) return;
    const uint32_t *rows = reinterpret_cast<const uint32_t *>(block + L.row);
    const uint8_t *cnt = block + L.cnt;
    int64_t s_lo = 0;  // slots of the rows before the stripe
    for (int64_t j = 0; j < lo; ++j) s_lo += cnt[j];
    memcpy(out->row + base_r + lo, rows + lo, (size_t)(hi - lo) * 4);
    const float *fb = reinterpret_cast<const float *>(block + L.feat);
    for (int k = 0; k < ADH_NUM_FEATURES; ++k)
        memcpy(out->features + (size_t)k * (size_t)out->rows_capacity + (size_t)(base_r + lo),
               fb + (size_t)k * (size_t)cnt_r + (size_t)lo, (size_t)(hi - lo) * 4);
    const uint16_t *slot = reinterpret_cast<const uint16_t *>(block + L.s_slot);
    int64_t d = base_s + s_lo, at = s_lo;
    for (int64_t j = lo; j < hi; ++j) {
        const uint32_t r = rows[j];
        const uint32_t p = c->precursor_idx[r];
        const uint8_t rk = c->rank ? c->rank[r] : (uint8_t)0;
        out->precursor_idx[base_r + j] = p;
        out->rank[base_r + j] = rk;
        const LibRec *base = lib + c->frag_start_idx[r];
        const int k = (int)cnt[j];
        for (int q = 0; q < k; ++q, ++d, ++at) {
            const LibRec &l = base[slot[at] - 1];
            out->fragment_row[d] = r;
            out->fragment_precursor_idx[d] = p;
            out->fragment_rank[d] = rk;
            out->fragment_mz_library[d] = l.mz_library;
            out->fragment_mz[d] = l.mz;
            out->fragment_position[d] = l.position;
            out->fragment_number[d] = l.number;
            out->fragment_type[d] = l.type;
            out->fragment_charge[d] = l.charge;
            out->fragment_loss_type[d] = l.loss_type;
        }
    }
    const size_t m = (size_t)(at - s_lo);
    if (m) {
        float *const fcol[5] = {out->fragment_mz_observed, out->fragment_height, out->fragment_intensity, out->fragment_mass_error,
                                out->fragment_correlation};
        for (int j = 0; j < 5; ++j) memcpy(fcol[j] + base_s + s_lo, block + L.s_f[j] + (size_t)s_lo * 4, m * 4);
    }
}
}  // namespace

extern "C" {

namespace {

// fill in what a host -> host call left out of the device tables (DevTables::partial) before somebody reads them
int materialise_tables(adh_handle *h) {
    if (h->last_tables < 0) return ADH_OK;
    DevTables &t = h->tables[h->last_tables];
    if (!t.partial) return ADH_OK;
    const int64_t n = h->last_rows;
    if (n > h->cs.n || !h->lib_staged)  // cannot happen through the ABI (replacing either settles the tables first)
        return fail(ADH_ERR_NOT_STAGED, "the candidate table / library the device tables were scored from is gone");
    if (n > 0) {
        HIP_TRY(hipSetDevice(h->device));
        adh_output_t view = t.view;
        view.n = n;
        const int64_t threads = n * (int64_t)view.top_k;
        hipLaunchKernelGGL(adh_rebuild_columns_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, h->stream, h->cs.d,
                           h->d_lib, view, n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    t.partial = false;
    return ADH_OK;
}

}  // namespace

int adh_upload_candidates(adh_handle_t *h, const adh_candidates_t *c) {
    if (!h || !c) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = check_candidate_args(h, c);
    if (rc != ADH_OK) return rc;
    HIP_TRY(hipSetDevice(h->device));
    rc = materialise_tables(h);  // the last call's tables are rebuilt from the candidate columns that go away now
    if (rc != ADH_OK) return rc;
    h->plan = Plan();
    h->cands_uploaded = false;
    rc = cand_reserve(h, c->n, c->n_isotope_cols);
    if (rc != ADH_OK) return rc;
    rc = cand_upload_range(h, c, 0, c->n, h->stream);
    if (rc != ADH_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->cands_uploaded = true;
    return ADH_OK;
}

int adh_score_uploaded(adh_handle_t *h, const adh_scoring_config_t *cfg, adh_output_t *out, void *hip_stream) {
    if (!h || !cfg || !out) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!h->cands_uploaded) return fail(ADH_ERR_NOT_STAGED, "no candidate table uploaded");
    if (out->n != h->cs.n) return fail(ADH_ERR_INVALID_ARGUMENT, "output rows != candidates");
    int rc = check_score_args(h, cfg, out);
    if (rc != ADH_OK) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if (h->cs.n == 0) return ADH_OK;
    hipStream_t st = (hipStream_t)hip_stream;  // NULL is HIP's default stream, taken literally
    // the plan of the uploaded table is kept until the table or the plan-relevant settings change
    Plan &p = h->plan;
    const PlanKey key = plan_key(cfg);
    if (!(p.ready && p.top_k_fragments == key.top_k_fragments && p.top_k_isotopes == key.top_k_isotopes &&
          p.fast_ok == key.fast_cfg && p.fused_ok == key.fused_cfg && p.wide_ok == key.wide_cfg && p.quant_all == key.quant_all)) {
        p = Plan();
        rc = plan_enqueue(h, h->slots[0], cfg, 0, h->cs.n, st);
        if (rc == ADH_OK) rc = plan_finish(h, h->slots[0], cfg, 0, h->cs.n, p);
        if (rc != ADH_OK) return rc;
    }
    return launch_scoring(h, p, cfg, out, st);
}

int adh_get_stream(adh_handle_t *h, void **hip_stream) {
    if (!h || !hip_stream) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    *hip_stream = (void *)h->stream;
    return ADH_OK;
}

int adh_synchronize(adh_handle_t *h) {
    if (!h) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ADH_OK;
}

int adh_kernel_time_ms(adh_handle_t *h, double *gather_ms, double *feature_ms, int64_t *launches, int reset) {
    if (!h || !gather_ms || !feature_ms || !launches) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    fold_timed(h, true);
    const int64_t n = h->n_timed;
    *gather_ms = n ? h->sum_gather_ms / (double)n : 0.0;
    *feature_ms = n ? h->sum_feature_ms / (double)n : 0.0;
    *launches = n;
    if (reset) {
        h->sum_gather_ms = h->sum_feature_ms = 0.0;
        h->n_timed = 0;
    }
    return ADH_OK;
}

namespace {

// rows [a, b) of the rebuildable host columns (OutputPsmDF columns that repeat the candidate table / the library,
// alphadia/search/scoring/output.py:17-97; written by the kernels as candidate.py:175-176, 403-481)
void rebuild_host_rows(const adh_handle *h, const adh_candidates_t *c, adh_output_t *out, const uint16_t *slots,
                       int64_t a, int64_t b) {
    const int top_k = out->top_k;
    const LibRec *lib = h->h_lib.data();
    for (int64_t i = a; i < b; ++i) {
        const bool skip = c->flags && (c->flags[i] & ADH_FLAG_SKIP);
        const uint32_t p = skip ? 0u : c->precursor_idx[i];
        const uint8_t r = skip ? (uint8_t)0 : c->rank[i];
        out->precursor_idx[i] = p;
        out->rank[i] = r;
        const LibRec *base = lib + c->frag_start_idx[i];
        for (int j = 0; j < top_k; ++j) {
            const size_t o = (size_t)i * (size_t)top_k + (size_t)j;
            const uint16_t s = slots[o];
            if (s) {
                const LibRec &l = base[s - 1];
                out->fragment_precursor_idx[o] = p;
                out->fragment_rank[o] = r;
                out->fragment_mz_library[o] = l.mz_library;
                out->fragment_mz[o] = l.mz;
                out->fragment_position[o] = l.position;
                out->fragment_number[o] = l.number;
                out->fragment_type[o] = l.type;
                out->fragment_charge[o] = l.charge;
                out->fragment_loss_type[o] = l.loss_type;
            } else {
                out->fragment_precursor_idx[o] = 0;
                out->fragment_rank[o] = 0;
                out->fragment_mz_library[o] = 0.0f;
                out->fragment_mz[o] = 0.0f;
                out->fragment_position[o] = 0;
                out->fragment_number[o] = 0;
                out->fragment_type[o] = 0;
                out->fragment_charge[o] = 0;
                out->fragment_loss_type[o] = 0;
            }
        }
    }
}

// Compact buffers of a call of n rows cut at `cut`: [offsets: one uint32 per row + one per chunk][5 float columns and
// the slot column, n * top_k entries each, a chunk's packed slots starting at cut[chunk] * top_k]
struct CompactLayout {
    size_t off_bytes, col_elems, total;
    size_t off_of(int64_t a, int64_t ci) const { return (size_t)(a + ci) * 4; }  // byte offset of chunk ci's offsets
    size_t col_f(int j) const { return off_bytes + (size_t)j * col_elems * 4; }
    size_t col_slot() const { return off_bytes + 5 * col_elems * 4; }
    CompactLayout(int64_t n, int64_t n_chunks, int top_k) {
        off_bytes = ((size_t)(n + n_chunks) * 4 + 255) / 256 * 256;
        col_elems = ((size_t)n * (size_t)top_k + 63) / 64 * 64;
        total = off_bytes + col_elems * 22;
    }
};

// rows [a, b) of a chunk that starts at row a0: the packed slots back into the caller's padded tables (zeros behind)
void expand_host_rows(adh_output_t *out, uint16_t *slot_host, const unsigned char *cmp, const CompactLayout &lay, int64_t a0,
                      int64_t ci, int64_t a, int64_t b) {
    const int top_k = out->top_k;
    const uint32_t *off = reinterpret_cast<const uint32_t *>(cmp + lay.off_of(a0, ci));
    float *dst[5] = {out->fragment_mz_observed, out->fragment_height, out->fragment_intensity, out->fragment_mass_error,
                     out->fragment_correlation};
    const size_t base = (size_t)a0 * (size_t)top_k;
    const float *src[5];
    for (int j = 0; j < 5; ++j) src[j] = reinterpret_cast<const float *>(cmp + lay.col_f(j)) + base;
    const uint16_t *src_s = reinterpret_cast<const uint16_t *>(cmp + lay.col_slot()) + base;
    if (top_k == 12) {
        // the usual width (default.yaml:185): a row is three 16-byte vectors; the packed source is read unmasked (the
        // buffer has slack behind its last entry) and cut to the row's k entries with a mask; streaming stores where the
        // destination allows - the rows are not read again on this side
        alignas(16) static const uint32_t kMask[13][12] = {
#define ADH_M(k) {k > 0 ? ~0u : 0u, k > 1 ? ~0u : 0u, k > 2 ? ~0u : 0u, k > 3 ? ~0u : 0u, k > 4 ? ~0u : 0u, k > 5 ? ~0u : 0u, \
                  k > 6 ? ~0u : 0u, k > 7 ? ~0u : 0u, k > 8 ? ~0u : 0u, k > 9 ? ~0u : 0u, k > 10 ? ~0u : 0u, k > 11 ? ~0u : 0u}
            ADH_M(0), ADH_M(1), ADH_M(2), ADH_M(3), ADH_M(4), ADH_M(5), ADH_M(6), ADH_M(7), ADH_M(8), ADH_M(9), ADH_M(10), ADH_M(11), ADH_M(12)
#undef ADH_M
        };
        bool aligned = true;
        for (int j = 0; j < 5; ++j) aligned = aligned && (reinterpret_cast<uintptr_t>(dst[j]) & 15u) == 0;
        for (int64_t i = a; i < b; ++i) {
            const uint32_t o = off[i - a0];
            const uint32_t k = std::min<uint32_t>(off[i - a0 + 1] - o, 12u);
            const size_t r0 = (size_t)i * 12;
            const __m128 m0 = _mm_load_ps(reinterpret_cast<const float *>(kMask[k]));
            const __m128 m1 = _mm_load_ps(reinterpret_cast<const float *>(kMask[k] + 4));
            const __m128 m2 = _mm_load_ps(reinterpret_cast<const float *>(kMask[k] + 8));
            for (int j = 0; j < 5; ++j) {
                const float *sp = src[j] + o;
                float *row = dst[j] + r0;
                const __m128 v0 = _mm_and_ps(_mm_loadu_ps(sp), m0), v1 = _mm_and_ps(_mm_loadu_ps(sp + 4), m1),
                             v2 = _mm_and_ps(_mm_loadu_ps(sp + 8), m2);
                if (aligned) {
                    _mm_stream_ps(row, v0);
                    _mm_stream_ps(row + 4, v1);
                    _mm_stream_ps(row + 8, v2);
                } else {
                    _mm_storeu_ps(row, v0);
                    _mm_storeu_ps(row + 4, v1);
                    _mm_storeu_ps(row + 8, v2);
                }
            }
            const uint16_t *sp = src_s + o;
            uint16_t *row = slot_host + r0;
            for (uint32_t t = 0; t < 12; ++t) row[t] = t < k ? sp[t] : (uint16_t)0;
        }
        _mm_sfence();
        return;
    }
    // any other width: plain loops (a memcpy / memset pair per row and column would be 36 M library calls per 3 M rows)
    for (int64_t i = a; i < b; ++i) {
        const uint32_t o = off[i - a0];
        const int k = (int)(off[i - a0 + 1] - o);
        const size_t r0 = (size_t)i * (size_t)top_k;
        for (int j = 0; j < 5; ++j) {
            const float *sp = src[j] + o;
            float *row = dst[j] + r0;
            int t = 0;
            for (; t < k; ++t) row[t] = sp[t];
            for (; t < top_k; ++t) row[t] = 0.0f;
        }
        const uint16_t *sp = src_s + o;
        uint16_t *row = slot_host + r0;
        int t = 0;
        for (; t < k; ++t) row[t] = sp[t];
        for (; t < top_k; ++t) row[t] = 0;
    }
}

// this rank's share of the host: at most 16 threads, and of the cores this process may use (host_cpu_budget: quota,
// affinity, hardware) only the LOCAL_WORLD_SIZE-th part - the ranks of a node run side by side under ONE quota
int host_thread_share() {
    int t = 16;
    if (const char *env = getenv("ADH_HOST_THREADS")) t = atoi(env);
    else {
        int ranks = 1;
        if (const char *lw = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(atoi(lw), 1);
        t = std::min<int>(t, std::max<int>(host_cpu_budget() / ranks, 1));
    }
    return std::max(t, 1);
}

int host_threads_for(int64_t n) {
    // the team that rebuilds the id / library columns behind the copy-out (or unpacks the compact blocks)
    const int t = (int)std::min<int64_t>(host_thread_share(), n / 16384);  // (a thread per 16 k rows at least: starting one costs ~20 us)
    return std::max(t, 1);
}

// Does the host rebuild the id / library columns of the padded tables (197 of 646 bytes per candidate stay off PCIe),
// or does the device write them and the link carry everything?  A thread rebuilds ~23 000 rows per ms, the link delivers
// 122 000 rows per ms of wire tables: below ~6 threads the team is what the call waits for (measured with 2 threads -
// the eighth part of the pool's 16-core quota: a 375 000-row shard takes 8.9 ms against 4.6), and the 44 % more bytes
// cost less (every GPU of a node has its own link, the ranks share the CPU quota).  ADH_REBUILD_MIN_THREADS moves the
// threshold (0: always rebuild).
bool host_rebuild_pays() {
    int least = 6;
    if (const char *env = getenv("ADH_REBUILD_MIN_THREADS")) least = atoi(env);
    return host_thread_share() >= least;
}

}  // namespace

namespace {
// the host -> host pipeline behind adh_score_candidates (padded tables into `out`) and adh_score_candidates_compact
// (`cop` set: `out` only carries n and top_k, the compacted columns go to `cop`)
int score_pipeline(adh_handle_t *h, const adh_candidates_t *c, const adh_scoring_config_t *cfg, adh_output_t *out,
                   adh_compact_output_t *cop) {
    if (!h || !c || !cfg || !out) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (out->n != c->n) return fail(ADH_ERR_INVALID_ARGUMENT, "output rows != candidates");
    int rc = check_candidate_args(h, c);
    if (rc == ADH_OK) rc = check_score_args(h, cfg, out);
    if (rc != ADH_OK) return rc;
    for (int i = 0; i < kNumOutFields && !cop; ++i)
        if (!kOutFields[i].optional && *out_member(out, kOutFields[i]) == nullptr)
            return fail(ADH_ERR_INVALID_ARGUMENT, "output buffer is NULL");
    HIP_TRY(hipSetDevice(h->device));
    const bool timing = getenv("ADH_DEBUG_TIMING") != nullptr;  // developer switch: stage times to stderr
    auto now = [] {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    const double t_0 = now();
    const int64_t n = c->n;
    const int top_k = out->top_k;
    h->plan = Plan();            // the resident table (adh_upload_candidates) is replaced
    h->cands_uploaded = false;
    rc = cand_reserve(h, n, c->n_isotope_cols);
    if (rc != ADH_OK) return rc;
    // device tables: with a communicator attached the layout is padded to the largest shard and
    // double-buffered (the all-gather of call i overlaps call i + 1)
    if (h->comm_attached() && n > h->comm_rows)
        return fail(ADH_ERR_INVALID_ARGUMENT,
                    "more candidates than max_rows_per_rank of adh_comm_init: the ranks would disagree on the table layout");
    const int slot = h->comm_attached() ? (h->table_slot ^= 1) : 0;
    rc = comm_wait_slot(h, slot);
    if (rc != ADH_OK) return rc;
    rc = ensure_tables(h, slot, std::max<int64_t>(n, h->comm_attached() ? h->comm_rows : 0), top_k);
    if (rc != ADH_OK) return rc;
    DevTables &tab = h->tables[slot];
    h->last_tables = slot;
    h->last_rows = n;
    {
        const double t_s = now();
        const bool was = g_big_free.load();
        settle_copy_path(h);
        if (timing && was) fprintf(stderr, "[adh] copy path settled in %.1f ms\n", now() - t_s);
    }
    if (cop) cop->n_rows = cop->n_slots = 0;
    if (n == 0) return comm_gather_slot(h, slot);
    std::vector<hipEvent_t> chunk_done;
    // a call that fails half way leaves no tables behind (a reader would rebuild columns of a half-filled
    // table), and its events go back to the pool
    std::vector<hipEvent_t> aux_events;  // (events of the compacted copy-out, returned with the others)
    struct Unwind {
        adh_handle *h;
        DevTables &tab;
        std::vector<hipEvent_t> &events;
        std::vector<hipEvent_t> &aux;
        bool ok = false;
        ~Unwind() {
            for (hipEvent_t ev : events) h->free_events.push_back(ev);
            events.clear();
            for (hipEvent_t ev : aux) h->free_events.push_back(ev);
            aux.clear();
            if (!ok) {
                (void)hipDeviceSynchronize();
                (void)hipGetLastError();
                h->last_tables = -1;
                h->last_rows = 0;
                tab.partial = false;
            }
        }
    } unwind{h, tab, chunk_done, aux_events};
    adh_output_t dev = tab.view;
    dev.n = n;
    hipStream_t sk = h->stream, si = h->stream_in, so = h->stream_out;
    HIP_TRY(hipMemsetAsync(tab.base, 0, tab.used, sk));
    tab.partial = false;

    // chunk boundaries: a short first chunk (its H2D, plan and kernels are the un-overlapped ramp
    // of the D2H-bound pipeline), then equal chunks
    int64_t chunk = pick_chunk(n, !h->tims_staged);
    if (h->tims_staged && n > 0) {
        // ion-mobility candidates reserve a scratch block sized for their dense tiles (116 KB at 38 scans x 29
        // cycles, although ~1 % of it is touched): bound the chunk so that the slab stays within a third of the
        // free device memory.  Tile sizes from the host columns; K <= top_k fragments, 3 observations assumed.
        int64_t cells = 1;
        for (int64_t i = 0; i < n; ++i) {
            const int64_t S = std::max<int64_t>(c->scan_stop[i] - c->scan_start[i], 1);
            const int64_t F = std::max<int64_t>((c->frame_stop[i] - c->frame_start[i]) / std::max(h->tims.cycle_len, 1) + 1, 1);
            cells = std::max(cells, S * F);
        }
        const int64_t k_max = std::max<int64_t>(std::min<int64_t>(cfg->top_k_fragments, 64), 1);
        const uint64_t block = (uint64_t)cells * 8 * (uint64_t)(k_max * 3 + 4) + 4096;
        if (h->im_scratch_budget == 0) {  // (asked once per staged run: hipMemGetInfo takes ~2 ms)
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                h->im_scratch_budget = std::max<uint64_t>((free_b + h->scratch_slab_bytes) / 3, 1ull << 30);
        }
        if (h->im_scratch_budget) {
            const uint64_t budget = h->im_scratch_budget;
            if (block > budget)
                return fail(ADH_ERR_UNSUPPORTED, "one ion-mobility candidate's tiles exceed a third of the free device memory");
            const int64_t fit = (int64_t)std::max<uint64_t>(budget / block, 64);  // (the budget holds, down to 64 rows per chunk)
            if (fit < chunk) {
                const int64_t parts = (n + fit - 1) / fit;
                chunk = (n + parts - 1) / parts;
            }
        }
    }
    // rebuildable columns: not copied back, rebuilt on the host from fragment_lib_slot (see kOutFields)
    if (cop)  // (a chunk's slot offsets are 32-bit words of the packed count)
        chunk = std::min<int64_t>(chunk, std::max<int64_t>((int64_t)(0xFFFFFFFFull / (uint64_t)top_k) - 1, 1));
    const bool rebuild = !cop && h->h_lib.size() == (size_t)h->n_lib && !getenv("ADH_DEBUG_COPY_ALL") && host_rebuild_pays();
    uint16_t *slot_host = out->fragment_lib_slot;
    if (rebuild && !slot_host) {
        const size_t need = (size_t)n * (size_t)top_k * sizeof(uint16_t);
        if (h->slot_stage_bytes < need) {
            if (h->slot_stage) (void)hipHostFree(h->slot_stage);
            h->slot_stage = nullptr;
            h->slot_stage_bytes = 0;
            HIP_TRY(hipHostMalloc(&h->slot_stage, need + need / 8, hipHostMallocDefault));
            h->slot_stage_bytes = need + need / 8;
        }
        slot_host = static_cast<uint16_t *>(h->slot_stage);
    }
    adh_output_t dev_k = dev;  // what the kernels write
    if ((rebuild || cop) && !getenv("ADH_DEBUG_WRITE_ALL")) {
        // ... and the kernels need not write them either: nobody on the host waits for them, and a reader of the
        // device tables (adh_get_device_tables, the resident FDR stage) gets them filled in on demand
        for (int i = 0; i < kNumOutFields; ++i) {
            const OutFieldDesc &f = kOutFields[i];
            if (!f.wire && f.member != offsetof(adh_output_t, stat_matched_peaks)) *out_member(&dev_k, f) = nullptr;
        }
        tab.partial = true;
    }
    std::vector<int64_t> cut{0};
    // a short first chunk: its copy-out starts early.  A quarter of a chunk - or half of one for a table of many
    // chunks, where the copy-out is what the call waits for and the first copy-out should cover the kernels of the
    // (full) second chunk: 3 M candidates 29.3 -> 28.6 ms; tables of two or three chunks lose with it
    // (ADH_FIRST_CHUNK_DIV fixes the divisor)
    const char *first_div_str = getenv("ADH_FIRST_CHUNK_DIV");
    const int first_div_env = first_div_str && atoi(first_div_str) > 0 ? atoi(first_div_str) : 0;
    // (tables of up to three chunks: no short first chunk - 48 000 rows 1.45 against 1.65 ms, 96 000 1.93 against 2.02)
    const int first_div = first_div_env ? first_div_env : (n >= 4 * chunk ? 2 : ((n > 3 * chunk || h->tims_staged) ? 4 : 1));
    if (n > chunk) cut.push_back(std::max<int64_t>(chunk / first_div, 1));
    while (cut.back() < n) cut.push_back(std::min(n, cut.back() + chunk));
    const int64_t n_chunks = (int64_t)cut.size() - 1;
    // compacted copy-out of the fragment tables (see adh_slot_count_kernel).  OFF unless ADH_COMPACT_COPY_OUT=1: it takes
    // a third of the bytes off PCIe (1.35 -> 0.86 GB per 3 M candidates) but hands the host team 0.9 GB of padded rows
    // to write, and on the pool's boxes - 16 cores' worth of CPU quota - that costs more than the copies it saves
    // (40.0 against 38.3 ms per step on the same box, 37.6 with 32 threads).  A host with cores to spare can turn it on.
    const char *cmp_env = getenv("ADH_COMPACT_COPY_OUT");
    const bool compact = rebuild && top_k <= 65535 && cmp_env && atoi(cmp_env) != 0;
    const CompactLayout clay(n, n_chunks, top_k);
    std::vector<hipEvent_t> &off_ready = aux_events;  // per chunk: its offsets are on the host
    if (compact) {
        if (h->cmp_dev_bytes < clay.total) {
            HIP_TRY(hipDeviceSynchronize());
            if (h->cmp_dev) (void)hipFree(h->cmp_dev);
            h->cmp_dev = nullptr;
            h->cmp_dev_bytes = 0;
            HIP_TRY(hipMalloc(&h->cmp_dev, clay.total + clay.total / 8));
            h->cmp_dev_bytes = clay.total + clay.total / 8;
        }
        if (h->cmp_host_bytes < clay.total) {
            if (h->cmp_host) (void)hipHostFree(h->cmp_host);
            h->cmp_host = nullptr;
            h->cmp_host_bytes = 0;
            HIP_TRY(hipHostMalloc(&h->cmp_host, clay.total + clay.total / 8, hipHostMallocDefault));
            h->cmp_host_bytes = clay.total + clay.total / 8;
        }
        int64_t longest = 0;
        for (int64_t ci = 0; ci < n_chunks; ++ci) longest = std::max(longest, cut[(size_t)ci + 1] - cut[(size_t)ci]);
        size_t scan_bytes = 0;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)(longest + 1), sk));
        if (h->cmp_scan_bytes < scan_bytes) {
            HIP_TRY(hipDeviceSynchronize());
            if (h->cmp_scan) (void)hipFree(h->cmp_scan);
            h->cmp_scan = nullptr;
            h->cmp_scan_bytes = 0;
            HIP_TRY(hipMalloc(&h->cmp_scan, scan_bytes + 256));
            h->cmp_scan_bytes = scan_bytes + 256;
        }
    }
    // operator path: counts / offsets of every chunk (one uint64 per row + one per chunk), scan scratch, staging block
    const CopLayout cop_lay(n, top_k, n_chunks);
    if (cop) {
        if (!cop->row || !cop->precursor_idx || !cop->rank || !cop->features || !cop->fragment_row ||
            !cop->fragment_precursor_idx || !cop->fragment_rank || !cop->fragment_mz_library || !cop->fragment_mz ||
            !cop->fragment_mz_observed || !cop->fragment_height || !cop->fragment_intensity || !cop->fragment_mass_error ||
            !cop->fragment_correlation || !cop->fragment_position || !cop->fragment_number || !cop->fragment_type ||
            !cop->fragment_charge || !cop->fragment_loss_type)
            return fail(ADH_ERR_INVALID_ARGUMENT, "compact output buffer is NULL");
        // the caller's arrays are usually fresh allocations: ask for huge pages where the kernel gives them on request
        // (270 000 first-touch faults of 4 KiB pages per 3 M candidates otherwise, taken by the copying threads)
        {
            auto advise = [](void *p, size_t bytes) {
                const uintptr_t lo = ((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
                const uintptr_t hi = ((uintptr_t)p + bytes) & ~(uintptr_t)((2u << 20) - 1);
                if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
            };
            const size_t rc_ = (size_t)cop->rows_capacity, sc_ = (size_t)cop->slots_capacity;
            advise(cop->features, rc_ * ADH_NUM_FEATURES * 4);
            advise(cop->row, rc_ * 4), advise(cop->precursor_idx, rc_ * 4);
            void *const s4[] = {cop->fragment_row, cop->fragment_precursor_idx, cop->fragment_mz_library, cop->fragment_mz,
                                cop->fragment_mz_observed, cop->fragment_height, cop->fragment_intensity,
                                cop->fragment_mass_error, cop->fragment_correlation};
            for (void *p4 : s4) advise(p4, sc_ * 4);
            void *const s1[] = {cop->fragment_rank, cop->fragment_position, cop->fragment_number, cop->fragment_type,
                                cop->fragment_charge, cop->fragment_loss_type};
            for (void *p1 : s1) advise(p1, sc_);
        }
        const size_t cnt_bytes = (size_t)(n + n_chunks) * 8;
        if (h->cop_cnt_bytes < cnt_bytes) {
            HIP_TRY(hipDeviceSynchronize());
            if (h->cop_cnt) (void)hipFree(h->cop_cnt);
            h->cop_cnt = nullptr, h->cop_cnt_bytes = 0;
            HIP_TRY(hipMalloc(&h->cop_cnt, cnt_bytes + cnt_bytes / 8));
            h->cop_cnt_bytes = cnt_bytes + cnt_bytes / 8;
        }
        int64_t longest = 0;
        for (int64_t ci = 0; ci < n_chunks; ++ci) longest = std::max(longest, cut[(size_t)ci + 1] - cut[(size_t)ci]);
        size_t scan_bytes = 0;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (int)(longest + 1), sk));
        if (h->cop_scan_bytes < scan_bytes) {
            HIP_TRY(hipDeviceSynchronize());
            if (h->cop_scan) (void)hipFree(h->cop_scan);
            h->cop_scan = nullptr, h->cop_scan_bytes = 0;
            HIP_TRY(hipMalloc(&h->cop_scan, scan_bytes + 256));
            h->cop_scan_bytes = scan_bytes + 256;
        }
        if (h->cop_stage_bytes < cop_lay.total) {
            if (h->cop_stage) (void)hipHostFree(h->cop_stage);
            h->cop_stage = nullptr, h->cop_stage_bytes = 0;
            HIP_TRY(hipHostMalloc(&h->cop_stage, cop_lay.total + cop_lay.total / 8, hipHostMallocDefault));
            h->cop_stage_bytes = cop_lay.total + cop_lay.total / 8;
        }
        if (n_chunks > 4096) return fail(ADH_ERR_UNSUPPORTED, "compact output: more than 4096 chunks");
        if (!h->cop_tot_pinned) HIP_TRY(hipHostMalloc((void **)&h->cop_tot_pinned, 4096 * 8, hipHostMallocDefault));
        if (h->cop_dev_bytes < cop_lay.total) {
            HIP_TRY(hipDeviceSynchronize());
            if (h->cop_dev) (void)hipFree(h->cop_dev);
            h->cop_dev = nullptr, h->cop_dev_bytes = 0;
            const size_t want = cop_lay.total + cop_lay.total / 8 + 4096;
            HIP_TRY(hipMalloc(&h->cop_dev, want));
            h->cop_dev_bytes = want;
        }
    }
    unsigned char *const cop_stage = static_cast<unsigned char *>(h->cop_stage);
    unsigned char *const cop_dev = static_cast<unsigned char *>(h->cop_dev);
    std::vector<hipEvent_t> cop_tot_ready;   // per chunk: its totals are on the host (events return with aux_events)
    std::vector<uint64_t> &cop_tot_host = h->cop_tot_host;
    if (cop) cop_tot_host.assign((size_t)n_chunks, 0);
    // the block of chunk ci: wait for its totals (its kernels are done then), ONE copy of the used bytes
    auto flush_cop = [&](int64_t ci) -> int {
        HIP_TRY(hipEventSynchronize(cop_tot_ready[(size_t)ci]));
        const uint64_t tot = h->cop_tot_pinned[ci];
        cop_tot_host[(size_t)ci] = tot;
        const CopBlock L(tot >> 32, tot & 0xFFFFFFFFull);
        const size_t base = cop_lay.base(cut[(size_t)ci], ci);
        if (L.total > 0) HIP_TRY(hipMemcpyAsync(cop_stage + base, cop_dev + base, L.total, hipMemcpyDeviceToHost, so));
        h->d2h_bytes += L.total + 8;
        hipEvent_t ev = nullptr;
        int rc_e = get_event(h, &ev);
        if (rc_e != ADH_OK) return rc_e;
        HIP_TRY(hipEventRecord(ev, so));
        chunk_done.push_back(ev);
        return ADH_OK;
    };
    // host threads unpack finished blocks into the caller's arrays WHILE the later chunks are enqueued and scored (the
    // enqueue loop waits for every chunk's totals, so it takes as long as the kernels: with the team started behind it
    // the unpacking - 2 ms per 450 000-row block - came on top: 32 ms per 3 M candidates).  The calling thread
    // publishes blocks as their copies complete (cop_publish), thread w takes stripe w of T of every block.
    // base[ci] / cnt[ci] are written once, before `ready` passes ci, and never again: a worker reads only its own
    // chunk's pair (a chunk that does not fit - and every one behind it - is published with a count of 0).
    std::vector<int64_t> cop_base_r((size_t)n_chunks + 1, 0), cop_base_s((size_t)n_chunks + 1, 0);
    std::vector<int64_t> cop_cnt_r((size_t)n_chunks, 0), cop_cnt_s((size_t)n_chunks, 0);
    bool cop_overflow = false;
    int64_t cop_published = 0;
    const int cop_T = cop ? host_threads_for(n) : 0;
    int cop_started = 0;
    struct CopTeam {  // (declared behind what its threads read: it is destroyed - joined - first)
        std::vector<std::thread> threads;
        std::atomic<int64_t> ready{0};
        std::atomic<bool> abort{false};
        void join_all() {
            for (std::thread &t : threads)
                if (t.joinable()) t.join();
        }
        ~CopTeam() {  // (an early return: whoever still waits for a block gives up)
            abort.store(true);
            join_all();
        }
    } cop_team;
    auto cop_stripe = [&](int64_t ci, int w) {
        if (cop_cnt_r[(size_t)ci] == 0 && cop_cnt_s[(size_t)ci] == 0) return;
        cop_copy_stripe(cop_stage + cop_lay.base(cut[(size_t)ci], ci), cop_cnt_r[(size_t)ci], cop_cnt_s[(size_t)ci],
                        cop_base_r[(size_t)ci], cop_base_s[(size_t)ci], cop, w, cop_T, c, h->h_lib.data());
    };
    auto cop_worker = [&](int w) {
        for (int64_t ci = 0; ci < n_chunks; ++ci) {
            while (cop_team.ready.load(std::memory_order_acquire) <= ci) {
                if (cop_team.abort.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            cop_stripe(ci, w);
        }
    };
    for (int w = 0; w < cop_T; ++w) {
        try {
            cop_team.threads.emplace_back(cop_worker, w);
            ++cop_started;
        } catch (const std::system_error &) {
            break;
        }
    }
    auto cop_publish = [&](bool wait) -> hipError_t {
        while (cop_published < (int64_t)chunk_done.size()) {
            const int64_t ci = cop_published;
            const hipError_t q = wait ? hipEventSynchronize(chunk_done[(size_t)ci]) : hipEventQuery(chunk_done[(size_t)ci]);
            if (q == hipErrorNotReady) return hipSuccess;
            if (q != hipSuccess) return q;
            const uint64_t tot = cop_tot_host[(size_t)ci];
            cop_base_r[(size_t)ci + 1] = cop_base_r[(size_t)ci] + (int64_t)(tot >> 32);
            cop_base_s[(size_t)ci + 1] = cop_base_s[(size_t)ci] + (int64_t)(tot & 0xFFFFFFFFull);
            if (cop_base_r[(size_t)ci + 1] > cop->rows_capacity || cop_base_s[(size_t)ci + 1] > cop->slots_capacity)
                cop_overflow = true;
            // count on (the caller learns what it needs); an overflowing chunk and all behind it copy nothing
            cop_cnt_r[(size_t)ci] = cop_overflow ? 0 : (int64_t)(tot >> 32);
            cop_cnt_s[(size_t)ci] = cop_overflow ? 0 : (int64_t)(tot & 0xFFFFFFFFull);
            if (timing)
                fprintf(stderr, "[adh]   compact chunk %lld: %lld rows, %lld slots on the host %.2f ms after the call began\n",
                        (long long)ci, (long long)(tot >> 32), (long long)(tot & 0xFFFFFFFFull), now() - t_0);
            cop_team.ready.store(ci + 1, std::memory_order_release);
            ++cop_published;
        }
        return hipSuccess;
    };
    unsigned char *const cmp_dev = static_cast<unsigned char *>(h->cmp_dev), *const cmp_host = static_cast<unsigned char *>(h->cmp_host);
    const double t_1 = now();
    const bool dbg_events = timing && atoi(getenv("ADH_DEBUG_TIMING")) >= 2;  // per-chunk D2H spans
    std::vector<hipEvent_t> dbg;
    std::vector<uint64_t> dbg_bytes;  // copy-out bytes of every chunk
    hipEvent_t dbg_start = nullptr;   // on the copy-in stream, before the first column goes up
    auto fail_sync = [&](int code) {
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
        return code;
    };
    // packed fragment columns of chunk ci: wait for its offsets (its kernels are done then; the next chunk's are already
    // queued), copy the filled part of the six columns, mark the chunk complete for the host team
    auto flush_compact = [&](int64_t ci) -> int {
        const int64_t a = cut[(size_t)ci], b = cut[(size_t)ci + 1];
        HIP_TRY(hipEventSynchronize(off_ready[(size_t)ci]));
        const uint32_t total = reinterpret_cast<const uint32_t *>(cmp_host + clay.off_of(a, ci))[b - a];
        if (total > 0) {
            const size_t e0 = (size_t)a * (size_t)top_k;
            for (int j = 0; j < 5; ++j)
                HIP_TRY(hipMemcpyAsync(cmp_host + clay.col_f(j) + e0 * 4, cmp_dev + clay.col_f(j) + e0 * 4, (size_t)total * 4,
                                       hipMemcpyDeviceToHost, so));
            HIP_TRY(hipMemcpyAsync(cmp_host + clay.col_slot() + e0 * 2, cmp_dev + clay.col_slot() + e0 * 2, (size_t)total * 2,
                                   hipMemcpyDeviceToHost, so));
            h->d2h_bytes += (uint64_t)total * 22;
        }
        hipEvent_t ev = nullptr;
        int rc_e = get_event(h, &ev);
        if (rc_e != ADH_OK) return rc_e;
        HIP_TRY(hipEventRecord(ev, so));
        chunk_done.push_back(ev);
        return ADH_OK;
    };
    // Chunk 0: its columns + plan on the copy-in stream.  The columns of ALL later chunks follow in
    // one go right behind it (one H2D per column): H2D copies issued while the D2H copies of earlier
    // chunks are in flight slowed those down four-fold for two chunks on MI355X (measured; the
    // copy engines are shared), whereas one early burst overlaps only the kernels of chunk 0.
    // (ADH_H2D_BURST_LATE=0: the whole burst before the plan of chunk 1, the order of round 3)
    const char *late_env = getenv("ADH_H2D_BURST_LATE");
    const bool late_burst = !(late_env && atoi(late_env) == 0);
    if (dbg_events) {
        (void)hipEventCreate(&dbg_start);
        (void)hipEventRecord(dbg_start, si);
    }
    rc = cand_upload_range(h, c, 0, cut[1], si);
    if (rc == ADH_OK) rc = plan_enqueue(h, h->slots[0], cfg, 0, cut[1], si);
    // ... of which chunk 1's rows go first and the rest behind the plan of chunk 1: its kernels - and with them the
    // second copy-out - start ~4 ms earlier than behind the whole burst, and the copy-out stream has no gap to wait out
    if (rc == ADH_OK && n_chunks > 1) rc = cand_upload_range(h, c, cut[1], late_burst ? cut[2] : n, si);
    if (rc != ADH_OK) return fail_sync(rc);
    for (int64_t ci = 0; ci < n_chunks; ++ci) {
        const int64_t a = cut[(size_t)ci], b = cut[(size_t)ci + 1];
        const int ps = (int)(ci & 1);
        if (cop && cop_publish(false) != hipSuccess) return fail_sync(fail(ADH_ERR_HIP, "scoring pipeline (compact copy-out)"));
        if (ci + 1 < n_chunks) {
            // plan of the next chunk: the other plan slot is free once the kernels of chunk ci - 1 are done
            const int64_t a2 = b, b2 = cut[(size_t)ci + 2];
            if (ci >= 1) HIP_TRY(hipStreamWaitEvent(si, h->ev_k[ps ^ 1], 0));
            rc = plan_enqueue(h, h->slots[ps ^ 1], cfg, a2, b2 - a2, si);
            if (rc == ADH_OK && ci == 0 && late_burst && n_chunks > 2) rc = cand_upload_range(h, c, cut[2], n, si);
            if (rc != ADH_OK) return fail_sync(rc);
        }
        Plan p;
        rc = plan_finish(h, h->slots[ps], cfg, a, b - a, p);
        if (rc == ADH_OK) rc = launch_scoring(h, p, cfg, &dev_k, sk);
        if (rc != ADH_OK) return fail_sync(rc);
        if (compact) {  // filled slots per row, their offsets, the packed columns: behind the chunk's kernels
            uint32_t *d_off = reinterpret_cast<uint32_t *>(cmp_dev + clay.off_of(a, ci));
            const int64_t nr = b - a;
            hipLaunchKernelGGL(adh_slot_count_kernel, dim3((unsigned)((nr + 256) / 256)), dim3(256), 0, sk, dev.fragment_lib_slot, a,
                               nr, top_k, d_off);
            size_t scan_bytes = h->cmp_scan_bytes;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(h->cmp_scan, scan_bytes, d_off, d_off, (int)(nr + 1), sk));
            CompactCols cc;
            for (int j = 0; j < 5; ++j) cc.f[j] = reinterpret_cast<float *>(cmp_dev + clay.col_f(j)) + (size_t)a * top_k;
            cc.slot = reinterpret_cast<uint16_t *>(cmp_dev + clay.col_slot()) + (size_t)a * top_k;
            hipLaunchKernelGGL(adh_compact_kernel, dim3((unsigned)((nr * top_k + 255) / 256)), dim3(256), 0, sk, dev, a, nr, top_k,
                               d_off, cc);
            HIP_TRY(hipGetLastError());
        }
        uint64_t *cop_off = cop ? static_cast<uint64_t *>(h->cop_cnt) + a + ci : nullptr;
        if (cop) {
            const int64_t nr = b - a;
            hipLaunchKernelGGL(adh_cop_count_kernel, dim3((unsigned)((nr + 256) / 256)), dim3(256), 0, sk, dev.valid,
                               dev.fragment_lib_slot, a, nr, top_k, cop_off);
            size_t scan_bytes = h->cop_scan_bytes;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(h->cop_scan, scan_bytes, cop_off, cop_off, (int)(nr + 1), sk));
            hipLaunchKernelGGL(adh_cop_pack_kernel, dim3((unsigned)std::min<int64_t>((nr * top_k + 255) / 256, 8192)), dim3(256), 0,
                               sk, dev, h->cs.d, h->d_lib, a, nr, top_k, cop_off, cop_dev + cop_lay.base(a, ci), h->cop_tot_pinned + ci);
            HIP_TRY(hipGetLastError());
            hipEvent_t ev = nullptr;  // (when the totals can be read)
            rc = get_event(h, &ev);
            if (rc != ADH_OK) return fail_sync(rc);
            HIP_TRY(hipEventRecord(ev, sk));
            aux_events.push_back(ev);
            cop_tot_ready.push_back(ev);
        }
        HIP_TRY(hipEventRecord(h->ev_k[ps], sk));
        if (cop && ci > 0) {  // the block of the previous chunk, now that the host can know its size
            rc = flush_cop(ci - 1);
            if (rc != ADH_OK) return fail_sync(rc);
        }
        if (compact && ci > 0) {  // the packed columns of the previous chunk, now that the host can know their length
            rc = flush_compact(ci - 1);
            if (rc != ADH_OK) return fail_sync(rc);
        }
        HIP_TRY(hipStreamWaitEvent(so, h->ev_k[ps], 0));
        if (compact) {
            const size_t ob = clay.off_of(a, ci);
            HIP_TRY(hipMemcpyAsync(cmp_host + ob, cmp_dev + ob, (size_t)(b - a + 1) * 4, hipMemcpyDeviceToHost, so));
            h->d2h_bytes += (uint64_t)(b - a + 1) * 4;
            hipEvent_t ev = nullptr;
            rc = get_event(h, &ev);
            if (rc != ADH_OK) return fail_sync(rc);
            HIP_TRY(hipEventRecord(ev, so));
            off_ready.push_back(ev);
        }
        if (dbg_events) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, so);
            dbg.push_back(e0);
            dbg.push_back(e1);
            dbg_bytes.push_back(h->d2h_bytes);
        }
        // (A chunk is up to nine copies and the engine idles ~10 us between two of them - a 47 000-row chunk, 21 MB, takes
        // 0.46 ms = 46 GB/s where each copy runs at 55, `rocprofv3 --memory-copy-trace` - but a second copy-out stream for
        // the feature table does not fill the gaps: measured in round 5, same times to the 0.01 ms, and taken out again.)
        for (int i = 0; i < kNumOutFields && !cop; ++i) {
            const OutFieldDesc &f = kOutFields[i];
            void *host = *out_member(out, f);
            const bool is_slot = f.member == offsetof(adh_output_t, fragment_lib_slot);
            const bool is_stat = f.member == offsetof(adh_output_t, stat_matched_peaks);
            if (is_slot && !host && rebuild) host = slot_host;
            if (!host) continue;
            if (rebuild && !f.wire && !is_stat) continue;  // rebuilt on the host below
            if (compact && f.per_row < 0) continue;        // the fragment tables travel packed (flush_compact)
            const size_t rb = out_row_bytes(f, top_k);
            hipError_t e = hipMemcpyAsync(static_cast<unsigned char *>(host) + (size_t)a * rb,
                                          static_cast<unsigned char *>(*out_member(&dev, f)) + (size_t)a * rb,
                                          (size_t)(b - a) * rb, hipMemcpyDeviceToHost, so);
            if (e != hipSuccess) {
                fail(ADH_ERR_HIP, std::string("hipMemcpyAsync D2H: ") + hipGetErrorString(e));
                return fail_sync(ADH_ERR_HIP);
            }
            h->d2h_bytes += (uint64_t)(b - a) * rb;
        }
        if (rebuild && !compact) {
            hipEvent_t ev = nullptr;
            rc = get_event(h, &ev);
            if (rc != ADH_OK) return fail_sync(rc);
            HIP_TRY(hipEventRecord(ev, so));
            chunk_done.push_back(ev);
        }
        if (dbg_events) {
            (void)hipEventRecord(dbg.back(), so);
            dbg_bytes.back() = h->d2h_bytes - dbg_bytes.back();
        }
    }
    if (compact) {
        rc = flush_compact(n_chunks - 1);
        if (rc != ADH_OK) return fail_sync(rc);
    }
    if (cop) {
        rc = flush_cop(n_chunks - 1);
        if (rc != ADH_OK) return fail_sync(rc);
    }
    const double t_2 = now();
    rc = comm_gather_slot(h, slot);  // after the last chunk's kernels; overlaps the remaining D2H
    if (rc != ADH_OK) return fail_sync(rc);
    if (cop) {
        // the blocks still on their way, then the threads (they have been unpacking since the first block landed)
        hipError_t ee = cop_publish(true);
        if (ee == hipSuccess) cop_team.join_all();
        if (ee != hipSuccess) {
            fail(ADH_ERR_HIP, std::string("scoring pipeline (compact copy-out): ") + hipGetErrorString(ee));
            return fail_sync(ADH_ERR_HIP);
        }
        for (int64_t ci = 0; ci < n_chunks; ++ci)  // (stripes of threads that could not be started)
            for (int w = cop_started; w < cop_T; ++w) cop_stripe(ci, w);
        if (timing) fprintf(stderr, "[adh]   compact: host team done %.2f ms after the call began (%d threads)\n", now() - t_0, cop_T);
        cop->n_rows = cop_base_r[(size_t)n_chunks];
        cop->n_slots = cop_base_s[(size_t)n_chunks];
        if (cop_overflow) {
            (void)hipStreamSynchronize(sk);
            (void)hipStreamSynchronize(so);
            unwind.ok = true;  // (the device tables are complete)
            return fail(ADH_ERR_INVALID_ARGUMENT, "compact output: rows_capacity / slots_capacity too small (n_rows / n_slots say what is needed)");
        }
    }
    if (rebuild) {
        // host threads follow the copy-out stream chunk by chunk: thread w takes the w-th stripe of every chunk
        const int T = host_threads_for(n);
        std::atomic<int64_t> ready{0};
        std::atomic<bool> abort{false};
        auto worker = [&](int w) {
            for (int64_t ci = 0; ci < n_chunks; ++ci) {
                while (ready.load(std::memory_order_acquire) <= ci) {
                    if (abort.load(std::memory_order_relaxed)) return;
                    std::this_thread::yield();
                }
                const int64_t a = cut[(size_t)ci], b = cut[(size_t)ci + 1];
                const int64_t lo = a + (b - a) * w / T, hi = a + (b - a) * (w + 1) / T;
                if (compact) expand_host_rows(out, slot_host, cmp_host, clay, a, ci, lo, hi);
                rebuild_host_rows(h, c, out, slot_host, lo, hi);
            }
        };
        std::vector<std::thread> team;
        for (int w = 1; w < T; ++w) {
            try {
                team.emplace_back(worker, w);
            } catch (const std::system_error &) {
                break;  // (the calling thread takes the stripes that have no thread)
            }
        }
        const int started = (int)team.size() + 1;
        hipError_t ee = hipSuccess;
        for (int64_t ci = 0; ci < n_chunks && ee == hipSuccess; ++ci) {
            ee = hipEventSynchronize(chunk_done[(size_t)ci]);
            if (ee != hipSuccess) break;
            ready.store(ci + 1, std::memory_order_release);
            const int64_t a = cut[(size_t)ci], b = cut[(size_t)ci + 1];
            auto stripe = [&](int w) {
                const int64_t lo = a + (b - a) * w / T, hi = a + (b - a) * (w + 1) / T;
                if (compact) expand_host_rows(out, slot_host, cmp_host, clay, a, ci, lo, hi);
                rebuild_host_rows(h, c, out, slot_host, lo, hi);
            };
            stripe(0);
            for (int w = started; w < T; ++w) stripe(w);
        }
        if (ee != hipSuccess) abort.store(true);
        for (std::thread &t : team) t.join();
        if (ee != hipSuccess) {
            fail(ADH_ERR_HIP, std::string("scoring pipeline (copy-out): ") + hipGetErrorString(ee));
            return fail_sync(ADH_ERR_HIP);
        }
    }
    hipError_t e = hipStreamSynchronize(sk);
    if (e == hipSuccess) e = hipStreamSynchronize(so);
    if (e != hipSuccess) {
        fail(ADH_ERR_HIP, std::string("scoring pipeline: ") + hipGetErrorString(e));
        return fail_sync(ADH_ERR_HIP);
    }
    if (timing)
        fprintf(stderr, "[adh] score_candidates n=%lld in %lld chunks: setup %.2f ms, enqueue %.2f, drain %.2f\n",
                (long long)n, (long long)n_chunks, t_1 - t_0, t_2 - t_1, now() - t_2);
    if (dbg_events) {
        for (size_t i = 0; i + 1 < dbg.size(); i += 2) {
            float ms = 0.f, since = 0.f;
            (void)hipEventElapsedTime(&ms, dbg[i], dbg[i + 1]);
            (void)hipEventElapsedTime(&since, dbg[0], dbg[i]);
            float from_start = 0.f;
            if (dbg_start) (void)hipEventElapsedTime(&from_start, dbg_start, dbg[i]);
            fprintf(stderr, "[adh]   chunk %zu (%lld rows): D2H starts %.2f ms after the first (%.2f ms after the call's first copy-in), "
                            "lasts %.2f ms for %.1f MB = %.1f GB/s\n",
                    i / 2, (long long)(cut[i / 2 + 1] - cut[i / 2]), since, from_start, ms, (double)dbg_bytes[i / 2] / 1e6,
                    (double)dbg_bytes[i / 2] / 1e6 / std::max(ms, 1e-3f));
        }
        if (dbg_start) (void)hipEventDestroy(dbg_start);
        for (hipEvent_t e : dbg) (void)hipEventDestroy(e);
    }
    unwind.ok = true;
    return ADH_OK;
}
}  // namespace

int adh_score_candidates(adh_handle_t *h, const adh_candidates_t *c, const adh_scoring_config_t *cfg,
                         adh_output_t *out) {
    return score_pipeline(h, c, cfg, out, nullptr);
}

int adh_score_candidates_compact(adh_handle_t *h, const adh_candidates_t *c, const adh_scoring_config_t *cfg,
                                 adh_compact_output_t *cop) {
    if (!h || !c || !cfg || !cop) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (cop->top_k <= 0 || cop->rows_capacity < 0 || cop->slots_capacity < 0)
        return fail(ADH_ERR_INVALID_ARGUMENT, "compact output: top_k / capacities");
    adh_output_t shape{};
    shape.n = c->n;
    shape.top_k = cop->top_k;
    return score_pipeline(h, c, cfg, &shape, cop);
}

int adh_table_layout(int64_t rows, int32_t top_k, int32_t capacity, adh_table_field_t *fields, int32_t *n_fields,
                     uint64_t *total_bytes, uint64_t *wire_bytes) {
    if (rows < 0 || top_k <= 0 || !n_fields) return fail(ADH_ERR_INVALID_ARGUMENT, "adh_table_layout: bad argument");
    *n_fields = kNumOutFields;
    size_t wire = 0;
    const size_t total = layout_tables(nullptr, rows, top_k, nullptr, &wire);
    if (total_bytes) *total_bytes = total;
    if (wire_bytes) *wire_bytes = wire;
    if (!fields) return ADH_OK;
    if (capacity < kNumOutFields) return fail(ADH_ERR_INVALID_ARGUMENT, "adh_table_layout: field array too short");
    size_t off = 0;
    for (int i = 0; i < kNumOutFields; ++i) {
        const OutFieldDesc &f = kOutFields[i];
        adh_table_field_t &o = fields[i];
        memset(&o, 0, sizeof(o));
        snprintf(o.name, sizeof(o.name), "%s", f.name);
        o.offset = off;
        o.row_elems = (uint32_t)(f.per_row < 0 ? top_k : f.per_row);
        o.elem_bytes = (uint32_t)f.elem;
        o.wire = f.wire ? 1 : 0;
        off += ((size_t)rows * out_row_bytes(f, top_k) + 255) / 256 * 256;
    }
    return ADH_OK;
}

int adh_get_device_tables(adh_handle_t *h, adh_output_t *device_view) {
    if (!h || !device_view) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (h->last_tables < 0) return fail(ADH_ERR_NOT_STAGED, "no adh_score_candidates call has filled the device tables");
    const int rc = materialise_tables(h);
    if (rc != ADH_OK) return rc;
    *device_view = h->tables[h->last_tables].view;
    device_view->n = h->last_rows;
    return ADH_OK;
}

int adh_debug_get_dense(adh_handle_t *h, int64_t frame_start, int64_t frame_stop, int64_t scan_start, int64_t scan_stop,
                        const float *mz_query, int32_t n_query, float mass_tolerance, float quad_lo, float quad_hi,
                        float *dense, int64_t dense_capacity, int32_t *obs_out, int32_t *n_obs, int32_t *n_scans,
                        int32_t *n_cycles) {
    if (!h || !mz_query || !dense || !obs_out || !n_obs || !n_scans || !n_cycles)
        return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!h->run_staged && !h->tims_staged) return fail(ADH_ERR_NOT_STAGED, "no run staged");
    if (n_query < 1 || n_query > 4096) return fail(ADH_ERR_INVALID_ARGUMENT, "n_query must be in 1..4096");
    HIP_TRY(hipSetDevice(h->device));
    const bool im = h->tims_staged;
    const int L = im ? h->tims.cycle_len : h->run.cycle_len;
    const int z = im ? h->tims.zeroth : 0;
    const int64_t n_fr = im ? h->tims.n_frames : h->run.n_spectra;
    if (frame_start < z || frame_stop < frame_start || frame_stop > n_fr || (frame_start - z) % L != 0)
        return fail(ADH_ERR_INVALID_ARGUMENT, "frame limits outside the staged run / not on a cycle boundary");
    const int K = n_query;
    const int F = (int)((frame_stop - z) / L - (frame_start - z) / L);
    const int S = im ? (int)(scan_stop - scan_start) : 1;
    if (im && (scan_start < 0 || scan_stop < scan_start || scan_stop > h->tims.scan_max))
        return fail(ADH_ERR_INVALID_ARGUMENT, "scan limits outside the staged run");
    // observations: the quadrupole test of get_dense (alpharaw_jit.py:19-50 / bruker_jit.py:315-350)
    const double *cyc = h->h_cycle.data();
    std::vector<uint16_t> obs;
    if (!im) {
        for (int row = 0; row < L; ++row)
            if ((double)quad_lo <= cyc[2 * row + 1] && (double)quad_hi >= cyc[2 * row]) obs.push_back((uint16_t)row);
    } else {
        std::vector<uint8_t> seen((size_t)L, 0);
        for (int fr = 0; fr < L; ++fr)
            for (int64_t sc = scan_start; sc < scan_stop; ++sc) {
                const int64_t rowi = (int64_t)fr * h->tims.scan_max + sc;
                if ((double)quad_lo <= cyc[2 * rowi + 1] && (double)quad_hi >= cyc[2 * rowi]) seen[(size_t)h->h_dpc[(size_t)rowi]] = 1;
            }
        for (int v = 0; v < L; ++v)
            if (seen[(size_t)v]) obs.push_back((uint16_t)v);
    }
    const int O = (int)obs.size();
    if (O > ADH_MAX_OBS) return fail(ADH_ERR_UNSUPPORTED, "the query overlaps more than 8 cycle rows");
    *n_obs = O;
    *n_scans = S;
    *n_cycles = std::max(F, 0);
    for (int o = 0; o < O; ++o) obs_out[o] = obs[(size_t)o];
    const int64_t need = 2ll * K * O * S * std::max(F, 0);
    if (need > dense_capacity) return fail(ADH_ERR_INVALID_ARGUMENT, "dense buffer too small");
    for (int64_t i = 0; i < need; ++i) dense[i] = 0.0f;
    if (O == 0 || F <= 0 || S <= 0) return ADH_OK;

    // a one-precursor library whose fragments are the query (all kept: distinct descending intensities)
    std::vector<LibRec> lib((size_t)K);
    for (int k = 0; k < K; ++k) {
        memset(&lib[(size_t)k], 0, sizeof(LibRec));
        lib[(size_t)k].mz_library = lib[(size_t)k].mz = mz_query[k];
        lib[(size_t)k].intensity = (float)(K - k);
        lib[(size_t)k].cardinality = 1;
    }
    adh_scoring_config_t cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.collect_fragments = 1;
    cfg.top_k_fragments = (uint32_t)K;
    cfg.top_k_isotopes = 1;
    cfg.reference_channel = -1;
    cfg.quant_window = 3;
    cfg.precursor_mz_tolerance = 10.0f;
    cfg.fragment_mz_tolerance = mass_tolerance;
    Caps caps{K, O, F, 1, K, ADH_DEBUG_DENSE, S, 0, quad_lo, quad_hi};
    const uint64_t sbytes = im ? adh_im_scratch_bytes((uint32_t)K, O, S, F, 1, 0) : adh_scratch_bytes((uint32_t)K, O, F, 1);
    DeviceBuffers tmp;
    const LibRec *d_lib = nullptr;
    int rc = upload(tmp, lib.data(), (int64_t)K, &d_lib, h->stream);
    unsigned char *d_scr = nullptr;
    uint32_t *d_small = nullptr;  // precursor_idx[1] + rank[1] of the kernel's bookkeeping writes
    if (rc == ADH_OK) {
        hipError_t e = hipMalloc((void **)&d_scr, sbytes);
        if (e == hipSuccess) tmp.ptrs.push_back(d_scr);
        if (e == hipSuccess) e = hipMalloc((void **)&d_small, 64);
        if (e == hipSuccess) tmp.ptrs.push_back(d_small);
        if (e != hipSuccess) rc = fail(ADH_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    adh_output_t out;
    memset(&out, 0, sizeof(out));
    out.n = 1;
    out.top_k = K;
    out.precursor_idx = d_small;
    out.rank = reinterpret_cast<uint8_t *>(d_small + 4);
    hipError_t e = hipSuccess;
    if (rc == ADH_OK && !im) {
        CandRec r;
        memset(&r, 0, sizeof(r));
        r.frag_stop = (uint32_t)K;
        r.frame_start = (int32_t)frame_start;
        r.frame_stop = (int32_t)frame_stop;
        r.frame_center = (int32_t)frame_start;
        r.scan_stop = 1;
        r.precursor_mz = 500.0f;
        r.charge = 1;
        r.n_obs = (uint8_t)O;
        for (int o = 0; o < O; ++o) r.obs[o] = obs[(size_t)o];
        r.k_cap = (uint32_t)K;
        const CandRec *d_rec = nullptr;
        rc = upload(tmp, &r, 1, &d_rec, h->stream);
        const size_t lds = adh_gather_lds_bytes(caps, h->run.n_ms1_obs);
        if (rc == ADH_OK && lds > 160 * 1024) rc = fail(ADH_ERR_UNSUPPORTED, "query too large for the gather kernel's LDS");
        if (rc == ADH_OK) {
            hipLaunchKernelGGL(adh_gather_kernel, dim3(1), dim3(ADH_WAVE), lds, h->stream, h->run, d_lib, d_rec, cfg, 1,
                               d_scr, out, caps);
            e = hipGetLastError();
        }
    } else if (rc == ADH_OK) {
        CandRecIM r;
        memset(&r, 0, sizeof(r));
        r.frag_stop = (uint32_t)K;
        r.frame_start = (int32_t)frame_start;
        r.frame_stop = (int32_t)frame_stop;
        r.frame_center = (int32_t)frame_start;
        r.scan_start = (int32_t)scan_start;
        r.scan_stop = (int32_t)scan_stop;
        r.scan_center = (int32_t)scan_start;
        r.precursor_mz = 500.0f;
        r.charge = 1;
        r.n_obs = (uint8_t)O;
        for (int o = 0; o < O; ++o) r.obs[o] = obs[(size_t)o];
        r.k_cap = (uint32_t)K;
        const CandRecIM *d_rec = nullptr;
        rc = upload(tmp, &r, 1, &d_rec, h->stream);
        const size_t lds = adh_gather_im_lds_bytes(caps);
        if (rc == ADH_OK && lds > 160 * 1024) rc = fail(ADH_ERR_UNSUPPORTED, "query too large for the gather kernel's LDS");
        if (rc == ADH_OK) {
            hipLaunchKernelGGL(adh_gather_im_kernel, dim3(1), dim3(ADH_WAVE), lds, h->stream, h->tims, d_lib, d_rec, cfg, 1,
                               d_scr, out, caps);
            e = hipGetLastError();
        }
    }
    std::vector<unsigned char> host((size_t)sbytes);
    if (rc == ADH_OK && e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (rc == ADH_OK && e == hipSuccess) e = hipMemcpy(host.data(), d_scr, sbytes, hipMemcpyDeviceToHost);
    tmp.release();
    if (rc != ADH_OK) return rc;
    if (e != hipSuccess) return fail(ADH_ERR_HIP, std::string("adh_debug_get_dense: ") + hipGetErrorString(e));
    const uint32_t k_sel = *reinterpret_cast<const uint32_t *>(host.data());
    if ((int)k_sel != K) return fail(ADH_ERR_HIP, "gather kernel did not keep every query fragment");
    const float2 *cells = reinterpret_cast<const float2 *>(host.data() + adh_scratch_frag_off((uint32_t)K));
    const int64_t plane = (int64_t)K * O * S * F;
    for (int k = 0; k < K; ++k)
        for (int o = 0; o < O; ++o)
            for (int sc = 0; sc < S; ++sc)
                for (int f = 0; f < F; ++f) {
                    const float2 v = im ? cells[(((int64_t)k * O + o) * S + sc) * F + f] : cells[((int64_t)o * F + f) * K + k];
                    const int64_t at = (((int64_t)k * O + o) * S + sc) * F + f;
                    dense[at] = v.x;
                    dense[plane + at] = v.y;
                }
    return ADH_OK;
}

int adh_zero_device_tables(adh_handle_t *h, void *hip_stream) {
    if (!h) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL handle");
    if (h->last_tables < 0) return fail(ADH_ERR_NOT_STAGED, "no device tables yet");
    HIP_TRY(hipSetDevice(h->device));
    DevTables &t = h->tables[h->last_tables];
    HIP_TRY(hipMemsetAsync(t.base, 0, t.used, (hipStream_t)hip_stream));
    return ADH_OK;
}

int adh_copy_to_host(adh_handle_t *h, void *dst, const void *src_device, uint64_t bytes) {
    if (!h || (bytes > 0 && (!dst || !src_device))) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    if (bytes > 0) HIP_TRY(hipMemcpy(dst, src_device, (size_t)bytes, hipMemcpyDeviceToHost));
    h->d2h_bytes += bytes;
    return ADH_OK;
}

int adh_transfer_counters(adh_handle_t *h, uint64_t *d2h_bytes, int reset) {
    if (!h || !d2h_bytes) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    *d2h_bytes = h->d2h_bytes;
    if (reset) h->d2h_bytes = 0;
    return ADH_OK;
}

int adh_host_alloc(void **ptr, uint64_t bytes) {
    if (!ptr) return fail(ADH_ERR_INVALID_ARGUMENT, "ptr is NULL");
    *ptr = nullptr;
    HIP_TRY(hipHostMalloc(ptr, std::max<size_t>((size_t)bytes, 64), hipHostMallocPortable));
    return ADH_OK;
}

int adh_trim_device_cache(void) {
    const hipError_t e = adh_dev_trim();
    if (e != hipSuccess) return fail(ADH_ERR_HIP, std::string("adh_trim_device_cache: ") + hipGetErrorString(e));
    return ADH_OK;
}

int adh_host_threads(int64_t n_rows, int32_t *threads, int32_t *cpu_budget) {
    if (threads) *threads = host_threads_for(n_rows);
    if (cpu_budget) *cpu_budget = host_cpu_budget();
    return ADH_OK;
}

int adh_host_free(void *ptr) {
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return ADH_OK;
}

// dst[i] = src[idx[i]] for arrays of reference-counted CPython object pointers (NumPy dtype=object), on several
// threads: the precursor-side string columns of the features frame (proteins, genes, sequence, mods, mod_sites) are
// 13.5 M such pointers per 3 M candidates and NumPy gathers them on one thread.  The CALLER HOLDS THE GIL for the
// whole call (ctypes.PyDLL): nothing else can touch a reference count meanwhile, the threads here only add to them,
// atomically (ob_refcnt is the first word of an object in CPython <= 3.11; alphadia_amd/runtime.py checks that on a
// probe object before it uses this).  dst must be fresh: every entry NULL or `fill` (np.empty gives an object array
// filled with references to None), whose reference count gives back what the gathered pointers replace.
int adh_host_take_objects(void **dst, void *const *src, const int64_t *idx, int64_t n, int64_t n_src, void *fill,
                          int32_t threads) {
    if ((!dst || !src || !idx) && n > 0) return fail(ADH_ERR_INVALID_ARGUMENT, "adh_host_take_objects: NULL argument");
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n / 65536));
    std::atomic<int> bad{0};
    auto work = [&](int w) {
        const int64_t a = n * w / T, b = n * (w + 1) / T;
        intptr_t replaced = 0;
        // runs of one object (the candidates of a precursor follow each other; a column of empty strings is ONE
        // interned object) are counted and added once: sixteen threads adding to one reference count one by one spend
        // their time passing its cache line around (measured: the gather of five columns slower than NumPy's)
        void *run = nullptr;
        intptr_t run_len = 0;
        for (int64_t i = a; i < b; ++i) {
            const int64_t k = idx[i];
            void *const old = dst[i];
            if ((uint64_t)k >= (uint64_t)n_src || (old != nullptr && old != fill)) {
                bad.store(1, std::memory_order_relaxed);
                continue;
            }
            void *o = src[k];
            if (o != run) {
                if (run && run_len) __atomic_fetch_add(reinterpret_cast<intptr_t *>(run), run_len, __ATOMIC_RELAXED);
                run = o;
                run_len = 0;
            }
            ++run_len;
            dst[i] = o;
            replaced += old != nullptr;
        }
        if (run && run_len) __atomic_fetch_add(reinterpret_cast<intptr_t *>(run), run_len, __ATOMIC_RELAXED);
        if (replaced) __atomic_fetch_sub(reinterpret_cast<intptr_t *>(fill), replaced, __ATOMIC_RELAXED);
    };
    std::vector<std::thread> team;
    for (int w = 1; w < T; ++w) {
        try {
            team.emplace_back(work, w);
        } catch (const std::system_error &) {
            for (int v = w; v < T; ++v) work(v);
            break;
        }
    }
    work(0);
    for (std::thread &t : team) t.join();
    if (bad.load()) return fail(ADH_ERR_INVALID_ARGUMENT, "adh_host_take_objects: index out of range or dst not fresh");
    return ADH_OK;
}
