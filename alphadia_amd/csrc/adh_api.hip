// adh_api.hip - host side of libalphadia_hip.so: the C ABI declared in
// include/alphadia_hip.h.  Owns the HBM-resident copies of the run, the fragment
// library and the candidate table, sizes the LDS of the scoring kernel per batch
// and times the kernel with HIP events on its launch stream.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "adh_score.hip"
#include "adh_fragcomp.hip"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            (void)hipGetLastError();                                                        \
            return fail(_e == hipErrorOutOfMemory ? ADH_ERR_OUT_OF_MEMORY : ADH_ERR_HIP,   \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                 \
        }                                                                                   \
    } while (0)

struct DeviceBuffers {
    std::vector<void *> ptrs;
    void release() {
        for (void *p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
};

}  // namespace

struct adh_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    DevRun run{};
    DevLib lib{};
    DevCands cands{};
    bool run_staged = false, lib_staged = false, cands_uploaded = false;
    DeviceBuffers run_buf, lib_buf, cand_buf;
    int32_t *d_maxima = nullptr;
    int32_t plan_n_lib = 0, plan_o = 0, plan_f = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;  // per-launch event pairs
    std::vector<hipEvent_t> free_events;
};

namespace {

template <typename T>
int upload(DeviceBuffers &owner, const T *host, int64_t n, const T **dev, hipStream_t) {
    *dev = nullptr;
    void *p = nullptr;
    size_t bytes = (size_t)std::max<int64_t>(n, 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    owner.ptrs.push_back(p);
    if (n > 0) HIP_TRY(hipMemcpy(p, host, (size_t)n * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<const T *>(p);
    return ADH_OK;
}

#define UP(owner, host, n, dev)                                            \
    do {                                                                   \
        int _rc = upload(owner, host, n, dev, h->stream);                  \
        if (_rc != ADH_OK) return _rc;                                     \
    } while (0)

int get_event(adh_handle *h, hipEvent_t *e) {
    if (!h->free_events.empty()) {
        *e = h->free_events.back();
        h->free_events.pop_back();
        return ADH_OK;
    }
    HIP_TRY(hipEventCreate(e));
    return ADH_OK;
}

}  // namespace

extern "C" {

const char *adh_last_error(void) { return g_last_error.c_str(); }

int adh_device_count(int *count) {
    if (!count) return fail(ADH_ERR_INVALID_ARGUMENT, "count is NULL");
    HIP_TRY(hipGetDeviceCount(count));
    return ADH_OK;
}

int adh_create(adh_handle_t **handle, int device) {
    if (!handle) return fail(ADH_ERR_INVALID_ARGUMENT, "handle is NULL");
    *handle = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n)
        return fail(ADH_ERR_INVALID_ARGUMENT, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    adh_handle *h = new adh_handle();
    h->device = device;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete h;
        return fail(ADH_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    e = hipMalloc((void **)&h->d_maxima, 4 * sizeof(int32_t));
    if (e != hipSuccess) {
        (void)hipStreamDestroy(h->stream);
        delete h;
        return fail(ADH_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    // the scoring kernel may need more than the default 64 KiB of dynamic LDS
    (void)hipFuncSetAttribute((const void *)adh_score_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    *handle = h;
    return ADH_OK;
}

int adh_destroy(adh_handle_t *h) {
    if (!h) return ADH_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    h->run_buf.release();
    h->lib_buf.release();
    h->cand_buf.release();
    for (auto &p : h->timed) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto e : h->free_events) (void)hipEventDestroy(e);
    if (h->d_maxima) (void)hipFree(h->d_maxima);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return ADH_OK;
}

int adh_stage_alpharaw(adh_handle_t *h, const adh_alpharaw_t *d) {
    if (!h || !d) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (d->cycle_len <= 0 || d->cycle_scans <= 0 || d->n_spectra < 0 || d->n_peaks < 0)
        return fail(ADH_ERR_INVALID_ARGUMENT, "invalid run dimensions");
    if (d->cycle_scans != 1)
        return fail(ADH_ERR_UNSUPPORTED,
                    "cycle with a scan axis (ion mobility) is not an AlphaRaw run");
    if (d->n_mobility < 1) return fail(ADH_ERR_INVALID_ARGUMENT, "mobility_values is empty");
    HIP_TRY(hipSetDevice(h->device));
    h->run_buf.release();
    h->run_staged = false;
    h->cands_uploaded = false;

    // validate the CSR on the host: kernels index with it unchecked
    float mz_lo = 0.f, mz_hi = 0.f;
    bool any = false;
    for (int64_t s = 0; s < d->n_spectra; ++s) {
        int64_t a = d->peak_start_idx[s], b = d->peak_stop_idx[s];
        if (a < 0 || b < a || b > d->n_peaks)
            return fail(ADH_ERR_INVALID_ARGUMENT, "peak_start/stop_idx out of range");
        if (b - a > (int64_t)0xFFFFFFFFll)
            return fail(ADH_ERR_UNSUPPORTED, "more than 2^32 peaks in one spectrum");
        if (b > a) {
            float lo = d->mz_values[a], hi = d->mz_values[b - 1];
            if (!any || lo < mz_lo) mz_lo = lo;
            if (!any || hi > mz_hi) mz_hi = hi;
            any = true;
        }
    }
    std::vector<int32_t> ms1;
    for (int r = 0; r < d->cycle_len * d->cycle_scans; ++r)
        if (-1.0 <= d->cycle[2 * r + 1] && -1.0 >= d->cycle[2 * r]) ms1.push_back(r);

    DevRun r{};
    r.n_spectra = d->n_spectra;
    r.n_peaks = d->n_peaks;
    r.cycle_len = d->cycle_len;
    r.cycle_scans = d->cycle_scans;
    r.n_ms1_obs = (int32_t)ms1.size();
    UP(h->run_buf, d->mz_values, d->n_peaks, &r.mz);
    UP(h->run_buf, d->intensity_values, d->n_peaks, &r.intensity);
    UP(h->run_buf, d->peak_start_idx, d->n_spectra, &r.pstart);
    UP(h->run_buf, d->peak_stop_idx, d->n_spectra, &r.pstop);
    UP(h->run_buf, d->rt_values, d->n_spectra, &r.rt);
    UP(h->run_buf, d->mobility_values, d->n_mobility, &r.mobility);
    UP(h->run_buf, d->cycle, (int64_t)d->cycle_len * d->cycle_scans * 2, &r.cycle);
    UP(h->run_buf, ms1.data(), (int64_t)ms1.size(), &r.ms1_obs);

    // m/z bucket index: about one bucket per peak of an average spectrum
    int64_t avg = d->n_spectra > 0 ? d->n_peaks / d->n_spectra : 0;
    int nb = (int)std::min<int64_t>(std::max<int64_t>(avg, 64), 4096);
    float span = mz_hi - mz_lo;
    if (!(span > 0.f)) span = 1.0f;
    r.n_buckets = nb;
    r.bucket_min = mz_lo;
    r.bucket_inv_width = (float)nb / span;
    uint32_t *bucket = nullptr;
    size_t bbytes = (size_t)std::max<int64_t>(d->n_spectra, 1) * (size_t)(nb + 1) * sizeof(uint32_t);
    HIP_TRY(hipMalloc((void **)&bucket, bbytes));
    h->run_buf.ptrs.push_back(bucket);
    r.bucket = bucket;
    if (d->n_spectra > 0) {
        hipLaunchKernelGGL(adh_bucket_build_kernel, dim3((unsigned)d->n_spectra), dim3(256), 0,
                           h->stream, r.mz, r.pstart, r.pstop, r.n_spectra, bucket, nb,
                           r.bucket_min, r.bucket_inv_width);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    h->run = r;
    h->run_staged = true;
    return ADH_OK;
}

int adh_stage_fragments(adh_handle_t *h, const adh_fragments_t *f) {
    if (!h || !f) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (f->n < 0) return fail(ADH_ERR_INVALID_ARGUMENT, "negative fragment count");
    HIP_TRY(hipSetDevice(h->device));
    h->lib_buf.release();
    h->lib_staged = false;
    DevLib l{};
    l.n = f->n;
    UP(h->lib_buf, f->mz_library, f->n, &l.mz_library);
    UP(h->lib_buf, f->mz, f->n, &l.mz);
    UP(h->lib_buf, f->intensity, f->n, &l.intensity);
    UP(h->lib_buf, f->type, f->n, &l.type);
    UP(h->lib_buf, f->loss_type, f->n, &l.loss_type);
    UP(h->lib_buf, f->charge, f->n, &l.charge);
    UP(h->lib_buf, f->number, f->n, &l.number);
    UP(h->lib_buf, f->position, f->n, &l.position);
    UP(h->lib_buf, f->cardinality, f->n, &l.cardinality);
    h->lib = l;
    h->lib_staged = true;
    return ADH_OK;
}

int adh_upload_candidates(adh_handle_t *h, const adh_candidates_t *c) {
    if (!h || !c) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!h->run_staged || !h->lib_staged)
        return fail(ADH_ERR_NOT_STAGED, "stage the run and the fragment library first");
    if (c->n < 0 || c->n_isotope_cols < 1)
        return fail(ADH_ERR_INVALID_ARGUMENT, "invalid candidate table dimensions");
    if (c->n > 0x7FFFFFFFll)
        return fail(ADH_ERR_UNSUPPORTED, "more than 2^31 candidates in one batch");
    HIP_TRY(hipSetDevice(h->device));
    h->cand_buf.release();
    h->cands_uploaded = false;

    // bounds the kernels rely on
    const int64_t L = h->run.cycle_len;
    for (int64_t i = 0; i < c->n; ++i) {
        if (c->flags && (c->flags[i] & ADH_FLAG_SKIP)) continue;
        if (c->frag_stop_idx[i] < c->frag_start_idx[i] || (int64_t)c->frag_stop_idx[i] > h->lib.n)
            return fail(ADH_ERR_INVALID_ARGUMENT, "fragment slice outside the staged library");
        int64_t fs = c->frame_start[i], fe = c->frame_stop[i], fc = c->frame_center[i];
        if (fs < 0 || fe < fs || fe > h->run.n_spectra || fc < 0 || fc >= h->run.n_spectra)
            return fail(ADH_ERR_INVALID_ARGUMENT, "frame limits outside the staged run");
        if ((fe - fs) % L != 0)
            return fail(ADH_ERR_INVALID_ARGUMENT,
                        "frame_stop - frame_start must be a multiple of the cycle length");
        if ((fe / L) * L > h->run.n_spectra)
            return fail(ADH_ERR_INVALID_ARGUMENT, "frame limits outside the staged run");
        int64_t ss = c->scan_start[i], se = c->scan_stop[i], sc = c->scan_center[i];
        if (se - ss != 1 || ss != 0 || sc != 0)
            return fail(ADH_ERR_UNSUPPORTED,
                        "AlphaRaw candidates must have scan_start=0, scan_stop=1, scan_center=0");
        if (c->charge[i] == 0) return fail(ADH_ERR_INVALID_ARGUMENT, "precursor charge is 0");
    }

    DevCands d{};
    d.n = c->n;
    d.n_isotope_cols = c->n_isotope_cols;
    UP(h->cand_buf, c->precursor_idx, c->n, &d.precursor_idx);
    UP(h->cand_buf, c->rank, c->n, &d.rank);
    if (c->flags)
        UP(h->cand_buf, c->flags, c->n, &d.flags);
    else
        d.flags = nullptr;
    UP(h->cand_buf, c->frag_start_idx, c->n, &d.frag_start);
    UP(h->cand_buf, c->frag_stop_idx, c->n, &d.frag_stop);
    UP(h->cand_buf, c->scan_start, c->n, &d.scan_start);
    UP(h->cand_buf, c->scan_stop, c->n, &d.scan_stop);
    UP(h->cand_buf, c->scan_center, c->n, &d.scan_center);
    UP(h->cand_buf, c->frame_start, c->n, &d.frame_start);
    UP(h->cand_buf, c->frame_stop, c->n, &d.frame_stop);
    UP(h->cand_buf, c->frame_center, c->n, &d.frame_center);
    UP(h->cand_buf, c->charge, c->n, &d.charge);
    UP(h->cand_buf, c->precursor_mz, c->n, &d.precursor_mz);
    UP(h->cand_buf, c->isotope_intensity, c->n * c->n_isotope_cols, &d.isotope_intensity);
    // processing order: by first cycle, so that concurrently resident wavefronts gather
    // from the same few spectra (L2 / Infinity-Cache reuse); output rows are unaffected
    {
        const int64_t n_cyc = h->run.n_spectra / L + 2;
        std::vector<uint32_t> head((size_t)n_cyc + 1, 0), order((size_t)c->n);
        for (int64_t i = 0; i < c->n; ++i) ++head[(size_t)(c->frame_start[i] / L) + 1];
        for (int64_t k = 0; k < n_cyc; ++k) head[(size_t)k + 1] += head[(size_t)k];
        for (int64_t i = 0; i < c->n; ++i) order[head[(size_t)(c->frame_start[i] / L)]++] = (uint32_t)i;
        if (getenv("ADH_DEBUG_NO_ORDER"))
            d.order = nullptr;
        else
            UP(h->cand_buf, order.data(), c->n, &d.order);
    }

    // LDS capacities of this batch (upper bounds: all isotope columns)
    HIP_TRY(hipMemsetAsync(h->d_maxima, 0, 4 * sizeof(int32_t), h->stream));
    if (d.n > 0) {
        unsigned blocks = (unsigned)((d.n + 255) / 256);
        hipLaunchKernelGGL(adh_plan_kernel, dim3(blocks), dim3(256), 0, h->stream, h->run, d,
                           (uint32_t)d.n_isotope_cols, h->d_maxima);
        HIP_TRY(hipGetLastError());
    }
    int32_t mx[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(mx, h->d_maxima, sizeof(mx), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->plan_n_lib = mx[0];
    h->plan_o = mx[1];
    h->plan_f = mx[2];
    h->cands = d;
    h->cands_uploaded = true;
    return ADH_OK;
}

int adh_score_uploaded(adh_handle_t *h, const adh_scoring_config_t *cfg, adh_output_t *out,
                       void *hip_stream) {
    if (!h || !cfg || !out) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!h->cands_uploaded) return fail(ADH_ERR_NOT_STAGED, "no candidate table uploaded");
    if (out->n != h->cands.n) return fail(ADH_ERR_INVALID_ARGUMENT, "output rows != candidates");
    if (cfg->top_k_fragments == 0 || cfg->top_k_isotopes == 0)
        return fail(ADH_ERR_INVALID_ARGUMENT, "top_k_fragments / top_k_isotopes must be > 0");
    if (out->top_k <= 0) return fail(ADH_ERR_INVALID_ARGUMENT, "output top_k must be > 0");
    HIP_TRY(hipSetDevice(h->device));
    if (h->cands.n == 0) return ADH_OK;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->stream;

    Caps caps;
    caps.n_lib = std::max(h->plan_n_lib, 1);
    caps.k = (int32_t)std::max<int64_t>(
        std::min<int64_t>((int64_t)cfg->top_k_fragments, (int64_t)caps.n_lib), 1);
    caps.o = std::max(h->plan_o, 1);
    caps.f = std::max(h->plan_f, 1);
    caps.i = std::max<int32_t>(
        (int32_t)std::min<uint32_t>(cfg->top_k_isotopes, (uint32_t)h->cands.n_isotope_cols), 1);
    {
        const char *dbg = getenv("ADH_DEBUG_STOP_PHASE");  // developer ablation switch
        caps.stop_phase = dbg ? atoi(dbg) : 0;
    }
    if (cfg->collect_fragments && caps.k > out->top_k)
        return fail(ADH_ERR_INVALID_ARGUMENT, "output top_k smaller than config.top_k_fragments");
    size_t lds = adh_score_lds_bytes(caps);
    if (lds > 160 * 1024) {
        char buf[256];
        snprintf(buf, sizeof(buf),
                 "candidate tile needs %zu bytes of LDS (K=%d O=%d F=%d): exceeds 160 KiB", lds,
                 caps.k, caps.o, caps.f);
        return fail(ADH_ERR_UNSUPPORTED, buf);
    }
    hipEvent_t e0, e1;
    int rc = get_event(h, &e0);
    if (rc != ADH_OK) return rc;
    rc = get_event(h, &e1);
    if (rc != ADH_OK) return rc;
    HIP_TRY(hipEventRecord(e0, st));
    hipLaunchKernelGGL(adh_score_kernel, dim3((unsigned)h->cands.n), dim3(ADH_WAVE), lds, st,
                       h->run, h->lib, h->cands, *cfg, *out, caps);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, st));
    h->timed.emplace_back(e0, e1);
    return ADH_OK;
}

int adh_synchronize(adh_handle_t *h) {
    if (!h) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ADH_OK;
}

int adh_kernel_time_ms(adh_handle_t *h, double *avg_ms, int64_t *launches, int reset) {
    if (!h || !avg_ms || !launches) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    double sum = 0;
    int64_t n = 0;
    for (auto &p : h->timed) {
        HIP_TRY(hipEventSynchronize(p.second));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        sum += ms;
        ++n;
    }
    *avg_ms = n ? sum / (double)n : 0.0;
    *launches = n;
    if (reset) {
        for (auto &p : h->timed) {
            h->free_events.push_back(p.first);
            h->free_events.push_back(p.second);
        }
        h->timed.clear();
    }
    return ADH_OK;
}

int adh_score_candidates(adh_handle_t *h, const adh_candidates_t *c, const adh_scoring_config_t *cfg,
                         adh_output_t *out) {
    if (!h || !c || !cfg || !out) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (out->n != c->n) return fail(ADH_ERR_INVALID_ARGUMENT, "output rows != candidates");
    if (out->top_k <= 0) return fail(ADH_ERR_INVALID_ARGUMENT, "output top_k must be > 0");
    int rc = adh_upload_candidates(h, c);
    if (rc != ADH_OK) return rc;
    const int64_t n = c->n;
    const size_t tk = (size_t)out->top_k;
    struct Field {
        void **host;
        size_t bytes;
    };
    adh_output_t dev = *out;
    Field fields[] = {
        {(void **)&out->valid, (size_t)n},
        {(void **)&out->precursor_idx, (size_t)n * 4},
        {(void **)&out->rank, (size_t)n},
        {(void **)&out->features, (size_t)n * ADH_NUM_FEATURES * 4},
        {(void **)&out->fragment_precursor_idx, (size_t)n * tk * 4},
        {(void **)&out->fragment_rank, (size_t)n * tk},
        {(void **)&out->fragment_mz_library, (size_t)n * tk * 4},
        {(void **)&out->fragment_mz, (size_t)n * tk * 4},
        {(void **)&out->fragment_mz_observed, (size_t)n * tk * 4},
        {(void **)&out->fragment_height, (size_t)n * tk * 4},
        {(void **)&out->fragment_intensity, (size_t)n * tk * 4},
        {(void **)&out->fragment_mass_error, (size_t)n * tk * 4},
        {(void **)&out->fragment_correlation, (size_t)n * tk * 4},
        {(void **)&out->fragment_position, (size_t)n * tk},
        {(void **)&out->fragment_number, (size_t)n * tk},
        {(void **)&out->fragment_type, (size_t)n * tk},
        {(void **)&out->fragment_charge, (size_t)n * tk},
        {(void **)&out->fragment_loss_type, (size_t)n * tk},
        {(void **)&out->stat_matched_peaks, (size_t)n * 4},
    };
    void **dev_slots[] = {
        (void **)&dev.valid, (void **)&dev.precursor_idx, (void **)&dev.rank, (void **)&dev.features,
        (void **)&dev.fragment_precursor_idx, (void **)&dev.fragment_rank,
        (void **)&dev.fragment_mz_library, (void **)&dev.fragment_mz,
        (void **)&dev.fragment_mz_observed, (void **)&dev.fragment_height,
        (void **)&dev.fragment_intensity, (void **)&dev.fragment_mass_error,
        (void **)&dev.fragment_correlation, (void **)&dev.fragment_position,
        (void **)&dev.fragment_number, (void **)&dev.fragment_type, (void **)&dev.fragment_charge,
        (void **)&dev.fragment_loss_type, (void **)&dev.stat_matched_peaks};
    const int NF = (int)(sizeof(fields) / sizeof(fields[0]));
    DeviceBuffers tmp;
    rc = ADH_OK;
    for (int i = 0; i < NF && rc == ADH_OK; ++i) {
        if (*fields[i].host == nullptr) {
            if (i == NF - 1) {  // stats are optional
                *dev_slots[i] = nullptr;
                continue;
            }
            rc = fail(ADH_ERR_INVALID_ARGUMENT, "output buffer is NULL");
            break;
        }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(fields[i].bytes, 1));
        if (e != hipSuccess) {
            rc = fail(ADH_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
            break;
        }
        tmp.ptrs.push_back(p);
        *dev_slots[i] = p;
        e = hipMemsetAsync(p, 0, std::max<size_t>(fields[i].bytes, 1), h->stream);
        if (e != hipSuccess) rc = fail(ADH_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e));
    }
    if (rc == ADH_OK) rc = adh_score_uploaded(h, cfg, &dev, nullptr);
    if (rc == ADH_OK) {
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess)
            rc = fail(ADH_ERR_HIP, std::string("scoring kernel: ") + hipGetErrorString(e));
    }
    for (int i = 0; i < NF && rc == ADH_OK; ++i) {
        if (*dev_slots[i] == nullptr || fields[i].bytes == 0) continue;
        hipError_t e = hipMemcpy(*fields[i].host, *dev_slots[i], fields[i].bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ADH_ERR_HIP, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
    }
    tmp.release();
    return rc;
}

int adh_fragcomp(adh_handle_t *h, int64_t n_windows, const int64_t *window_start,
                 const int64_t *window_stop, int64_t n_psm, const float *rt,
                 const int64_t *frag_start_idx, const int64_t *frag_stop_idx, int64_t n_frag,
                 const float *fragment_mz, double rt_tol_seconds, double mass_tol_ppm,
                 uint8_t *valid) {
    if (!h || (n_windows > 0 && (!window_start || !window_stop)) ||
        (n_psm > 0 && (!rt || !frag_start_idx || !frag_stop_idx || !valid)) ||
        (n_frag > 0 && !fragment_mz))
        return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_windows < 0 || n_psm < 0 || n_frag < 0)
        return fail(ADH_ERR_INVALID_ARGUMENT, "negative size");
    for (int64_t w = 0; w < n_windows; ++w)
        if (window_start[w] < 0 || window_stop[w] < window_start[w] || window_stop[w] > n_psm)
            return fail(ADH_ERR_INVALID_ARGUMENT, "window range outside the PSM table");
    for (int64_t i = 0; i < n_psm; ++i)
        if (frag_start_idx[i] < 0 || frag_stop_idx[i] < frag_start_idx[i] || frag_stop_idx[i] > n_frag)
            return fail(ADH_ERR_INVALID_ARGUMENT, "fragment range outside the fragment table");
    if (n_windows == 0 || n_psm == 0) return ADH_OK;
    HIP_TRY(hipSetDevice(h->device));
    DeviceBuffers tmp;
    const int64_t *d_ws, *d_we, *d_fs, *d_fe;
    const float *d_rt, *d_mz;
    const uint8_t *d_valid_c;
    int rc;
#define FC_UP(host, n, dev)                                   \
    rc = upload(tmp, host, n, dev, h->stream);                \
    if (rc != ADH_OK) {                                       \
        tmp.release();                                        \
        return rc;                                            \
    }
    FC_UP(window_start, n_windows, &d_ws);
    FC_UP(window_stop, n_windows, &d_we);
    FC_UP(rt, n_psm, &d_rt);
    FC_UP(frag_start_idx, n_psm, &d_fs);
    FC_UP(frag_stop_idx, n_psm, &d_fe);
    FC_UP(fragment_mz, n_frag, &d_mz);
    FC_UP(valid, n_psm, &d_valid_c);
#undef FC_UP
    uint8_t *d_valid = const_cast<uint8_t *>(d_valid_c);
    hipLaunchKernelGGL(adh_fragcomp_kernel, dim3((unsigned)n_windows), dim3(ADH_FC_THREADS), 0,
                       h->stream, n_windows, d_ws, d_we, d_rt, d_fs, d_fe, d_mz, rt_tol_seconds,
                       mass_tol_ppm, d_valid);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(valid, d_valid, (size_t)n_psm, hipMemcpyDeviceToHost);
    tmp.release();
    if (e != hipSuccess) return fail(ADH_ERR_HIP, std::string("fragcomp: ") + hipGetErrorString(e));
    return ADH_OK;
}

}  // extern "C"
