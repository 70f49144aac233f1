// adh_score.hip - candidate scoring on gfx950: one 64-lane wavefront per candidate.
//
// The kernel does the work of the reference's per-candidate routine
//   Candidate.process            alphadia/search/scoring/containers/candidate.py:166-481
// including
//   AlphaRawJIT.get_dense        alphadia/search/jitclasses/alpharaw_jit.py:208-337
//   quadrupole transfer/template alphadia/search/scoring/quadrupole.py:261-335
//   location / precursor / fragment / profile features
//                                alphadia/search/scoring/features/*.py
// but is organised for a CDNA4 wavefront rather than for a CPU thread:
//
//   * the XIC tile of a candidate (fragments x observations x cycles, two
//     channels) lives in LDS; the duplicated "scan" axis of non-IM data is never
//     materialised, its effect on every sum is applied arithmetically
//   * the reference walks the K m/z windows of a spectrum with one monotone
//     cursor and a binary search per window; here every (fragment, observation,
//     cycle) cell is an independent lane: the cursor rule collapses to
//     "m/z > max(hi of earlier windows)", and the window start comes from a
//     per-spectrum m/z bucket table in HBM (one 4-byte load + a short scan)
//   * exp() weights of weighted_center_mean depend only on (observation, scan,
//     cycle), so they are computed once per candidate into an LDS table instead
//     of once per non-zero cell per fragment
//   * reductions that the reference performs sequentially in float32 keep their
//     order (one lane walks the short axis) so results agree with the CPU
//     restatement to the last bit wherever libm agrees
//
// Arithmetic follows Numba's typing of the reference source (float32 vs float64
// per expression); compile with -ffp-contract=off.
#include "adh_device.h"

namespace {

constexpr double ISOTOPE_DELTA = 1.0033548350700006;  // candidate.py:160

__device__ __forceinline__ int lane_id() { return threadIdx.x; }

// python slice(start, stop) on length n
__device__ __forceinline__ void py_slice(int start, int stop, int n, int &a, int &b) {
    if (start < 0) start += n;
    if (stop < 0) stop += n;
    a = min(max(start, 0), n);
    b = min(max(stop, 0), n);
    if (b < a) b = a;
}

struct Lds {
    float *fi, *fm, *ffp;      // [K][O][F] fragment intensity / m/z channel / frame profile
    float *pi, *pm;            // [I][F] collapsed precursor channels
    float *tpl, *tfp;          // [O][F] template (one scan) / enveloped frame profile
    float *bp;                 // [K][F] best profile
    double *wt;                // [O][2][F] exp weights around the template centre
    double *wtp;               // [2][F] exp weights around the precursor "centre"
    double *qtf;               // [I][O]
    float *l_int, *l_mz;       // library slice scratch [NL]
    int *l_rank;
    float *f_mzlib, *f_mz, *f_int, *f_lo, *f_hi, *f_excl;  // pre-mask fragments [K]
    float *g_mzlib, *g_mz, *g_int, *g_fin;                 // post-mask fragments [K]
    int *f_src, *kmap, *ord;
    double *omz, *ohe;         // [K][O]
    double *mzmean, *height, *area, *merr;  // [K]
    float *obs_int, *corr, *rowsum, *fw, *ftc;
    int *fpeak;
    int *obs;                  // [O]
    float *oi, *tsum, *qmask, *medpk;
    double *esc, *efc;
    float *iso_mz, *iso_int, *i_lo, *i_hi, *i_excl, *spi;  // [I]
    double *hp, *omzp;
    float *frame_rt, *med, *xm;  // [F]
    float *feat;               // [46]
    uint8_t *f_type, *f_loss, *f_charge, *f_number, *f_pos;
    uint8_t *g_type, *g_loss, *g_charge, *g_number, *g_pos;
    uint8_t *l_ok;
};

template <typename T>
__host__ __device__ __forceinline__ T *carve(unsigned char *&p, size_t n) {
    T *r = reinterpret_cast<T *>(p);
    p += ((n * sizeof(T) + 7) / 8) * 8;
    return r;
}

__host__ __device__ inline size_t lds_layout(unsigned char *base, const Caps &c, Lds *l) {
    unsigned char *p = base;
    const size_t K = c.k, O = c.o, F = c.f, I = c.i, NL = c.n_lib;
    Lds t;
    t.wt = carve<double>(p, O * 2 * F);
    t.wtp = carve<double>(p, 2 * F);
    t.qtf = carve<double>(p, I * O);
    t.omz = carve<double>(p, K * O);
    t.ohe = carve<double>(p, K * O);
    t.mzmean = carve<double>(p, K);
    t.height = carve<double>(p, K);
    t.area = carve<double>(p, K);
    t.merr = carve<double>(p, K);
    t.esc = carve<double>(p, O);
    t.efc = carve<double>(p, O);
    t.hp = carve<double>(p, I);
    t.omzp = carve<double>(p, I);
    t.fi = carve<float>(p, K * O * F);
    t.fm = carve<float>(p, K * O * F);
    t.ffp = carve<float>(p, K * O * F);
    t.pi = carve<float>(p, I * F);
    t.pm = carve<float>(p, I * F);
    t.tpl = carve<float>(p, O * F);
    t.tfp = carve<float>(p, O * F);
    t.bp = carve<float>(p, K * F);
    t.l_int = carve<float>(p, NL);
    t.l_mz = carve<float>(p, NL);
    t.l_rank = carve<int>(p, NL);
    t.f_mzlib = carve<float>(p, K);
    t.f_mz = carve<float>(p, K);
    t.f_int = carve<float>(p, K);
    t.f_lo = carve<float>(p, K);
    t.f_hi = carve<float>(p, K);
    t.f_excl = carve<float>(p, K);
    t.g_mzlib = carve<float>(p, K);
    t.g_mz = carve<float>(p, K);
    t.g_int = carve<float>(p, K);
    t.g_fin = carve<float>(p, K);
    t.f_src = carve<int>(p, K);
    t.kmap = carve<int>(p, K);
    t.ord = carve<int>(p, K);
    t.obs_int = carve<float>(p, K);
    t.corr = carve<float>(p, K);
    t.rowsum = carve<float>(p, K * O);
    t.fw = carve<float>(p, K * O);
    t.ftc = carve<float>(p, K * O);
    t.fpeak = carve<int>(p, K * O);
    t.obs = carve<int>(p, O);
    t.oi = carve<float>(p, O);
    t.tsum = carve<float>(p, O);
    t.qmask = carve<float>(p, O);
    t.medpk = carve<float>(p, O);
    t.iso_mz = carve<float>(p, I);
    t.iso_int = carve<float>(p, I);
    t.i_lo = carve<float>(p, I);
    t.i_hi = carve<float>(p, I);
    t.i_excl = carve<float>(p, I);
    t.spi = carve<float>(p, I);
    t.frame_rt = carve<float>(p, F);
    t.med = carve<float>(p, F);
    t.xm = carve<float>(p, F);
    t.feat = carve<float>(p, ADH_NUM_FEATURES);
    t.f_type = carve<uint8_t>(p, K);
    t.f_loss = carve<uint8_t>(p, K);
    t.f_charge = carve<uint8_t>(p, K);
    t.f_number = carve<uint8_t>(p, K);
    t.f_pos = carve<uint8_t>(p, K);
    t.g_type = carve<uint8_t>(p, K);
    t.g_loss = carve<uint8_t>(p, K);
    t.g_charge = carve<uint8_t>(p, K);
    t.g_number = carve<uint8_t>(p, K);
    t.g_pos = carve<uint8_t>(p, K);
    t.l_ok = carve<uint8_t>(p, NL);
    if (l) *l = t;
    return (size_t)(p - base);
}

// One window of one spectrum: the reference's inner loop (alpharaw_jit.py:292-335)
// with the monotone cursor expressed as the exclusion bound `excl`.
__device__ __forceinline__ void gather_window(const DevRun &run, int64_t spec, float lo, float hi,
                                              float excl, float &acc_i, float &acc_m,
                                              uint32_t &hits) {
    const int64_t ps = run.pstart[spec];
    const int64_t pe = run.pstop[spec];
    const int b = adh_bucket_of(lo, run.bucket_min, run.bucket_inv_width, run.n_buckets);
    int64_t idx = ps + (int64_t)run.bucket[spec * (int64_t)(run.n_buckets + 1) + b];
    while (idx < pe) {
        float m = run.mz[idx];
        if (m >= lo && m > excl) break;
        ++idx;
    }
    while (idx < pe) {
        float m = run.mz[idx];
        if (!(m <= hi)) break;
        float ni = run.intensity[idx];
        ni = ((double)ni > 1e-26) ? ni : ni * 0.0f;
        float a = acc_m * acc_i;
        float bb = ni * m;
        float n32 = a + bb;
        float d32 = acc_i + ni;
        acc_m = (float)(((double)n32 + 1e-36) / ((double)d32 + 1e-36));
        acc_i = d32;
        ++hits;
        ++idx;
    }
}

__device__ __forceinline__ double logistic(double x, double mu, double sigma) {
    double a = (x - mu) / sigma;
    return 1.0 / (1.0 + exp(-a));
}

// np.corrcoef(x, y)[0, 1] in float64 (sequential sums)
__device__ double corrcoef01(const double *x, const float *y, int n) {
    double sx = 0, sy = 0;
    for (int i = 0; i < n; ++i) sx += x[i];
    for (int i = 0; i < n; ++i) sy += (double)y[i];
    double mx = sx / (double)n, my = sy / (double)n;
    double cxx = 0, cyy = 0, cxy = 0;
    for (int i = 0; i < n; ++i) {
        double a = x[i] - mx, b = (double)y[i] - my;
        cxx += a * a;
        cyy += b * b;
        cxy += a * b;
    }
    double fact = fmax((double)n - 1.0, 0.0);
    double inv = 1.0 / fact;
    cxx *= inv;
    cyy *= inv;
    cxy *= inv;
    double s0 = sqrt(cxx), s1 = sqrt(cyy);
    double c = cxy / s1 / s0;
    if (fabs(c) > 1.0) c = (c > 0) ? 1.0 : -1.0;
    return c;
}

}  // namespace

// ------------------------------------------------------------------ plan kernel
// maxima of (library slice length, observations, cycles) over the batch -> LDS capacities
__global__ void adh_plan_kernel(DevRun run, DevCands cd, uint32_t top_k_isotopes, int32_t *maxima) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cd.n) return;
    if (cd.flags && (cd.flags[i] & ADH_FLAG_SKIP)) return;
    int n_lib = (int)cd.frag_stop[i] - (int)cd.frag_start[i];
    if (n_lib < 0) n_lib = 0;
    int64_t L = run.cycle_len;
    int F = (int)(cd.frame_stop[i] / L - cd.frame_start[i] / L);
    if (F < 0) F = 0;
    int I = min((int)cd.n_isotope_cols, (int)top_k_isotopes);
    float pmz = cd.precursor_mz[i];
    double ch = (double)cd.charge[i];
    float mn = 0, mx = 0;
    for (int k = 0; k < I; ++k) {
        float m = (float)((double)k * ISOTOPE_DELTA / ch) + pmz;
        if (k == 0 || m < mn) mn = m;
        if (k == 0 || m > mx) mx = m;
    }
    float q_lo = (float)((double)mn - 0.5), q_hi = (float)((double)mx + 0.5);
    int O = 0;
    int rows = run.cycle_len * run.cycle_scans;
    for (int r = 0; r < rows; ++r)
        O += ((double)q_lo <= run.cycle[2 * r + 1]) && ((double)q_hi >= run.cycle[2 * r]);
    atomicMax(&maxima[0], n_lib);
    atomicMax(&maxima[1], O);
    atomicMax(&maxima[2], F);
}

size_t adh_score_lds_bytes(const Caps &c) { return lds_layout(nullptr, c, nullptr); }

// ------------------------------------------------------------------ main kernel
__global__ __launch_bounds__(ADH_WAVE) void adh_score_kernel(DevRun run, DevLib lib, DevCands cd,
                                                             adh_scoring_config_t cfg, DevOut out,
                                                             Caps caps) {
    extern __shared__ __align__(16) unsigned char smem[];
    Lds s;
    lds_layout(smem, caps, &s);
    const int lane = lane_id();
    const int64_t row = cd.order ? (int64_t)cd.order[blockIdx.x] : (int64_t)blockIdx.x;
    if (cd.flags && (cd.flags[row] & ADH_FLAG_SKIP)) return;

    const uint32_t precursor_idx = cd.precursor_idx[row];
    const uint8_t rank = cd.rank[row];
    if (lane == 0) {
        out.precursor_idx[row] = precursor_idx;
        out.rank[row] = rank;
    }
    const int64_t frag_start = cd.frag_start[row];
    const int n_lib = max((int)((int64_t)cd.frag_stop[row] - frag_start), 0);
    const int64_t scan_start = cd.scan_start[row], scan_stop = cd.scan_stop[row];
    const int64_t scan_center = cd.scan_center[row];
    const int64_t frame_start = cd.frame_start[row], frame_stop = cd.frame_stop[row];
    const int64_t frame_center = cd.frame_center[row];
    const int64_t L = run.cycle_len;
    const int top_k = out.top_k;

    // ---- fragments: slice, cardinality filter, top-k by intensity, sort by m/z
    //      (candidate.py:181-188, fragment_container.py:56-102)
    for (int j = lane; j < n_lib; j += ADH_WAVE) {
        s.l_int[j] = lib.intensity[frag_start + j];
        s.l_mz[j] = lib.mz[frag_start + j];
        s.l_ok[j] = !(cfg.exclude_shared_ions && lib.cardinality[frag_start + j] > 1);
    }
    __syncthreads();
    for (int a = lane; a < n_lib; a += ADH_WAVE) {
        int r = -1;
        if (s.l_ok[a]) {
            r = 0;
            float ia = s.l_int[a];
            for (int b = 0; b < n_lib; ++b) {
                if (!s.l_ok[b]) continue;
                float ib = s.l_int[b];
                r += (ib > ia) || (ib == ia && b > a);  // argsort()[::-1]
            }
            if (r >= (int)cfg.top_k_fragments) r = -1;
        }
        s.l_rank[a] = r;
    }
    __syncthreads();
    int K = 0;
    for (int a = 0; a < n_lib; ++a) K += s.l_rank[a] >= 0;  // uniform
    for (int a = lane; a < n_lib; a += ADH_WAVE) {
        int ra = s.l_rank[a];
        if (ra < 0) continue;
        float ma = s.l_mz[a];
        int slot = 0;
        for (int b = 0; b < n_lib; ++b) {
            int rb = s.l_rank[b];
            if (rb < 0) continue;
            float mb = s.l_mz[b];
            slot += (mb < ma) || (mb == ma && rb < ra);
        }
        int64_t g = frag_start + a;
        s.f_src[slot] = a;
        s.f_mzlib[slot] = lib.mz_library[g];
        s.f_mz[slot] = ma;
        s.f_int[slot] = s.l_int[a];
        s.f_type[slot] = lib.type[g];
        s.f_loss[slot] = lib.loss_type[g];
        s.f_charge[slot] = lib.charge[g];
        s.f_number[slot] = lib.number[g];
        s.f_pos[slot] = lib.position[g];
    }
    if (K <= 3) return;  // candidate.py:190
    if (caps.stop_phase == 1) return;

    // ---- isotopes (candidate.py:151-163) and quadrupole limits (candidate.py:203-205)
    const int I = min((int)cd.n_isotope_cols, (int)cfg.top_k_isotopes);
    if (lane < I) {
        s.iso_int[lane] = cd.isotope_intensity[row * cd.n_isotope_cols + lane];
        double off = (double)lane * ISOTOPE_DELTA / (double)cd.charge[row];
        s.iso_mz[lane] = (float)off + cd.precursor_mz[row];
    }
    __syncthreads();
    float iso_min = s.iso_mz[0], iso_max = s.iso_mz[0];
    for (int i = 1; i < I; ++i) {
        iso_min = fminf(iso_min, s.iso_mz[i]);
        iso_max = fmaxf(iso_max, s.iso_mz[i]);
    }
    const float q_lo = (float)((double)iso_min - 0.5), q_hi = (float)((double)iso_max + 0.5);

    // ---- observation list: _calculate_valid_scans (alpharaw_jit.py:19-50)
    int O = 0;
    {
        const int rows = run.cycle_len * run.cycle_scans;
        for (int base = 0; base < rows; base += ADH_WAVE) {
            int r = base + lane;
            bool hit = r < rows && ((double)q_lo <= run.cycle[2 * r + 1]) &&
                       ((double)q_hi >= run.cycle[2 * r]);
            unsigned long long m = __ballot(hit);
            if (hit) {
                int pos = O + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < caps.o) s.obs[pos] = r;
            }
            O += __popcll(m);
        }
    }
    const int64_t c0 = frame_start / L, c1 = frame_stop / L;
    const int F = (int)(c1 - c0);
    if (F <= 0) return;          // candidate.py:230
    if (O <= 0 || O > caps.o) return;  // no overlapping window: fails at candidate.py:323

    // ---- m/z windows: mass_range (jitclasses/utils.py:15-20), all float32
    if (lane < K) {
        float mzq = s.f_mz[lane];
        float t = cfg.fragment_mz_tolerance * mzq;
        float q = t / 1000000.0f;
        s.f_lo[lane] = mzq - q;
        s.f_hi[lane] = mzq + q;
    }
    for (int k = lane + ADH_WAVE; k < K; k += ADH_WAVE) {
        float mzq = s.f_mz[k];
        float t = cfg.fragment_mz_tolerance * mzq;
        float q = t / 1000000.0f;
        s.f_lo[k] = mzq - q;
        s.f_hi[k] = mzq + q;
    }
    if (lane < I) {
        float mzq = s.iso_mz[lane];
        float t = cfg.precursor_mz_tolerance * mzq;
        float q = t / 1000000.0f;
        s.i_lo[lane] = mzq - q;
        s.i_hi[lane] = mzq + q;
    }
    __syncthreads();
    if (lane == 0) {
        float e = -INFINITY;
        for (int k = 0; k < K; ++k) {
            s.f_excl[k] = e;
            e = fmaxf(e, s.f_hi[k]);
        }
        e = -INFINITY;
        for (int i = 0; i < I; ++i) {
            s.i_excl[i] = e;
            e = fmaxf(e, s.i_hi[i]);
        }
    }
    __syncthreads();

    // ---- XIC gather: AlphaRawJIT.get_dense (alpharaw_jit.py:208-337), absolute_masses=True
    uint32_t hits = 0;
    const int OF = O * F;
    for (int c = lane; c < K * OF; c += ADH_WAVE) {
        int k = c / OF, rem = c - k * OF;
        int o = rem / F, f = rem - o * F;
        int64_t spec = (int64_t)s.obs[o] + (c0 + f) * L;
        float ai = 0.0f, am = 0.0f;
        gather_window(run, spec, s.f_lo[k], s.f_hi[k], s.f_excl[k], ai, am, hits);
        s.fi[c] = ai;
        s.fm[c] = am;
    }
    // precursor isotopes in MS1, observations collapsed (candidate.py:239-269)
    for (int c = lane; c < I * F; c += ADH_WAVE) {
        int i = c / F, f = c - i * F;
        float acc = 0.0f;
        double sum = 0.0;
        int count = 0;
        for (int j = 0; j < run.n_ms1_obs; ++j) {
            int64_t spec = (int64_t)run.ms1_obs[j] + (c0 + f) * L;
            float ai = 0.0f, am = 0.0f;
            gather_window(run, spec, s.i_lo[i], s.i_hi[i], s.i_excl[i], ai, am, hits);
            acc += ai;
            sum += (double)am;
            count += am > 0.0f;
        }
        s.pi[c] = acc;
        s.pm[c] = (float)(sum / ((double)count + 1e-6));
    }
    if (out.stat_matched_peaks) {
        for (int off = 32; off > 0; off >>= 1) hits += __shfl_xor(hits, off);
        if (lane == 0) out.stat_matched_peaks[row] = hits;
    }

    if (caps.stop_phase == 2) return;
    // ---- quadrupole transfer function (quadrupole.py:261-301), n_scans == 1 (non-IM)
    for (int c = lane; c < I * O; c += ADH_WAVE) {
        int i = c / O, o = c - i * O;
        const double *cy = run.cycle + 2 * ((int64_t)s.obs[o] * run.cycle_scans + scan_start);
        double x = (double)s.iso_mz[i];
        s.qtf[c] = logistic(x, cy[0], 0.2) - logistic(x, cy[1], 0.2);
    }
    __syncthreads();
    if (lane < O) {
        double sum = 0;
        for (int i = 0; i < I; ++i) sum += s.qtf[i * O + lane];
        s.qmask[lane] = (float)(sum / (double)I);  // candidate.py:287-289
    }
    __syncthreads();
    for (int c = lane; c < K * OF; c += ADH_WAVE) {
        int o = (c % OF) / F;
        s.fi[c] = s.fi[c] * s.qmask[o];  // candidate.py:290
    }
    // template (quadrupole.py:304-324); both scan slots are identical for non-IM data
    for (int c = lane; c < OF; c += ADH_WAVE) {
        int o = c / F, f = c - o * F;
        double acc = 0;
        for (int i = 0; i < I; ++i) {
            float a = s.pi[i * F + f] * s.iso_int[i];
            acc += (double)a * s.qtf[i * O + o];
        }
        s.tpl[c] = (float)acc;
    }
    __syncthreads();

    // ---- observation importance (quadrupole.py:327-335)
    if (lane < O) {
        float sf = 0;
        for (int f = 0; f < F; ++f) sf += s.tpl[lane * F + f];
        s.tsum[lane] = sf + sf;  // sum over the two identical scan slots
    }
    // ---- fragment presence mask (candidate.py:319-329)
    for (int k = lane; k < K; k += ADH_WAVE) {
        float so = 0;
        for (int o = 0; o < O; ++o) {
            float sf = 0;
            for (int f = 0; f < F; ++f) sf += s.fi[(k * O + o) * F + f];
            float ss = sf + sf;
            s.rowsum[k * O + o] = ss;
            so += ss;
        }
        s.l_rank[k] = so > 0.0f;  // reuse as "present" flag (K <= n_lib)
    }
    __syncthreads();
    {
        float tot = 0;
        for (int o = 0; o < O; ++o) tot += s.tsum[o];
        if (lane < O) s.oi[lane] = (tot == 0.0f) ? 1.0f / (float)O : s.tsum[lane] / tot;
    }
    const int K0 = K;
    int n_present = 0;
    for (int k = 0; k < K0; ++k) {
        if (s.l_rank[k]) {
            if (lane == 0) s.kmap[n_present] = k;
            ++n_present;
        }
    }
    if (n_present < 2) return;  // candidate.py:323
    K = n_present;
    __syncthreads();
    for (int k = lane; k < K; k += ADH_WAVE) {
        int src = s.kmap[k];
        s.g_mzlib[k] = s.f_mzlib[src];
        s.g_mz[k] = s.f_mz[src];
        s.g_int[k] = s.f_int[src];
        s.g_type[k] = s.f_type[src];
        s.g_loss[k] = s.f_loss[src];
        s.g_charge[k] = s.f_charge[src];
        s.g_number[k] = s.f_number[src];
        s.g_pos[k] = s.f_pos[src];
    }
    __syncthreads();
    {
        // apply_mask renormalisation (fragment_container.py:119-120), then the
        // second normalisation inside fragment_features (fragment_features.py:218)
        float sum1 = 0;
        for (int k = 0; k < K; ++k) sum1 += s.g_int[k];
        __syncthreads();
        for (int k = lane; k < K; k += ADH_WAVE) s.g_int[k] = s.g_int[k] / sum1;
        __syncthreads();
        float sum2 = 0;
        for (int k = 0; k < K; ++k) sum2 += s.g_int[k];
        for (int k = lane; k < K; k += ADH_WAVE) s.g_fin[k] = s.g_int[k] / sum2;
    }

    // ---- profiles (candidate.py:333-347; scoring/utils.py:26-66)
    for (int c = lane; c < K * OF; c += ADH_WAVE) {
        int k = c / OF, rem = c - k * OF;
        float v = s.fi[s.kmap[k] * OF + rem];
        s.ffp[c] = v + v;
    }
    for (int c = lane; c < OF; c += ADH_WAVE) {
        int f = c % F;
        float x = s.tpl[c] + s.tpl[c];
        float r = x;
        if (f >= 1 && f < F - 1) {
            float xl = s.tpl[c - 1] + s.tpl[c - 1];
            float xr = s.tpl[c + 1] + s.tpl[c + 1];
            if (x < xl || x < xr) {
                float sm = xl + xr;
                r = (float)((double)sm / 2.0);
            }
        }
        s.tfp[c] = r;
    }
    const int n_frame_rt = (int)((frame_stop - frame_start + L - 1) / L);
    for (int f = lane; f < min(n_frame_rt, caps.f); f += ADH_WAVE)
        s.frame_rt[f] = run.rt[frame_start + (int64_t)f * L];
    if (lane < ADH_NUM_FEATURES) s.feat[lane] = 0.0f;
    __syncthreads();

    if (caps.stop_phase == 3) return;
    // =========================== features ===========================
    // ---- precursor weight table around (scan, frame) = (S, 1) = (2, 1)
    //      (precursor_features.py:52-57, features_utils.py:9-25)
    for (int c = lane; c < 2 * F; c += ADH_WAVE) {
        int sc = c / F, f = c - sc * F;
        double ds = (double)(sc - 2), df = (double)(f - 1);
        double dist = sqrt(ds * ds + df * df);
        s.wtp[c] = exp(-0.1 * dist);
    }
    // ---- template centre of mass per observation (fragment_features.py:20-68)
    if (lane < O) {
        double isum = 0, ssum = 0, fsum = 0;
        bool any = false;
        for (int sc = 0; sc < 2; ++sc)
            for (int f = 0; f < F; ++f) {
                float v = s.tpl[lane * F + f];
                if (v > 0.0f) {
                    any = true;
                    isum += (double)v;
                    ssum += (double)sc * (double)v;
                    fsum += (double)f * (double)v;
                }
            }
        s.esc[lane] = (any && isum > 0) ? ssum / isum : 0.0;
        s.efc[lane] = (any && isum > 0) ? fsum / isum : 0.0;
    }
    if (lane < I) {
        float sf = 0;
        for (int f = 0; f < F; ++f) sf += s.pi[lane * F + f];
        s.spi[lane] = sf + sf;
    }
    __syncthreads();
    for (int c = lane; c < O * 2 * F; c += ADH_WAVE) {
        int o = c / (2 * F), rem = c - o * 2 * F;
        int sc = rem / F, f = rem - sc * F;
        double ds = (double)sc - s.esc[o], df = (double)f - s.efc[o];
        double dist = sqrt(ds * ds + df * df);
        s.wt[c] = exp(-0.1 * dist);
    }
    // precursor heights / observed m/z
    for (int c = lane; c < 2 * I; c += ADH_WAVE) {
        int i = c >> 1, plane = c & 1;
        const float *p = (plane ? s.pm : s.pi) + i * F;
        double values = 0, weights = 0;
        bool any = false;
        for (int sc = 0; sc < 2; ++sc)
            for (int f = 0; f < F; ++f) {
                float v = p[f];
                if (v > 0.0f) {
                    any = true;
                    double w = s.wtp[sc * F + f];
                    values += (double)v * w;
                    weights += w;
                }
            }
        double r = (any && weights > 0) ? values / weights : 0.0;
        if (plane)
            s.omzp[i] = r;
        else
            s.hp[i] = r;
    }
    __syncthreads();

    if (caps.stop_phase == 4) return;
    // ---- best profile + centre envelope (fragment_features.py:240-250)
    int best_obs = 0;
    if (!cfg.quant_all)
        for (int o = 1; o < O; ++o)
            if (s.oi[o] > s.oi[best_obs]) best_obs = o;
    for (int k = lane; k < K; k += ADH_WAVE) {
        float *x;
        if (cfg.quant_all) {
            x = s.bp + k * F;
            for (int f = 0; f < F; ++f) {
                float a = 0;
                for (int o = 0; o < O; ++o) a += s.ffp[(k * O + o) * F + f];
                x[f] = a;
            }
        } else {
            x = s.ffp + (k * O + best_obs) * F;  // a VIEW in the reference: mutated in place
        }
        // center_envelope_1d (fragment_features.py:71-159)
        const int n = F;
        if (n >= 2) {
            if (n % 2 == 0) {
                int cr = n / 2, cl = cr - 1;
                double left = x[cl], right = x[cr];
                for (int i = 1; i <= cl; ++i) {
                    x[cl - i] = (float)fmin(left, (double)x[cl - i]);
                    left = (double)(x[cl - i] + x[cl - i + 1]) * 0.5;
                    x[cr + i] = (float)fmin(right, (double)x[cr + i]);
                    right = (double)(x[cr + i] + x[cr + i - 1]) * 0.5;
                }
            } else {
                int cc = n / 2;
                double left = (double)(x[cc - 1] + x[cc]) * 0.5;
                double right = (double)(x[cc + 1] + x[cc]) * 0.5;
                for (int i = 1; i <= cc; ++i) {
                    x[cc - i] = (float)fmin(left, (double)x[cc - i]);
                    left = (double)(x[cc - i] + x[cc - i + 1]) * 0.5;
                    x[cc + i] = (float)fmin(right, (double)x[cc + i]);
                    right = (double)(x[cc + i] + x[cc + i - 1]) * 0.5;
                }
            }
        }
        if (!cfg.quant_all)
            for (int f = 0; f < F; ++f) s.bp[k * F + f] = x[f];
        // quantification window, trapezoid area (fragment_features.py:252-273)
        int qw = min(F / 2 - 1, (int)cfg.quant_window);
        int center = F / 2;
        int a, b, ra, rb;
        py_slice(center - qw, center + qw + 1, F, a, b);
        py_slice(center - qw, center + qw + 1, n_frame_rt, ra, rb);
        const float *p = s.bp + k * F + a;
        int W = b - a;
        double area = 0;
        for (int i = 0; i + 1 < W && ra + i + 1 < rb; ++i) {
            float sm = p[i + 1] + p[i];
            float drt = s.frame_rt[ra + i + 1] - s.frame_rt[ra + i];
            float m = sm * drt;
            area += (double)m * 0.5;
        }
        s.area[k] = area * (double)qw;
        float t = 0;
        for (int i = 0; i < W; ++i) t += p[i];
        s.obs_int[k] = t;
    }
    // ---- per (fragment, observation) weighted centre means (features_utils.py:9-37)
    for (int c = lane; c < 2 * K * O; c += ADH_WAVE) {
        int plane = c & 1, ko = c >> 1;
        int k = ko / O, o = ko - k * O;
        const float *p = (plane ? s.fm : s.fi) + (s.kmap[k] * O + o) * F;
        const double *w = s.wt + o * 2 * F;
        double values = 0, weights = 0;
        bool any = false;
        for (int sc = 0; sc < 2; ++sc)
            for (int f = 0; f < F; ++f) {
                float v = p[f];
                if (v > 0.0f) {
                    any = true;
                    double ww = w[sc * F + f];
                    values += (double)v * ww;
                    weights += ww;
                }
            }
        double r = (any && weights > 0) ? values / weights : 0.0;
        if (plane)
            s.omz[ko] = r;
        else
            s.ohe[ko] = r;
    }
    __syncthreads();
    // importance-weighted means over observations (fragment_features.py:311-336)
    for (int k = lane; k < K; k += ADH_WAVE) {
        float ws = 0;
        for (int o = 0; o < O; ++o) {
            bool m = s.ohe[k * O + o] > 0;
            float w32 = m ? s.oi[o] : s.oi[o] * 0.0f;
            ws += w32;
        }
        double msum = 0;
        int nm = 0;
        for (int o = 0; o < O; ++o) {
            bool m = s.ohe[k * O + o] > 0;
            float w32 = m ? s.oi[o] : s.oi[o] * 0.0f;
            double w = (double)w32 / ((double)ws + 1e-20);
            if (w > 0) {
                msum += w;
                ++nm;
            }
        }
        double m1 = 0, m2 = 0;
        if (nm > 0)
            for (int o = 0; o < O; ++o) {
                bool m = s.ohe[k * O + o] > 0;
                float w32 = m ? s.oi[o] : s.oi[o] * 0.0f;
                double w = (double)w32 / ((double)ws + 1e-20);
                if (w > 0) {
                    double lw = w / msum;
                    m1 += s.omz[k * O + o] * lw;
                    m2 += s.ohe[k * O + o] * lw;
                }
            }
        s.mzmean[k] = m1;
        s.height[k] = m2;
        s.merr[k] = (m1 - (double)s.g_mz[k]) / (double)s.g_mz[k] * 1e6;  // fragment_features.py:387
        // rank of k in argsort(intensity)[::-1]
        int r = 0;
        float ia = s.g_int[k];
        for (int b = 0; b < K; ++b) {
            float ib = s.g_int[b];
            r += (ib > ia) || (ib == ia && b > k);
        }
        s.ord[r] = k;
    }
    __syncthreads();

    if (caps.stop_phase == 5) return;
    // ---- scalar feature assembly by lane 0 (short sequential float sums)
    if (lane == 0) {
        float *feat = s.feat;
        feat[28] = (float)((double)n_present / (double)K0);  // candidate.py:362
        // location_features.py:8-33
        feat[0] = run.mobility[scan_start] - run.mobility[scan_stop - 1];
        feat[1] = run.rt[frame_stop - 1] - run.rt[frame_start];
        feat[2] = run.rt[frame_center];
        feat[3] = run.mobility[scan_center];

        // precursor_features.py:13-102
        int amax = 0;
        for (int i = 1; i < I; ++i)
            if (s.iso_int[i] > s.iso_int[amax]) amax = i;
        float w4 = 0, w5 = 0, f6 = 0, f7 = 0;
        for (int i = 0; i < I; ++i) {
            float a = 0;
            for (int o = 0; o < O; ++o) a += s.spi[i] * s.oi[o];
            if (i == 0) w4 = a;
            if (i == amax) w5 = a;
            f6 += a;
            f7 += a * s.iso_int[i];
        }
        feat[4] = w4;
        feat[5] = w5;
        feat[6] = f6;
        feat[7] = f7;
        double wme = 0;
        for (int i = 0; i < I; ++i)
            if (s.omzp[i] > 0) {
                double me = (s.omzp[i] - (double)s.iso_mz[i]) / (double)s.iso_mz[i] * 1e6;
                wme += me * (double)s.iso_int[i];
            }
        feat[8] = (float)wme;
        feat[9] = (float)fabs(wme);
        feat[10] = (float)((double)s.iso_mz[0] + wme * 1e-6 * (double)s.iso_mz[0]);
        feat[11] = (float)s.hp[0];
        feat[12] = (float)s.hp[amax];
        {
            double a = 0, b = 0;
            for (int i = 0; i < I; ++i) a += s.hp[i];
            for (int i = 0; i < I; ++i) b += s.hp[i] * (double)s.iso_int[i];
            feat[13] = (float)a;
            feat[14] = (float)b;
        }
        {
            // save_corrcoeff (scoring/utils.py:478-510): (f32, f32) and (f32, f64)
            float sx = 0, sy = 0;
            double sh = 0;
            for (int i = 0; i < I; ++i) sx += s.iso_int[i];
            for (int i = 0; i < I; ++i) sy += s.spi[i];
            for (int i = 0; i < I; ++i) sh += s.hp[i];
            float xb = (float)((double)sx / (double)I), yb = (float)((double)sy / (double)I);
            double hb = sh / (double)I;
            float num = 0, sxx = 0, syy = 0;
            for (int i = 0; i < I; ++i) num += (s.iso_int[i] - xb) * (s.spi[i] - yb);
            for (int i = 0; i < I; ++i) sxx += (s.iso_int[i] - xb) * (s.iso_int[i] - xb);
            for (int i = 0; i < I; ++i) syy += (s.spi[i] - yb) * (s.spi[i] - yb);
            float den = sqrtf(sxx * syy);
            feat[15] = (float)((double)num / ((double)den + 1e-12));
            double numd = 0, shh = 0;
            for (int i = 0; i < I; ++i) numd += (double)(s.iso_int[i] - xb) * (s.hp[i] - hb);
            for (int i = 0; i < I; ++i) shh += (s.hp[i] - hb) * (s.hp[i] - hb);
            double dend = sqrt((double)sxx * shh);
            feat[16] = (float)(numd / (dend + 1e-12));
        }

        // fragment_features.py:198-427
        feat[17] = (float)O;
        int n_height_rows = 0;
        for (int k = 0; k < K; ++k) {
            int cnt = 0;
            for (int o = 0; o < O; ++o) cnt += s.ohe[k * O + o] > 0;
            n_height_rows += cnt > 0;
        }
        if (n_height_rows > 0) feat[18] = (float)corrcoef01(s.area, s.g_fin, K);
        {
            double sh = 0;
            for (int k = 0; k < K; ++k) sh += s.height[k];
            if (sh > 0.0) feat[19] = (float)corrcoef01(s.height, s.g_fin, K);
        }
        int n_int = 0, n_hei = 0;
        float w_int = 0, w_hei = 0;
        for (int k = 0; k < K; ++k)
            if (s.obs_int[k] > 0.0f) {
                ++n_int;
                w_int += s.g_fin[k];
            }
        for (int k = 0; k < K; ++k)
            if (s.height[k] > 0.0) {
                ++n_hei;
                w_hei += s.g_fin[k];
            }
        feat[20] = (float)((double)n_int / (double)K);
        feat[21] = (float)((double)n_hei / (double)K);
        feat[22] = w_int;
        feat[23] = w_hei;
        if (n_int > 0) {
            // cosine_similarity_a1 (features_utils.py:40-47)
            float tn = 0;
            for (int o = 0; o < O; ++o) tn += s.tsum[o] * s.tsum[o];
            tn = sqrtf(tn);
            float acc = 0;
            int cnt = 0;
            for (int k = 0; k < K; ++k) {
                if (!(s.obs_int[k] > 0.0f)) continue;
                const float *rs = s.rowsum + s.kmap[k] * O;
                float fn = 0, dot = 0;
                for (int o = 0; o < O; ++o) fn += rs[o] * rs[o];
                fn = sqrtf(fn);
                for (int o = 0; o < O; ++o) dot += rs[o] * s.tsum[o];
                float pr = fn * tn;
                float score = (float)((double)dot / ((double)pr + 0.0001));
                acc += score;
                ++cnt;
            }
            feat[24] = (float)((double)acc / (double)cnt);
        }
        float sb = 0, sy = 0;
        int nb = 0, ny = 0;
        for (int k = 0; k < K; ++k)
            if (s.g_type[k] == 98) {
                sb += s.obs_int[k];
                ++nb;
            }
        for (int k = 0; k < K; ++k)
            if (s.g_type[k] == 121) {
                sy += s.obs_int[k];
                ++ny;
            }
        feat[25] = nb > 0 ? (float)log((double)sb + 1.0) : 0.0f;
        feat[26] = ny > 0 ? (float)log((double)sy + 1.0) : 0.0f;
        feat[27] = feat[25] - feat[26];
        {
            int n3 = min(K, 3);
            double a = 0, b = 0;
            for (int i = 0; i < n3; ++i) a += s.merr[s.ord[i]];
            for (int k = 0; k < K; ++k) b += s.merr[k];
            feat[41] = (float)(a / (double)n3);
            feat[42] = (float)(b / (double)K);
        }
        if (nb > 0 && ny > 0) {
            int min_y = 255, max_b = 0;
            for (int k = 0; k < K; ++k) {
                if (s.g_type[k] == 121) min_y = min(min_y, (int)s.g_pos[k]);
                if (s.g_type[k] == 98) max_b = max(max_b, (int)s.g_pos[k]);
            }
            int n_ov = 0;
            double sa = 0, se = 0;
            for (int k = 0; k < K; ++k) {
                bool ov = (s.g_type[k] == 121 && (int)s.g_pos[k] < max_b) ||
                          (s.g_type[k] == 98 && (int)s.g_pos[k] > min_y);
                if (ov) {
                    ++n_ov;
                    sa += s.area[k];
                    se += s.merr[k];
                }
            }
            feat[43] = (float)n_ov;
            if (n_ov > 0) {
                feat[44] = (float)(sa / (double)n_ov);
                feat[45] = (float)(se / (double)n_ov);
            } else {
                feat[44] = 0.0f;
                feat[45] = 15.0f;
            }
        }
    }

    if (caps.stop_phase == 6) return;
    // =========================== profile features (profile_features.py:18-206)
    // fi / fm are dead from here on: reuse them as isl[K][F] and nrm[K][F]
    __syncthreads();
    float *isl = s.fi, *nrm = s.fm;
    float *cen = s.fm;  // non-xic path: centred profiles [K][O][F] (nrm unused there)
    if (cfg.experimental_xic) {
        for (int c = lane; c < K * F; c += ADH_WAVE) {
            int k = c / F, f = c - k * F;
            float a = 0;
            for (int o = 0; o < O; ++o) a += s.ffp[(k * O + o) * F + f];
            isl[c] = a;
        }
        __syncthreads();
        // normalize_profiles (scoring_utils.py:71-117)
        int cidx = F / 2, wa, wb;
        py_slice(cidx - 1, cidx + 2, F, wa, wb);
        for (int k = lane; k < K; k += ADH_WAVE) {
            float sm = 0;
            for (int i = wa; i < wb; ++i) sm += isl[k * F + i];
            double ci = (double)sm / (double)(wb - wa);
            for (int f = 0; f < F; ++f)
                nrm[k * F + f] = (ci > 0) ? (float)((double)isl[k * F + f] / ci) : 0.0f;
        }
        __syncthreads();
        // median over fragments per cycle (scoring_utils.py:120-152) by rank selection
        for (int f = lane; f < F; f += ADH_WAVE) {
            float lo_v = 0, hi_v = 0;
            int r_lo = (K - 1) / 2, r_hi = K / 2;
            for (int a = 0; a < K; ++a) {
                float va = nrm[a * F + f];
                int r = 0;
                for (int b = 0; b < K; ++b) {
                    float vb = nrm[b * F + f];
                    r += (vb < va) || (vb == va && b < a);
                }
                if (r == r_lo) lo_v = va;
                if (r == r_hi) hi_v = va;
            }
            float m;
            if (K & 1)
                m = hi_v;
            else {
                float sm = lo_v + hi_v;
                m = (float)((double)sm / 2.0);
            }
            s.med[f] = m;
        }
        __syncthreads();
        // correlation_coefficient (scoring_utils.py:14-68)
        float sx = 0;
        for (int f = 0; f < F; ++f) sx += s.med[f];
        float mx = (float)((double)sx / (double)F);
        for (int f = lane; f < F; f += ADH_WAVE) s.xm[f] = s.med[f] - mx;
        __syncthreads();
        float sxx = 0;
        for (int f = 0; f < F; ++f) sxx += s.xm[f] * s.xm[f];
        double var_x = (double)sxx / (double)F;
        for (int k = lane; k < K; k += ADH_WAVE) {
            float sy = 0;
            for (int f = 0; f < F; ++f) sy += isl[k * F + f];
            float my = (float)((double)sy / (double)F);
            float sxy = 0, syy = 0;
            for (int f = 0; f < F; ++f) sxy += s.xm[f] * (isl[k * F + f] - my);
            for (int f = 0; f < F; ++f) {
                float ym = isl[k * F + f] - my;
                syy += ym * ym;
            }
            double cov = (double)sxy / (double)F;
            double var_y = (double)syy / (double)F;
            double var_xy = var_x * var_y;
            s.corr[k] = (var_xy == 0) ? 0.0f : (float)(cov / sqrt(var_xy));
        }
    } else {
        // fragment_correlation (scoring/utils.py:513-571): centred rows + std per (k, o)
        for (int c = lane; c < K * O; c += ADH_WAVE) {
            const float *p = s.ffp + c * F;
            float sm = 0;
            for (int f = 0; f < F; ++f) sm += p[f];
            float mean = sm / (float)F;
            float q = 0;
            for (int f = 0; f < F; ++f) cen[c * F + f] = p[f] - mean;
            for (int f = 0; f < F; ++f) q += cen[c * F + f] * cen[c * F + f];
            s.fw[c] = sqrtf(q / (float)F);  // std, parked in fw until the FWHM step
        }
        __syncthreads();
        // list[a] = sum_b red[a][b] * intensity[b]; red = sum_o corr_o * importance_o
        for (int a = lane; a < K; a += ADH_WAVE) {
            float acc = 0;
            for (int b = 0; b < K; ++b) {
                float red = 0;
                for (int o = 0; o < O; ++o) {
                    float dot = 0;
                    for (int f = 0; f < F; ++f)
                        dot += cen[(a * O + o) * F + f] * cen[(b * O + o) * F + f];
                    float cov = dot / (float)F;
                    float sm = s.fw[a * O + o] * s.fw[b * O + o];
                    float cm = (float)((double)cov / ((double)sm + 1e-12));
                    red += cm * s.oi[o];
                }
                acc += red * s.g_int[b];
            }
            s.corr[a] = acc;
        }
    }
    __syncthreads();
    float top3 = 0.0f;
    if (lane == 0) {
        int n3 = min(K, 3);
        if (cfg.experimental_xic) {
            float sm = 0;
            for (int i = 0; i < n3; ++i) sm += s.corr[s.ord[i]];
            top3 = (float)((double)sm / (double)n3);
        } else {
            float sm = 0;
            for (int i = 0; i < n3; ++i)
                for (int j = 0; j < n3; ++j) {
                    int a = s.ord[i], b = s.ord[j];
                    float red = 0;
                    for (int o = 0; o < O; ++o) {
                        float dot = 0;
                        for (int f = 0; f < F; ++f)
                            dot += cen[(a * O + o) * F + f] * cen[(b * O + o) * F + f];
                        float cov = dot / (float)F;
                        float sd = s.fw[a * O + o] * s.fw[b * O + o];
                        float cm = (float)((double)cov / ((double)sd + 1e-12));
                        red += cm * s.oi[o];
                    }
                    sm += red;
                }
            top3 = (float)((double)sm / (double)(n3 * n3));
        }
    }
    __syncthreads();
    // fragment-vs-template frame correlation (scoring/utils.py:574-647), FWHM and apex
    for (int c = lane; c < K * O; c += ADH_WAVE) {
        int k = c / O, o = c - k * O;
        const float *px = s.ffp + c * F;
        const float *py = s.tfp + o * F;
        float sy = 0;
        for (int f = 0; f < F; ++f) sy += py[f];
        float ym = sy / (float)F;
        float qy = 0;
        for (int f = 0; f < F; ++f) {
            float d = py[f] - ym;
            qy += d * d;
        }
        float ysd = sqrtf(qy / (float)F);
        float sx = 0;
        for (int f = 0; f < F; ++f) sx += px[f];
        float xmn = sx / (float)F;
        float qx = 0;
        for (int f = 0; f < F; ++f) {
            float d = px[f] - xmn;
            qx += d * d;
        }
        float xsd = sqrtf(qx / (float)F);
        float dot = 0;
        for (int f = 0; f < F; ++f) dot += (px[f] - xmn) * (py[f] - ym);
        float cov = dot / (float)F;
        float sm = xsd * ysd;
        s.ftc[o * K + k] = (float)((double)cov / ((double)sm + 1e-12));
        // FWHM in RT (profile_features.py:117-146) and apex (profile_features.py:192-193)
        float mxv = px[0];
        int am = 0;
        for (int f = 1; f < F; ++f)
            if (px[f] > mxv) {
                mxv = px[f];
                am = f;
            }
        double half = (double)mxv / 2.0;
        int n_above = 0;
        for (int f = 0; f < F; ++f) n_above += ((double)px[f] > half);
        double frac = (double)n_above / (double)F;
        float rt_width = run.rt[frame_stop - 1] - run.rt[frame_start];
        s.fpeak[c] = am;
        s.fw[c] = (float)(frac * (double)rt_width);  // std values parked here are dead by now
    }
    __syncthreads();
    if (lane < O) {
        // median of the apex index over fragments (profile_features.py:196-198)
        int o = lane;
        int lo_v = 0, hi_v = 0, r_lo = (K - 1) / 2, r_hi = K / 2;
        for (int a = 0; a < K; ++a) {
            int va = s.fpeak[a * O + o];
            int r = 0;
            for (int b = 0; b < K; ++b) {
                int vb = s.fpeak[b * O + o];
                r += (vb < va) || (vb == va && b < a);
            }
            if (r == r_lo) lo_v = va;
            if (r == r_hi) hi_v = va;
        }
        double m = (K & 1) ? (double)hi_v : (double)(lo_v + hi_v) / 2.0;
        s.medpk[o] = (float)m;
    }
    __syncthreads();
    if (lane == 0) {
        float *feat = s.feat;
        float sm = 0;
        for (int k = 0; k < K; ++k) sm += s.corr[k];
        feat[31] = (float)((double)sm / (double)K);
        feat[32] = top3;
        float dot = 0;
        for (int k = 0; k < K; ++k) {
            float r = 0;
            for (int o = 0; o < O; ++o) r += s.ftc[o * K + k] * s.oi[o];
            dot += r * s.g_int[k];
        }
        feat[33] = dot;
        // b / y: mask in original order applied to the sorted index array (profile_features.py:94-113)
        int nbi = 0, nyi = 0;
        float sbv = 0, syv = 0;
        for (int k = 0; k < K; ++k) {
            if (s.g_type[k] == 98) {
                if (nbi < 3) sbv += s.corr[s.ord[k]];
                ++nbi;
            }
        }
        for (int k = 0; k < K; ++k) {
            if (s.g_type[k] == 121) {
                if (nyi < 3) syv += s.corr[s.ord[k]];
                ++nyi;
            }
        }
        if (nbi > 0) {
            feat[34] = (float)((double)sbv / (double)min(nbi, 3));
            feat[35] = (float)nbi;
        }
        if (nyi > 0) {
            feat[36] = (float)((double)syv / (double)min(nyi, 3));
            feat[37] = (float)nyi;
        }
        float agg = 0;
        for (int k = 0; k < K; ++k) {
            float ml = 0;
            for (int o = 0; o < O; ++o) ml += s.fw[k * O + o] * s.oi[o];
            agg += ml * s.g_int[k];
        }
        feat[38] = agg;
        double acc = 0;
        for (int o = 0; o < O; ++o) {
            double delta = (double)s.medpk[o] - floor((double)F / 2.0);
            acc += delta * (double)s.oi[o];
        }
        feat[40] = (float)acc;
    }
    __syncthreads();

    if (caps.stop_phase == 7) return;
    // ---- write the row: features, fragment table, valid flag (candidate.py:403-481)
    if (lane < ADH_NUM_FEATURES) out.features[row * ADH_NUM_FEATURES + lane] = s.feat[lane];
    if (cfg.collect_fragments) {
        const int n = min(K, top_k);
        const int64_t base = row * (int64_t)top_k;
        for (int k = lane; k < n; k += ADH_WAVE) {
            out.fragment_precursor_idx[base + k] = precursor_idx;
            out.fragment_rank[base + k] = rank;
            out.fragment_mz_library[base + k] = s.g_mzlib[k];
            out.fragment_mz[base + k] = s.g_mz[k];
            out.fragment_mz_observed[base + k] = (float)s.mzmean[k];
            out.fragment_height[base + k] = (float)s.height[k];
            out.fragment_intensity[base + k] = (float)s.area[k];
            out.fragment_mass_error[base + k] = (float)s.merr[k];
            out.fragment_correlation[base + k] = s.corr[k];
            out.fragment_position[base + k] = s.g_pos[k];
            out.fragment_number[base + k] = s.g_number[k];
            out.fragment_type[base + k] = s.g_type[k];
            out.fragment_charge[base + k] = s.g_charge[k];
            out.fragment_loss_type[base + k] = s.g_loss[k];
        }
    }
    if (lane == 0) out.valid[row] = 1;
}

// ------------------------------------------------------------------ bucket index
// one thread per peak: write the table entries whose first peak this is.
__global__ void adh_bucket_build_kernel(const float *mz, const int64_t *pstart, const int64_t *pstop,
                                        int64_t n_spectra, uint32_t *bucket, int nb, float bmin,
                                        float binv) {
    int64_t spec = blockIdx.x;
    if (spec >= n_spectra) return;
    const int64_t ps = pstart[spec], pe = pstop[spec];
    uint32_t *tab = bucket + spec * (int64_t)(nb + 1);
    const int64_t n = pe - ps;
    for (int64_t j = threadIdx.x; j <= n; j += blockDim.x) {
        // entries (b_prev, b_cur] point at peak j; j == n closes the table
        int b_prev = (j == 0) ? -1 : adh_bucket_of(mz[ps + j - 1], bmin, binv, nb);
        int b_cur = (j == n) ? nb : adh_bucket_of(mz[ps + j], bmin, binv, nb);
        for (int b = b_prev + 1; b <= b_cur; ++b) tab[b] = (uint32_t)j;
    }
}
