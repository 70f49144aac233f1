// adh_features_im.hip - the 46-feature stack for ion-mobility (timsTOF) candidates.
//
// Same phases as adh_features.hip (the restatement of Candidate.process after get_dense,
// alphadia/search/scoring/containers/candidate.py:248-481) with a REAL scan axis:
//   * the tiles normally arrive in sparse form (sorted non-zero cells, adh_device.h); the fragment
//     cells are folded ONCE, one lane per (fragment, observation) plane: scan profile, frame
//     profile, row sum and weighted centre means in the reference's order; LDS holds profiles,
//     the template and the weight tables only
//   * rt / mobility arrays are float64
//     (TimsTOFTransposeJIT, alphadia/search/jitclasses/bruker_jit.py:35,45), which changes
//     the typing of the quantification area and of the location / FWHM features
//   * the quadrupole transfer function is evaluated per scan (quadrupole.py:261-301)
//   * scan profiles get their OR-envelope (scoring/utils.py:56-66)
//   * the ion-mobility-only features are computed: fragment / template scan correlation
//     (fragment_features.py:430-480 -> features 29, 30) and mobility FWHM
//     (profile_features.py:151-188 -> feature 39)
// One 64-lane wavefront per candidate; float32 reductions keep the reference's order.
// The K x K scan-profile correlation is a small dense contraction (12 x 12 x S per candidate,
// BLAS SGEMM in the reference): one v_mfma_f32_16x16x4_f32 tile per observation.
#include "adh_device.h"
#include "adh_feature_common.h"

namespace featim {
using feat::Assemble;

__device__ __forceinline__ void py_slice(int start, int stop, int n, int &a, int &b) {
    if (start < 0) start += n;
    if (stop < 0) stop += n;
    a = min(max(start, 0), n);
    b = min(max(stop, 0), n);
    if (b < a) b = a;
}

__device__ __forceinline__ double logistic(double x, double mu, double sigma) {
    double a = (x - mu) / sigma;
    return 1.0 / (1.0 + exp(-a));
}

// Register rows.  A loop `for (i < n) s += p[i]` over an LDS row with a run-time bound is a chain of LDS
// latencies (load, add, load, add ...) on a wavefront that has nothing else to do; with the row in
// registers - all loads issued together, compile-time indices, the tail beyond n filled with +0 - the chain
// is the additions alone.  Adding +0 leaves a float sum as it is (sums start at +0), so the order and the
// bits of the reference's sequential sums are kept.  FR cycles / SR scans fit; longer rows take the loops.
constexpr int FR = 32;
constexpr int SR = 40;
constexpr int KR = 16;

template <int N>
__device__ __forceinline__ void load_row(float (&x)[N], const float *p, int stride, int n) {
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = (i < n) ? p[i * stride] : 0.0f;
}
// p[0] + p[stride] + ... (n terms, in order)
template <int N>
__device__ __forceinline__ float osum(const float *p, int stride, int n) {
    float s = 0.0f;
    if (n <= N) {
        float v[N];
        load_row<N>(v, p, stride, n);
#pragma unroll
        for (int i = 0; i < N; ++i) s += v[i];
    } else {
        for (int i = 0; i < n; ++i) s += p[i * stride];
    }
    return s;
}
// Pearson statistics of a register row against a second one, as scoring/utils.py:574-647 takes them:
// means, population standard deviations and the covariance, float32 sums in index order
template <int N>
__device__ __forceinline__ void row_moments(const float (&x)[N], int n, float &mean, float &sd) {
    float sx = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) sx += x[i];
    mean = sx / (float)n;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float d = x[i] - mean;
        d = (i < n) ? d : 0.0f;
        q += d * d;
    }
    sd = sqrtf(q / (float)n);
}

// Precursor features 4-16 (precursor_features.py:13-102; feat::assemble_precursor is the loop form) for up to
// three isotopes and four observations: inputs to registers first, every loop unrolled, so that the one lane
// that runs this waits for no LDS load in the middle of a sum
__device__ __forceinline__ void precursor_features_rows(float *ft, int I, int O, const float *iso_int_p,
                                                        const float *iso_mz_p, const float *spi_p, const double *hp_p,
                                                        const double *omzp_p, const float *oi_p) {
    float ii[3], mz[3], spi[3], oi[4];
    double hp[3], omzp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = i < I ? i : 0;
        ii[i] = iso_int_p[j];
        mz[i] = iso_mz_p[j];
        spi[i] = spi_p[j];
        hp[i] = hp_p[j];
        omzp[i] = omzp_p[j];
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) oi[o] = oi_p[o < O ? o : 0];
    int amax = 0;
#pragma unroll
    for (int i = 1; i < 3; ++i)
        if (i < I && ii[i] > (amax == 0 ? ii[0] : (amax == 1 ? ii[1] : ii[2]))) amax = i;
    float w4 = 0, w5 = 0, f6 = 0, f7 = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < I) {
            float a = 0;
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < O) a += spi[i] * oi[o];
            if (i == 0) w4 = a;
            if (i == amax) w5 = a;
            f6 += a;
            f7 += a * ii[i];
        }
    }
    ft[4] = w4;
    ft[5] = w5;
    ft[6] = f6;
    ft[7] = f7;
    double wme = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (i < I && omzp[i] > 0) {
            const double me = (omzp[i] - (double)mz[i]) / (double)mz[i] * 1e6;
            wme += me * (double)ii[i];
        }
    ft[8] = (float)wme;
    ft[9] = (float)fabs(wme);
    ft[10] = (float)((double)mz[0] + wme * 1e-6 * (double)mz[0]);
    ft[11] = (float)hp[0];
    ft[12] = (float)(amax == 0 ? hp[0] : (amax == 1 ? hp[1] : hp[2]));
    {
        double a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < I) a += hp[i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < I) b += hp[i] * (double)ii[i];
        ft[13] = (float)a;
        ft[14] = (float)b;
    }
    {
        // save_corrcoeff (scoring/utils.py:478-510): (f32, f32) and (f32, f64)
        float sx = 0, sy = 0;
        double sh = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < I) sx += ii[i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < I) sy += spi[i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < I) sh += hp[i];
        const float xb = (float)((double)sx / (double)I), yb = (float)((double)sy / (double)I);
        const double hb = sh / (double)I;
        float num = 0, sxx = 0, syy = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < I) num += (ii[i] - xb) * (spi[i] - yb);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < I) sxx += (ii[i] - xb) * (ii[i] - xb);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < I) syy += (spi[i] - yb) * (spi[i] - yb);
        const float den = sqrtf(sxx * syy);
        ft[15] = (float)((double)num / ((double)den + 1e-12));
        double numd = 0, shh = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < I) numd += (double)(ii[i] - xb) * (hp[i] - hb);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < I) shh += (hp[i] - hb) * (hp[i] - hb);
        const double dend = sqrt((double)sxx * shh);
        ft[16] = (float)(numd / (dend + 1e-12));
    }
}

// LDS capacities of a launch: taken from the launch's Caps (DimsDyn) or fixed at compile time (DimsFix), which
// turns every array address below into an instruction offset - with run-time capacities the ~60 array bases
// are scalar registers, more than there are, and the kernel spends a sixth of its instructions moving
// them between scalar registers and the lanes of spill registers
struct DimsDyn {
    int Kc, Oc, Sc, Fc, Ic, SFc;  // SFc: cells of one (scan, cycle) plane
    __host__ __device__ DimsDyn(const Caps &c) : Kc(c.k), Oc(c.o), Sc(c.s), Fc(c.f), Ic(c.i), SFc(c.s * c.f) {}
};
template <int K, int O, int S, int F, int I, int SF>
struct DimsFix {
    static constexpr int Kc = K, Oc = O, Sc = S, Fc = F, Ic = I, SFc = SF;
    __host__ __device__ DimsFix(const Caps &) {}
    __host__ __device__ static bool holds(const Caps &c) { return holds_axes(c) && c.s * c.f <= SF; }
    // (without the plane size: for a launch whose candidates were admitted one by one, plan class ADH_CLASS_IM_SMALL)
    __host__ __device__ static bool holds_axes(const Caps &c) {
        return c.k <= K && c.o <= O && c.s <= S && c.f <= F && c.i <= I;
    }
};
template <class D>
struct LayoutT : D {
    __host__ __device__ LayoutT(const Caps &c) : D(c) {}
    // doubles
    // (the transfer function is dead after the template; the per-fragment results and the frame times,
    // born later, take its place)
    __host__ __device__ int d_qtf() const { return 0; }                      // [Ic][Oc][Sc], early
    __host__ __device__ int qtf_size() const {
        const int a = D::Ic * D::Oc * D::Sc, b = 4 * D::Kc + D::Fc;
        return a > b ? a : b;
    }
    __host__ __device__ int d_pk() const { return d_qtf(); }                 // [4][Kc], late
    __host__ __device__ int d_frt() const { return d_qtf() + 4 * D::Kc; }       // [Fc] frame rt (float64), late
    __host__ __device__ int d_omz() const { return d_qtf() + qtf_size(); }
    __host__ __device__ int d_ohe() const { return d_omz() + D::Kc * D::Oc; }
    __host__ __device__ int d_omzu() const { return d_ohe() + D::Kc * D::Oc; }  // per selected fragment,
    __host__ __device__ int d_oheu() const { return d_omzu() + D::Kc * D::Oc; } // before the presence mask
    __host__ __device__ int d_accw() const { return d_oheu() + D::Kc * D::Oc; } // [2][Kc*Oc] weight sums
    __host__ __device__ int d_po() const { return d_accw() + 2 * D::Kc * D::Oc; }   // [2][Oc]
    __host__ __device__ int d_pi() const { return d_po() + 2 * D::Oc; }     // [2][Ic]
    __host__ __device__ int n_double() const { return d_pi() + 2 * D::Ic; }
    // floats
    __host__ __device__ int smax() const { return D::Sc > D::Fc ? D::Sc : D::Fc; }
    __host__ __device__ int f_wb() const { return 0; }                            // work [Kc*Oc*max(Sc,Fc)]
    // Two regions are used twice.  R1 holds the template until its profiles are taken, then the masked
    // frame / scan profiles; R2 holds the profiles before the presence mask, then the scan envelopes and
    // the quantification profiles.  (LDS per block is what bounds the resident waves of this kernel, and
    // the kernel's time follows them: 26 KB -> 40 KB per block costs 38 %.)
    __host__ __device__ int f_work_end() const { return f_wb() + D::Kc * D::Oc * smax(); }
    __host__ __device__ int r1_size() const {
        const int a = D::Oc * D::SFc, b = D::Kc * D::Oc * (D::Fc + D::Sc);
        return a > b ? a : b;
    }
    __host__ __device__ int f_r1() const { return f_work_end(); }
    __host__ __device__ int f_r2() const { return f_r1() + r1_size(); }
    __host__ __device__ int f_tpl() const { return f_r1(); }                      // [Oc*Sc*Fc]       R1, early
    __host__ __device__ int f_ffp() const { return f_r1(); }                      // [Kc*Oc*Fc]       R1, late
    __host__ __device__ int f_fspr() const { return f_r1() + D::Kc * D::Oc * D::Fc; }      // [Kc*Oc*Sc] raw   R1, late
    __host__ __device__ int f_ffpu() const { return f_r2(); }                     // [Kc*Oc*Fc] before the mask  R2, early
    __host__ __device__ int f_fspu() const { return f_r2() + D::Kc * D::Oc * D::Fc; }      // [Kc*Oc*Sc] before the mask  R2, early
    __host__ __device__ int f_fspe() const { return f_r2(); }                     // [Kc*Oc*Sc] envelope  R2, late
    __host__ __device__ int f_bp() const { return f_r2() + D::Kc * D::Oc * D::Sc; }        // [Kc*Fc]          R2, late
    __host__ __device__ int f_tfp() const { return f_r2() + D::Kc * D::Oc * (D::Fc + D::Sc); }  // [Oc*Fc]
    __host__ __device__ int f_tsp() const { return f_tfp() + 2 * D::Oc * D::Fc; }       // [2][Oc*Sc] (tfp: raw, env)
    __host__ __device__ int f_qm() const { return f_tsp() + 2 * D::Oc * D::Sc; }        // [Oc*Sc] qtf mask
    __host__ __device__ int f_pk() const { return f_qm() + D::Oc * D::Sc; }             // [8][Kc]
    __host__ __device__ int f_pko() const { return f_pk() + 8 * D::Kc; }             // [4][Kc*Oc]
    __host__ __device__ int f_po() const { return f_pko() + 4 * D::Kc * D::Oc; }        // [4][Oc]
    __host__ __device__ int f_pi() const { return f_po() + 4 * D::Oc; }              // [3][Ic]
    __host__ __device__ int f_pf() const { return f_pi() + 3 * D::Ic; }              // [2][Fc]
    __host__ __device__ int f_feat() const { return f_pf() + 2 * D::Fc; }
    __host__ __device__ int n_float() const { return f_feat() + ADH_NUM_FEATURES; }
    // ints
    __host__ __device__ int i_pk() const { return 0; }                   // [4][Kc]
    __host__ __device__ int i_pko() const { return i_pk() + 4 * D::Kc; }    // [Kc*Oc]
    __host__ __device__ int i_obs() const { return i_pko() + D::Kc * D::Oc; }  // [Oc]
    __host__ __device__ int n_int() const { return i_obs() + D::Oc; }
    __host__ __device__ int n_byte() const { return ((5 * D::Kc + 7) / 8) * 8; }
    __host__ __device__ size_t bytes() const {
        size_t b = (size_t)n_double() * 8;
        b += ((size_t)n_float() * 4 + 7) / 8 * 8;
        b += ((size_t)n_int() * 4 + 7) / 8 * 8;
        b += n_byte();
        return b;
    }
};
using Layout = LayoutT<DimsDyn>;
// the common shape: up to 12 fragments, one observation, 40 scans, 32 cycles, 1152 cells per plane, 3 isotopes
// (13 704 + 2 560 static bytes: ten blocks per CU, as the specified 38 x 29 tiles get with run-time capacities)
using DimsCommon = DimsFix<12, 1, 40, 32, 3, 1152>;
using LayoutCommon = LayoutT<DimsCommon>;
// the same shape with two observations (plan class 1; only the tile part of the split path uses it)
using DimsCommon2 = DimsFix<12, 2, 40, 32, 3, 1152>;
using LayoutCommon2 = LayoutT<DimsCommon2>;
// small tiles (plan class ADH_CLASS_IM_SMALL): 10 248 + 2 560 bytes, twelve blocks per CU
using DimsSmall = DimsFix<ADH_IM_SMALL_K, 1, ADH_IM_SMALL_S, ADH_IM_SMALL_F, 3, ADH_IM_SMALL_SF>;
using LayoutSmall = LayoutT<DimsSmall>;

}  // namespace featim

#define ADH_IM_STAGE 128          // cells staged per round of the tile passes (two per lane)
#define ADH_IM_STATIC_LDS (ADH_IM_STAGE * 20)  // static LDS of adh_feature_im_kernel (chunk lists / Gram matrices: 2 176 B)
size_t adh_feature_im_lds_bytes(const Caps &c) {
    return featim::DimsSmall::holds(c)    ? featim::LayoutSmall(c).bytes()
           : featim::DimsCommon::holds(c) ? featim::LayoutCommon(c).bytes()
                                          : featim::Layout(c).bytes();
}

// SPLIT (round 4, fixed layouts only): the kernel ends after the passes over the tiles and writes the candidate's
// ImProfRec<LAY::Fc, LAY::Sc> to `prof` (record blockIdx.x); adh_feature_im_profiles_kernel does the rest, four
// candidates per wavefront.
// The kernel's body for candidate `ci` of `plan` (the kernels below: one block per candidate, or a block per list entry)
// DYNPOOL: the 2.5 KB of chunk lists behind the layout in the dynamic LDS block instead of a static array - for the
// kernels that take the rare materialised tiles through this body in a few blocks of a grid whose other blocks should
// not pay for it (adh_feature_im_fused4_kernel, adh_feature_im_tile4_kernel: LDS is their occupancy).
template <class LAY, bool SPLIT, bool DYNPOOL = false>
__device__ __forceinline__ void adh_feature_im_body(
    const int ci, const DevTims &run, const CandRecIM *__restrict__ plan, const float *__restrict__ iso_table,
    int32_t n_iso_cols, const adh_scoring_config_t &cfg, const unsigned char *__restrict__ scratch, const DevOut &out,
    const Caps &caps, unsigned char *__restrict__ prof) {
    using namespace featim;
    extern __shared__ __align__(16) unsigned char smem[];
    // ordered list of the non-zero cells of one 64-cell chunk (see the tile passes below)
    // static LDS: the chunk lists of the tile pass; the two 16 x 17 matrices of the scan
    // correlation reuse the same bytes later (an extra 2 KB of LDS per wavefront cost 22 % of the
    // kernel's throughput in resident waves)
    __shared__ __align__(16) unsigned char pool_static[DYNPOOL ? 16 : ADH_IM_STATIC_LDS];
    unsigned char *const pool = DYNPOOL ? smem + ((LAY(caps).bytes() + 15) / 16 * 16) : pool_static;
    double *const l_ti = reinterpret_cast<double *>(pool);
    double *const l_tm = l_ti + ADH_WAVE;
    double *const l_w = l_tm + ADH_WAVE;
    float *const l_v = reinterpret_cast<float *>(l_w + ADH_WAVE);
    float *const l_rx = l_v + ADH_WAVE;
    float *const l_ry = l_rx + ADH_WAVE;
    const LAY lay(caps);
    const int Kc = lay.Kc, Oc = lay.Oc, Sc = lay.Sc, Fc = lay.Fc, Ic = lay.Ic;
    double *const D = reinterpret_cast<double *>(smem);
    float *const Fl = reinterpret_cast<float *>(smem + (size_t)lay.n_double() * 8);
    int *const In = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(Fl) +
                                           ((size_t)lay.n_float() * 4 + 7) / 8 * 8);
    uint8_t *const By = reinterpret_cast<uint8_t *>(In) + ((size_t)lay.n_int() * 4 + 7) / 8 * 8;

    const int lane = threadIdx.x;
    const CandRecIM &r = plan[ci];
    if (r.flags & ADH_FLAG_SKIP) return;
    const unsigned char *block = scratch + r.scratch_off;
    const uint32_t *header = reinterpret_cast<const uint32_t *>(block);
    const int K0 = (int)header[0];
    if (K0 == 0) return;
    const uint32_t row = r.row;
    const int L = run.cycle_len, z = run.zeroth;
    const int c0 = (r.frame_start - z) / L;
    const int F = (r.frame_stop - z) / L - c0;
    const int S = r.scan_stop - r.scan_start;
    // (a candidate that gets this far has at least one observation - the gather kernel leaves K0 = 0 otherwise -
    // so a launch whose capacity is one observation has exactly one: a compile-time 1 in the fixed layout)
    const int O = (lay.Oc == 1) ? 1 : r.n_obs, Op = r.n_ms1;
    const int I = min(n_iso_cols, (int)cfg.top_k_isotopes);
    const int SF = S * F, OSF = O * SF;
    const int top_k = out.top_k;
    if (lane == 0 && out.stat_matched_peaks) out.stat_matched_peaks[row] = header[1];
    if (caps.stop_phase == 10) return;  // developer ablation switches (ADH_DEBUG_IM)

    float *const work_b = Fl + lay.f_wb();
    float *const ffp_u = Fl + lay.f_ffpu();
    float *const fsp_u = Fl + lay.f_fspu();
    float *const ffp = Fl + lay.f_ffp();
    float *const fsp_raw = Fl + lay.f_fspr();
    float *const fsp = Fl + lay.f_fspe();
    float *const tpl = Fl + lay.f_tpl();
    float *const tfp_raw = Fl + lay.f_tfp();
    float *const tfp = tfp_raw + Oc * Fc;
    float *const tsp_raw = Fl + lay.f_tsp();
    float *const tsp = tsp_raw + Oc * Sc;
    float *const qmask = Fl + lay.f_qm();
    float *const bp = Fl + lay.f_bp();
    float *const oi = Fl + lay.f_po();
    float *const tsum = oi + Oc;
    float *const medpk = tsum + Oc;
    float *const iso_mz = Fl + lay.f_pi();
    float *const iso_int = iso_mz + Ic;
    float *const spi = iso_int + Ic;
    float *const med = Fl + lay.f_pf();
    float *const xm = med + Fc;
    float *const featv = Fl + lay.f_feat();
    int *const present = In + lay.i_pk();
    int *const kmap = present + Kc;
    int *const ord = kmap + Kc;
    int *const mkeep = ord + Kc;
    int *const fpeak = In + lay.i_pko();
    int *const obs = In + lay.i_obs();
    double *const qtf = D + lay.d_qtf();
    double *const frame_rt = D + lay.d_frt();

    // ---- the fragment tile [k][o][s][f] and the raw precursor tile [i][Op][s][f] stay in the scratch
    // block.  The MS1 observations are collapsed (candidate.py:248-269) where a cell is used: with one
    // unfragmented frame per cycle (the usual acquisition) that is the cell itself, and keeping the
    // collapsed tile out of LDS (26 KB at 38 scans x 29 cycles) more than doubles the resident waves.
    const float2 *const fcells = reinterpret_cast<const float2 *>(block + adh_scratch_frag_off(r.k_cap));
    const float2 *const pcells = reinterpret_cast<const float2 *>(block + adh_im_prec_off(r.k_cap, O, S, F));
    // sparse form of both tiles (see adh_device.h): fragment entries, then the collapsed precursor entries
    const bool compact = header[3] == ADH_IM_MODE_COMPACT;
    const ImEntry *const entries = reinterpret_cast<const ImEntry *>(block + adh_scratch_frag_off(r.k_cap));
    const int n_fe = compact ? (int)header[2] : 0, n_pe = compact ? (int)header[4] : 0;
    auto prec_cell = [&](int i, int sf) -> float2 {  // (summed intensity, mean m/z of the non-empty observations)
        float acc = 0.0f;
        double sum = 0.0;
        int count = 0;
        for (int j = 0; j < Op; ++j) {
            const float2 v = pcells[(i * Op + j) * SF + sf];
            acc += v.x;
            sum += (double)v.y;
            count += v.y > 0.0f;
        }
        return make_float2(acc, (float)(sum / ((double)count + 1e-6)));
    };
    auto prec_int = [&](int i, int sf) -> float {  // the intensity plane only
        float acc = 0.0f;
        for (int j = 0; j < Op; ++j) acc += pcells[(i * Op + j) * SF + sf].x;
        return acc;
    };
    {
        if (lane < I) {
            iso_int[lane] = iso_table[(int64_t)row * n_iso_cols + lane];
            double off = (double)lane * 1.0033548350700006 / (double)r.charge;
            iso_mz[lane] = (float)off + r.precursor_mz;
        }
        if (lane < O) obs[lane] = r.obs[lane];
        if (lane < ADH_NUM_FEATURES) featv[lane] = 0.0f;
    }
    adh_wave_sync();

    // ---- quadrupole transfer function per (isotope, observation, scan) (quadrupole.py:261-301)
    for (int c = lane; c < I * O * S; c += ADH_WAVE) {
        int i = c / (O * S), rem = c - i * O * S;
        int o = rem / S, sc = rem - o * S;
        const double *cy = run.cycle + 2 * ((int64_t)obs[o] * run.scan_max + (r.scan_start + sc));
        double x = (double)iso_mz[i];
        const QuadParams qp = adh_quad_params(cfg);
        qtf[c] = logistic(x, cy[0] + qp.delta_lo, qp.sigma_lo) - logistic(x, cy[1] + qp.delta_hi, qp.sigma_hi);
    }
    adh_wave_sync();
    for (int c = lane; c < O * S; c += ADH_WAVE) {
        double sum = 0;
        for (int i = 0; i < I; ++i) sum += qtf[i * O * S + c];
        qmask[c] = (float)(sum / (double)I);  // candidate.py:287-289
    }
    adh_wave_sync();
    // (the qtf mask of candidate.py:290 is applied while the fragment tile is streamed)
    double *const hp = D + lay.d_pi();
    double *const omzp = hp + Ic;
    if (caps.stop_phase == 11) return;
    // The template (O, S, F) is as sparse as the precursor tile it is made of: ~20 cells of 1 152.  The split path
    // (one observation) never builds it: its non-zero cells go to a list in cell order - (scan << 12 | cycle,
    // value), in the bytes the dense template would take - and ONE lane folds the list into everything the
    // template is needed for: scan profile, frame profile, centre of mass.  Empty cells add +0 to those sums.
    constexpr int TL_CAP = 256;
    const bool sparse_tpl = SPLIT && compact && n_pe <= TL_CAP && lay.Oc == 1 && lay.Oc * lay.SFc * 4 >= TL_CAP * 8;
    int *const tl_cell = reinterpret_cast<int *>(tpl);
    float *const tl_val = tpl + TL_CAP;
    int tl_n = 0;
    if (compact) {
        // ---- everything that reads the precursor tile, in ONE pass over its sparse form: the entries are
        // the non-zero (scan, cycle, isotope) cells in that order, staged ADH_IM_STAGE at a time in the chunk lists.
        //   template (O, S, F) (quadrupole.py:304-324): one lane per (scan, cycle) cell with any isotope
        //   isotope intensity sums (per scan, then over the scans) and the weighted centre means of the
        //   isotope planes around (scan, frame) = (S, 1) (precursor_features.py:52-66): sequential sums in
        //   (scan, cycle) order - lanes 0..5I-1 walk the staged entries, one sum each.  Empty cells add
        //   0 to every one of these sums, so skipping them changes nothing.
        double *const c_w = reinterpret_cast<double *>(pool);
        uint32_t *const c_cell = reinterpret_cast<uint32_t *>(c_w + ADH_IM_STAGE);
        float *const c_x = reinterpret_cast<float *>(c_cell + ADH_IM_STAGE);
        float *const c_y = c_x + ADH_IM_STAGE;
        if (!sparse_tpl)
            for (int c = lane; c < OSF; c += ADH_WAVE) tpl[c] = 0.0f;
        const ImEntry *const pent = entries + n_fe;
        const int role = lane / I, iso = lane - role * I;  // role 0: intensity sum; 1, 2: intensity mean; 3, 4: m/z mean
        double acc = 0.0;
        float part = 0.0f, tot = 0.0f;
        int cur_sc = -1;
        // (a staged cell is packed as scan << 16 | cycle << 4 | isotope: the sequential walks below must
        // not divide)
        for (int base = 0; base < n_pe;) {
            int cnt = min(ADH_IM_STAGE, n_pe - base);
            for (int e = lane; e < cnt; e += ADH_WAVE) {
                const ImEntry en = pent[base + e];
                const int sf = (int)en.cell / I, i = (int)en.cell - sf * I, sc = sf / F, f = sf - sc * F;
                const double ds = (double)(sc - S), df = (double)(f - 1);
                c_cell[e] = (uint32_t)(sc << 16 | f << 4 | i);
                c_x[e] = en.x;
                c_y[e] = en.y;
                c_w[e] = exp(-0.1 * sqrt(ds * ds + df * df));
            }
            adh_wave_sync();
            if (base + cnt < n_pe) {  // never cut the isotopes of a cell in two: stop at the last cell start
                const int e = cnt - ADH_WAVE + lane;
                const unsigned long long st = __ballot((c_cell[e] >> 4) != (c_cell[e - 1] >> 4));
                cnt = cnt - ADH_WAVE + (63 - __clzll(st));
            }
            for (int e0 = 0; e0 < cnt; e0 += ADH_WAVE) {  // (every lane makes every trip: the list positions come from a ballot)
                const int e = e0 + lane;
                const uint32_t cs = e < cnt ? c_cell[e] >> 4 : 0u;
                const bool first = e < cnt && !(e > 0 && (c_cell[e - 1] >> 4) == cs);
                const int sc = (int)(cs >> 12), sf = sc * F + (int)(cs & 0xFFFu);
                const unsigned long long firsts = sparse_tpl ? __ballot(first) : 0ull;
                if (first) {
                    for (int o = 0; o < O; ++o) {
                        double a = 0;
                        for (int q = e; q < cnt && (c_cell[q] >> 4) == cs; ++q) {
                            const int i = (int)(c_cell[q] & 15u);
                            const float t = c_x[q] * iso_int[i];
                            a += (double)t * qtf[(i * O + o) * S + sc];
                        }
                        if (sparse_tpl) {  // (one observation) the cell joins the list, which stays in cell order
                            const int at = tl_n + __popcll(firsts & ((1ull << lane) - 1ull));
                            tl_cell[at] = (int)cs;
                            tl_val[at] = (float)a;
                        } else {
                            tpl[o * SF + sf] = (float)a;
                        }
                    }
                }
                tl_n += __popcll(firsts);
            }
            if (lane < 5 * I) {
                // four entries per step, all loads first, no branch: the walk is a chain of LDS latencies
                // otherwise.  (Adding 0.0 leaves a sum as it is.)
                for (int e0 = 0; e0 < cnt; e0 += 4) {
                    uint32_t cp[4];
                    float xs[4], ys[4];
                    double ws[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = min(e0 + u, cnt - 1);
                        cp[u] = c_cell[e];
                        xs[u] = c_x[e];
                        ys[u] = c_y[e];
                        ws[u] = c_w[e];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool mine = e0 + u < cnt && (int)(cp[u] & 15u) == iso;
                        const int sc = (int)(cp[u] >> 16);
                        const bool fresh = mine && sc != cur_sc;  // role 0: per-scan sums, added up in scan order
                        tot += fresh ? part : 0.0f;
                        part = fresh ? 0.0f : part;
                        cur_sc = fresh ? sc : cur_sc;
                        part += mine ? xs[u] : 0.0f;
                        const float flag = role <= 2 ? xs[u] : ys[u];
                        const double term = role == 1 ? (double)xs[u] * ws[u] : (role == 3 ? (double)ys[u] * ws[u] : ws[u]);
                        acc += mine && flag > 0.0f ? term : 0.0;
                    }
                }
            }
            adh_wave_sync();
            base += cnt;
        }
        const int il = lane < I ? lane : 0;
        const double vh = __shfl(acc, I + il), wh = __shfl(acc, 2 * I + il);
        const double vmz = __shfl(acc, 3 * I + il), wmz = __shfl(acc, 4 * I + il);
        if (lane < I) {
            spi[lane] = tot + part;
            hp[lane] = (wh > 0) ? vh / wh : 0.0;  // weights are exp(...) > 0: "any non-zero cell" == "w sum > 0"
            omzp[lane] = (wmz > 0) ? vmz / wmz : 0.0;
        }
    } else {
    // template (O, S, F) (quadrupole.py:304-324)
    for (int c = lane; c < OSF; c += ADH_WAVE) {
        int o = c / SF, sf = c - o * SF, sc = sf / F;
        double acc = 0;
        for (int i = 0; i < I; ++i) {
            float a = prec_int(i, sf) * iso_int[i];
            acc += (double)a * qtf[(i * O + o) * S + sc];
        }
        tpl[c] = (float)acc;
    }
    }
    adh_wave_sync();

    if (caps.stop_phase == 1) return;  // developer ablation switches (ADH_DEBUG_IM)
    // ---- observation importance (quadrupole.py:327-335), fragment presence (candidate.py:319-329)
    float *const rowsum = Fl + lay.f_pko();
    float *const fw = rowsum + Kc * Oc;
    float *const ftc = fw + Kc * Oc;
    float *const mfw = ftc + Kc * Oc;
    // np.sum(np.sum(template, axis=-1), axis=-1): the inner sums (one per scan) are independent, the
    // outer one adds them in scan order
    // (the per-scan sums are the template's scan profile: tsp_raw keeps them)
    double *const esc = D + lay.d_po();
    double *const efc = esc + Oc;
    double *const omz = D + lay.d_omz();
    double *const ohe = D + lay.d_ohe();
    double *const omz_u = D + lay.d_omzu();
    double *const ohe_u = D + lay.d_oheu();
    if (sparse_tpl) {
        for (int c = lane; c < S; c += ADH_WAVE) tsp_raw[c] = 0.0f;
        for (int c = lane; c < F; c += ADH_WAVE) tfp_raw[c] = 0.0f;
        adh_wave_sync();
        if (lane == 0) {
            // the list is in (scan, cycle) order: a scan's cells are consecutive and in cycle order (its sum =
            // np.sum over the cycle axis), a cycle's cells come in scan order (the sum over the scan axis), and
            // the centre-of-mass sums take the cells v > 0 in list order (fragment_features.py:20-68)
            double isum = 0.0, ssum = 0.0, fsum = 0.0;
            float srow = 0.0f;
            int cur = -1;
            for (int e = 0; e < tl_n; ++e) {
                const int cs = tl_cell[e], sc = cs >> 12, f = cs & 0xFFF;
                const float v = tl_val[e];
                if (sc != cur) {
                    if (cur >= 0) tsp_raw[cur] = srow;
                    srow = 0.0f;
                    cur = sc;
                }
                srow += v;
                tfp_raw[f] = tfp_raw[f] + v;
                if (v > 0.0f) {
                    isum += (double)v;
                    ssum += (double)sc * (double)v;
                    fsum += (double)f * (double)v;
                }
            }
            if (cur >= 0) tsp_raw[cur] = srow;
            esc[0] = (isum > 0) ? ssum / isum : 0.0;
            efc[0] = (isum > 0) ? fsum / isum : 0.0;
        }
        adh_wave_sync();
        if (lane < O) tsum[lane] = osum<SR>(tsp_raw + lane * S, 1, S);
        adh_wave_sync();
    } else {
    for (int c = lane; c < O * S; c += ADH_WAVE) tsp_raw[c] = osum<FR>(tpl + c * F, 1, F);
    adh_wave_sync();
    if (lane < O) tsum[lane] = osum<SR>(tsp_raw + lane * S, 1, S);
    adh_wave_sync();
    // ---- template centre of mass and the weight tables (fragment_features.py:20-68,
    // features_utils.py:9-25): they only depend on the precursor tile and are needed by the pass
    // over the fragment tile
    // sequential float64 sums over the non-zero template cells (s outer, f inner); the terms are
    // computed by all lanes, compacted in order, and lanes 0..2 add up one sum each
    for (int o = 0; o < O; ++o) {
        double acc = 0.0;  // lane 0: sum v, lane 1: sum scan * v, lane 2: sum frame * v
        for (int base = 0; base < SF; base += ADH_WAVE) {
            const int ci = base + lane;
            const float v = (ci < SF) ? tpl[o * SF + ci] : 0.0f;
            const bool nz = v > 0.0f;
            const unsigned long long mask = __ballot(nz);
            if (mask == 0ull) continue;
            if (nz) {
                const int sc = ci / F, f = ci - sc * F;
                const int pos = __popcll(mask & ((1ull << lane) - 1ull));
                l_w[pos] = (double)v;
                l_ti[pos] = (double)sc * (double)v;
                l_tm[pos] = (double)f * (double)v;
            }
            adh_wave_sync();
            const int n_ent = __popcll(mask);
            if (lane < 3) {
                const double *src = lane == 0 ? l_w : (lane == 1 ? l_ti : l_tm);
                for (int e = 0; e < n_ent; ++e) acc += src[e];
            }
            adh_wave_sync();
        }
        const double isum = __shfl(acc, 0), ssum = __shfl(acc, 1), fsum = __shfl(acc, 2);
        if (lane == 0) {
            esc[o] = (isum > 0) ? ssum / isum : 0.0;
            efc[o] = (isum > 0) ? fsum / isum : 0.0;
        }
    }
    }
    for (int c = lane; c < K0 * O * F; c += ADH_WAVE) ffp_u[c] = 0.0f;
    for (int c = lane; c < K0 * O * S; c += ADH_WAVE) fsp_u[c] = 0.0f;
    adh_wave_sync();
    if (caps.stop_phase == 2) return;
    // ---- ONE pass over the fragment tile [k][o][s][f]:
    //   scan profile  fsp[s] = sum_f x[s][f]            (scoring/utils.py:56-66 input)
    //   frame profile ffp[f] = sum_s x[s][f]            (scoring/utils.py:26-53 input)
    //   weighted centre means of both channels         (features_utils.py:9-37)
    // Every (fragment, observation) plane has its own lane, which folds the plane's cells in the
    // reference's order (scan outer, cycle inner) with the running sums in registers.  A round stages
    // QT = ADH_IM_STAGE / planes cells of every plane in the chunk lists (two per lane: the weights
    // exp(-0.1 * distance to the template centre) are computed by all 64 lanes, and the loads of the
    // next round are in flight meanwhile); lane l then folds the QT staged cells of plane l.  The planes
    // advance in parallel, so a signal-rich candidate costs (cells of its fullest plane) / QT rounds.
    // Ion-mobility tiles are sparse and adding a zero leaves every one of these sums unchanged: normally
    // the gather kernel delivers the non-zero cells only, as (cell, intensity, m/z) entries sorted by cell,
    // and a plane's cells are a range of that list; a candidate whose tile had to be materialised
    // (ADH_IM_MODE_DENSE) walks all cells of its planes the same way.
    {
        double *const acc_vi = ohe_u, *const acc_vm = omz_u;   // value sums; turned into the means below
        double *const acc_wi = D + lay.d_accw(), *const acc_wm = acc_wi + Kc * Oc;
        int *const pl_beg = fpeak;                              // [K0 * O], idle until the peak search
        int *const pl_end = reinterpret_cast<int *>(rowsum);    // [K0 * O], idle until the pass is over
        const int n_pl = K0 * O;
        if (compact) {
            for (int c = lane; c < n_pl; c += ADH_WAVE) {
                pl_beg[c] = 0;
                pl_end[c] = 0;
            }
            adh_wave_sync();
            for (int e = lane; e < n_fe; e += ADH_WAVE) {
                const int pc = (int)entries[e].cell / SF;
                const int pp = e > 0 ? (int)entries[e - 1].cell / SF : -1;
                if (pc != pp) {
                    pl_beg[pc] = e;
                    if (pp >= 0) pl_end[pp] = e;
                }
                if (e == n_fe - 1) pl_end[pc] = n_fe;
            }
        } else {
            for (int c = lane; c < n_pl; c += ADH_WAVE) {
                pl_beg[c] = c * SF;
                pl_end[c] = (c + 1) * SF;
            }
        }
        adh_wave_sync();
        // cell e of the list (sparse form) or of the tile (dense form)
        auto fetch = [&](int e) -> ImEntry {
            if (compact) return entries[e];
            const float2 v = fcells[e];
            ImEntry en;
            en.cell = (uint32_t)e, en.x = v.x, en.y = v.y;
            return en;
        };
        double *const s_w = reinterpret_cast<double *>(pool);
        int *const s_c = reinterpret_cast<int *>(s_w + ADH_IM_STAGE);
        float *const s_v = reinterpret_cast<float *>(s_c + ADH_IM_STAGE);
        float *const s_y = s_v + ADH_IM_STAGE;
        const double inv_f = 1.0 / (double)F;
        for (int pbase = 0; pbase < n_pl; pbase += ADH_WAVE) {
            const int np = min(ADH_WAVE, n_pl - pbase);
            const int QT = ADH_IM_STAGE / np;
            const bool fl = lane < np;
            const int p = pbase + lane;
            const int beg_f = fl ? pl_beg[p] : 0, end_f = fl ? pl_end[p] : 0;
            int R = (end_f - beg_f + QT - 1) / QT;
            for (int off = 32; off > 0; off >>= 1) R = max(R, __shfl_xor(R, off));
            int t_beg[ADH_IM_STAGE / ADH_WAVE], t_end[ADH_IM_STAGE / ADH_WAVE], t_j[ADH_IM_STAGE / ADH_WAVE], t_o[ADH_IM_STAGE / ADH_WAVE],
                t_psf[ADH_IM_STAGE / ADH_WAVE];
            ImEntry nxt[ADH_IM_STAGE / ADH_WAVE];
#pragma unroll
            for (int t = 0; t < ADH_IM_STAGE / ADH_WAVE; ++t) {
                const int slot = lane + t * ADH_WAVE;
                const int ps = slot / QT;
                const bool okp = ps < np;
                const int pp = pbase + (okp ? ps : 0);
                t_j[t] = slot - ps * QT;
                t_beg[t] = okp ? pl_beg[pp] : 0;
                t_end[t] = okp ? pl_end[pp] : 0;
                t_o[t] = pp % O;
                t_psf[t] = pp * SF;
                nxt[t].cell = 0u, nxt[t].x = 0.0f, nxt[t].y = 0.0f;
                if (t_beg[t] + t_j[t] < t_end[t]) nxt[t] = fetch(t_beg[t] + t_j[t]);
            }
            double vi = 0.0, wi = 0.0, vm = 0.0, wm = 0.0;
            float fs = 0.0f;
            int cur_sc = -1;
            for (int rr = 0; rr < R; ++rr) {
                ImEntry cur[ADH_IM_STAGE / ADH_WAVE];
#pragma unroll
                for (int t = 0; t < ADH_IM_STAGE / ADH_WAVE; ++t) {
                    cur[t] = nxt[t];
                    const int e2 = t_beg[t] + (rr + 1) * QT + t_j[t];
                    if (e2 < t_end[t]) nxt[t] = fetch(e2);
                }
#pragma unroll
                for (int t = 0; t < ADH_IM_STAGE / ADH_WAVE; ++t) {
                    const int e = t_beg[t] + rr * QT + t_j[t];
                    if (e < t_end[t]) {
                        const int rem = (int)cur[t].cell - t_psf[t];
                        int sc = (int)((double)rem * inv_f);  // exact quotient: float64 estimate, one fix-up
                        if (rem - sc * F >= F) ++sc;
                        const int f = rem - sc * F, o = t_o[t];
                        const int slot = lane + t * ADH_WAVE;
                        double w = 1.0;
                        if (cur[t].x > 0.0f || cur[t].y > 0.0f) {  // (an empty cell of a dense tile adds 0 whatever its weight)
                            const double ds = (double)sc - esc[o], df = (double)f - efc[o];
                            w = exp(-0.1 * sqrt(ds * ds + df * df));
                        }
                        s_c[slot] = sc << 16 | f;  // (the serial fold below must not divide)
                        s_v[slot] = cur[t].x * qmask[o * S + sc];  // candidate.py:290
                        s_y[slot] = cur[t].y;
                        s_w[slot] = w;
                    }
                }
                adh_wave_sync();
                if (fl) {
                    const int cnt = min(QT, end_f - (beg_f + rr * QT));
                    for (int t = 0; t < cnt; ++t) {
                        const int at = lane * QT + t;
                        const int cf = s_c[at];
                        const float v = s_v[at], y = s_y[at];
                        const double w = s_w[at];
                        const int sc = cf >> 16, f = cf & 0xFFFF;
                        const float col = ffp_u[p * F + f];
                        if (sc != cur_sc) {  // the cells of a scan are consecutive: its sum is complete
                            if (cur_sc >= 0) fsp_u[p * S + cur_sc] = fs;
                            fs = 0.0f;
                            cur_sc = sc;
                        }
                        fs += v;
                        ffp_u[p * F + f] = col + v;
                        const double tm = (double)y * w;  // (w > 0: the product is > 0 exactly when the m/z channel is)
                        vi += v > 0.0f ? (double)v * w : 0.0;  // (adding 0.0 leaves a sum as it is)
                        wi += v > 0.0f ? w : 0.0;
                        vm += tm > 0.0 ? tm : 0.0;
                        wm += tm > 0.0 ? w : 0.0;
                    }
                }
                adh_wave_sync();
            }
            if (fl) {
                if (cur_sc >= 0) fsp_u[p * S + cur_sc] = fs;
                acc_vi[p] = vi;
                acc_wi[p] = wi;
                acc_vm[p] = vm;
                acc_wm[p] = wm;
            }
        }
        adh_wave_sync();
        for (int c = lane; c < K0 * O; c += ADH_WAVE) {
            const double vi = acc_vi[c], wi = acc_wi[c], vm = acc_vm[c], wm = acc_wm[c];
            ohe_u[c] = (wi > 0) ? vi / wi : 0.0;  // weights are exp(...) > 0: "any non-zero cell" == "wi > 0"
            omz_u[c] = (wm > 0) ? vm / wm : 0.0;
        }
    }
    adh_wave_sync();
    if (caps.stop_phase == 3) return;
    // Dense-mode candidates (rare: a tile that had to be materialised): the isotope sums and centre means the
    // sparse form gets from its precursor pass.  Independent of everything between here and its call.
    auto dense_precursor_sums = [&]() {

    for (int c = lane; c < I * S; c += ADH_WAVE) {
        float sf = 0;
        for (int f = 0; f < F; ++f) sf += prec_int(c / S, (c % S) * F + f);
        work_b[c] = sf;
    }
    adh_wave_sync();
    if (lane < I) {
        float ss = 0;
        for (int sc = 0; sc < S; ++sc) ss += work_b[lane * S + sc];
        spi[lane] = ss;
    }
    // weighted centre means of the precursor planes around (scan, frame) = (S, 1)
    // (precursor_features.py:52-66): float64 sums over the non-zero cells in (scan, frame) order.  The
    // 64 lanes look at 64 cells at once, compact the non-zero ones in order into the chunk lists and
    // lanes 0..3 fold one of the four sums each (intensity plane: values, weights; m/z plane: the same).
    for (int i = 0; i < I; ++i) {
        double acc = 0.0;
        for (int base = 0; base < SF; base += ADH_WAVE) {
            const int ci = base + lane;
            const float2 pc = (ci < SF) ? prec_cell(i, ci) : make_float2(0.0f, 0.0f);
            const float vi = pc.x, vm = pc.y;
            const bool nz = vi > 0.0f || vm > 0.0f;
            const unsigned long long mask = __ballot(nz);
            if (mask == 0ull) continue;
            if (nz) {
                const int sc = ci / F, f = ci - sc * F;
                const double ds = (double)(sc - S), df = (double)(f - 1);
                const double w = exp(-0.1 * sqrt(ds * ds + df * df));
                const int pos = __popcll(mask & ((1ull << lane) - 1ull));
                l_w[pos] = w;
                l_ti[pos] = (double)vi * w;
                l_tm[pos] = (double)vm * w;
                l_rx[pos] = vi;
                l_ry[pos] = vm;
            }
            adh_wave_sync();
            const int n_ent = __popcll(mask);
            if (lane < 4) {
                const float *flag = (lane < 2) ? l_rx : l_ry;
                const double *src = (lane & 1) ? l_w : (lane == 0 ? l_ti : l_tm);
                for (int e = 0; e < n_ent; ++e)
                    if (flag[e] > 0.0f) acc += src[e];
            }
            adh_wave_sync();
        }
        const double vh = __shfl(acc, 0), wh = __shfl(acc, 1), vmz = __shfl(acc, 2), wmz = __shfl(acc, 3);
        if (lane == 0) {
            hp[i] = (wh > 0) ? vh / wh : 0.0;      // weights are exp(...) > 0: "any non-zero cell" == "w sum > 0"
            omzp[i] = (wmz > 0) ? vmz / wmz : 0.0;
        }
    }
        };
    if constexpr (SPLIT) {
        if (!compact) {
            dense_precursor_sums();
            adh_wave_sync();
        }
        // ---- hand-over: profiles, template profiles and the per-plane means of this candidate
        static_assert(LAY::Oc <= 2 && LAY::Kc <= ADH_IM_PROF_K && LAY::Ic <= 4, "the split path takes one or two observations");
        constexpr int NOc = LAY::Oc;
        typedef ImProfRec<LAY::Fc, LAY::Sc, NOc> Rec;
        Rec &rec = reinterpret_cast<Rec *>(prof)[ci];
        constexpr int FMc = LAY::Fc, SMc = LAY::Sc;
        const int shift = F / 2 - FMc / 2;  // entry r <-> cycle r + shift
        // template frame profiles: sums over the scans, in scan order (the monolithic kernel takes them below);
        // observations beyond the candidate's own (a launch of up to two) are zero rows
        for (int c = lane; c < NOc * FMc; c += ADH_WAVE) {
            const int o = c / FMc, rr = c - o * FMc, f = rr + shift;
            const bool on = o < O && f >= 0 && f < F;
            rec.tfp_raw[o][rr] = on ? (sparse_tpl ? tfp_raw[f] : osum<SR>(tpl + o * SF + f, F, S)) : 0.0f;
        }
        for (int c = lane; c < NOc * SMc; c += ADH_WAVE) {
            const int o = c / SMc, sc = c - o * SMc;
            rec.tsp_raw[o][sc] = (o < O && sc < S) ? tsp_raw[o * S + sc] : 0.0f;
        }
        for (int c = lane; c < K0 * NOc * FMc; c += ADH_WAVE) {
            const int k = c / (NOc * FMc), rem = c - k * NOc * FMc, o = rem / FMc, rr = rem - o * FMc, f = rr + shift;
            rec.ffp[k][o][rr] = (o < O && f >= 0 && f < F) ? ffp_u[(k * O + o) * F + f] : 0.0f;
        }
        for (int c = lane; c < K0 * NOc * SMc; c += ADH_WAVE) {
            const int k = c / (NOc * SMc), rem = c - k * NOc * SMc, o = rem / SMc, sc = rem - o * SMc;
            rec.fsp[k][o][sc] = (o < O && sc < S) ? fsp_u[(k * O + o) * S + sc] : 0.0f;
        }
        for (int c = lane; c < K0 * NOc; c += ADH_WAVE) {
            const int k = c / NOc, o = c - k * NOc;
            rec.ohe[k][o] = o < O ? ohe_u[k * O + o] : 0.0;
            rec.omz[k][o] = o < O ? omz_u[k * O + o] : 0.0;
        }
        if (lane < 4) {
            const bool on = lane < I;
            rec.hp[lane] = on ? hp[lane] : 0.0;
            rec.omzp[lane] = on ? omzp[lane] : 0.0;
            rec.spi[lane] = on ? spi[lane] : 0.0f;
            rec.iso_int[lane] = on ? iso_int[lane] : 0.0f;
            rec.iso_mz[lane] = on ? iso_mz[lane] : 0.0f;
        }
        if (lane < NOc) rec.tsum[lane] = lane < O ? tsum[lane] : 0.0f;
        if (lane == 0) rec.K0 = (uint32_t)K0;
        return;
    }
    for (int k = lane; k < K0; k += ADH_WAVE) {
        float so = 0;
        for (int o = 0; o < O; ++o) {
            const float ss = osum<SR>(fsp_u + (k * O + o) * S, 1, S);  // sum of the per-scan sums
            rowsum[k * O + o] = ss;
            so += ss;
        }
        present[k] = so > 0.0f;
    }
    adh_wave_sync();
    {
        float tot = 0;
        for (int o = 0; o < O; ++o) tot += tsum[o];
        if (lane < O) oi[lane] = (tot == 0.0f) ? 1.0f / (float)O : tsum[lane] / tot;
    }
    int K = 0;
    for (int k = 0; k < K0; ++k) {
        if (present[k]) {
            if (lane == 0) kmap[K] = k;
            ++K;
        }
    }
    if (K < 2) return;  // candidate.py:323
    const int n_present = K;
    adh_wave_sync();

    float *const g_mzlib = Fl + lay.f_pk();
    float *const g_mz = g_mzlib + Kc;
    float *const g_int = g_mz + Kc;
    float *const g_fin = g_int + Kc;
    float *const obs_int = g_fin + Kc;
    float *const corr = obs_int + Kc;
    float *const mnorm = corr + Kc;
    float *const mlist = mnorm + Kc;
    uint8_t *const g_type = By;
    uint8_t *const g_loss = g_type + Kc;
    uint8_t *const g_charge = g_loss + Kc;
    uint8_t *const g_number = g_charge + Kc;
    uint8_t *const g_pos = g_number + Kc;
    {
        const LibRec *sel = reinterpret_cast<const LibRec *>(block + 32);
        for (int k = lane; k < K; k += ADH_WAVE) {
            LibRec rec = sel[kmap[k]];
            g_mzlib[k] = rec.mz_library;
            g_mz[k] = rec.mz;
            g_int[k] = rec.intensity;
            g_type[k] = rec.type;
            g_loss[k] = rec.loss_type;
            g_charge[k] = rec.charge;
            g_number[k] = rec.number;
            g_pos[k] = rec.position;
        }
    }
    adh_wave_sync();
    {
        float sum1 = 0;
        for (int k = 0; k < K; ++k) sum1 += g_int[k];
        adh_wave_sync();
        for (int k = lane; k < K; k += ADH_WAVE) g_int[k] = g_int[k] / sum1;
        adh_wave_sync();
        float sum2 = 0;
        for (int k = 0; k < K; ++k) sum2 += g_int[k];
        for (int k = lane; k < K; k += ADH_WAVE) g_fin[k] = g_int[k] / sum2;
    }

    // ---- profiles (candidate.py:333-347; scoring/utils.py:26-66); k is the compacted index.
    // The template's profiles first: the masked fragment profiles then take the template's place.
    for (int c = lane; c < O * F; c += ADH_WAVE) {
        int o = c / F, f = c - o * F;
        tfp_raw[c] = osum<SR>(tpl + o * S * F + f, F, S);
    }
    // (tsp_raw: the per-scan sums taken for the observation importance above)
    adh_wave_sync();
    for (int c = lane; c < K * O * F; c += ADH_WAVE) {
        int k = c / (O * F), rem = c - k * O * F;
        ffp[c] = ffp_u[kmap[k] * O * F + rem];
    }
    for (int c = lane; c < K * O * S; c += ADH_WAVE) {
        int k = c / (O * S), rem = c - k * O * S;
        fsp_raw[c] = fsp_u[kmap[k] * O * S + rem];
    }
    for (int c = lane; c < K * O; c += ADH_WAVE) {
        int k = c / O, o = c - k * O;
        ohe[c] = ohe_u[kmap[k] * O + o];
        omz[c] = omz_u[kmap[k] * O + o];
    }
    for (int f = lane; f < F; f += ADH_WAVE) frame_rt[f] = run.rt[r.frame_start + f * L];
    adh_wave_sync();
    // OR-envelopes: interior points lower than a neighbour become the neighbours' mean
    for (int c = lane; c < O * F; c += ADH_WAVE) {
        int f = c % F;
        float x = tfp_raw[c], rr = x;
        if (f >= 1 && f < F - 1 && (x < tfp_raw[c - 1] || x < tfp_raw[c + 1])) {
            float sm = tfp_raw[c - 1] + tfp_raw[c + 1];
            rr = (float)((double)sm / 2.0);
        }
        tfp[c] = rr;
    }
    for (int c = lane; c < K * O * S; c += ADH_WAVE) {
        int sc = c % S;
        float x = fsp_raw[c], rr = x;
        if (sc >= 1 && sc < S - 1 && (x < fsp_raw[c - 1] || x < fsp_raw[c + 1])) {
            float sm = fsp_raw[c - 1] + fsp_raw[c + 1];
            rr = (float)((double)sm / 2.0);
        }
        fsp[c] = rr;
    }
    for (int c = lane; c < O * S; c += ADH_WAVE) {
        int sc = c % S;
        float x = tsp_raw[c], rr = x;
        if (sc >= 1 && sc < S - 1 && (x < tsp_raw[c - 1] || x < tsp_raw[c + 1])) {
            float sm = tsp_raw[c - 1] + tsp_raw[c + 1];
            rr = (float)((double)sm / 2.0);
        }
        tsp[c] = rr;
    }
    adh_wave_sync();

    if (caps.stop_phase == 4) return;
    // =========================== features ===========================
    // isotope intensity sums / centre means of a dense-mode candidate (sparse form: done with the template above)
    if (!compact) dense_precursor_sums();
    adh_wave_sync();

    double *const mzmean = D + lay.d_pk();
    double *const height = mzmean + Kc;
    double *const area = height + Kc;
    double *const merr = area + Kc;
    int best_obs = 0;
    if (!cfg.quant_all)
        for (int o = 1; o < O; ++o)
            if (oi[o] > oi[best_obs]) best_obs = o;
    for (int k = lane; k < K; k += ADH_WAVE) {
        float *x;
        if (cfg.quant_all) {
            x = bp + k * F;
            for (int f = 0; f < F; ++f) {
                float a = 0;
                for (int o = 0; o < O; ++o) a += ffp[(k * O + o) * F + f];
                x[f] = a;
            }
        } else {
            x = ffp + (k * O + best_obs) * F;
        }
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
            for (int f = 0; f < F; ++f) bp[k * F + f] = x[f];
        int qw = min(F / 2 - 1, (int)cfg.quant_window);
        int center = F / 2;
        int a, b;
        py_slice(center - qw, center + qw + 1, F, a, b);
        const float *p = bp + k * F + a;
        int W = b - a;
        double ar = 0;
        for (int i = 0; i + 1 < W; ++i) {
            float sm = p[i + 1] + p[i];
            double drt = frame_rt[a + i + 1] - frame_rt[a + i];  // float64 rt_values
            ar += (double)sm * drt * 0.5;                         // f32 * f64 -> f64
        }
        area[k] = ar * (double)qw;
        float t = 0;
        for (int i = 0; i < W; ++i) t += p[i];
        obs_int[k] = t;
    }
    adh_wave_sync();
    if (caps.stop_phase == 45) return;
    for (int k = lane; k < K; k += ADH_WAVE) {
        float ws = 0;
        for (int o = 0; o < O; ++o) {
            bool m = ohe[k * O + o] > 0;
            float w32 = m ? oi[o] : oi[o] * 0.0f;
            ws += w32;
        }
        double msum = 0;
        int nm = 0;
        for (int o = 0; o < O; ++o) {
            bool m = ohe[k * O + o] > 0;
            float w32 = m ? oi[o] : oi[o] * 0.0f;
            double w = (double)w32 / ((double)ws + 1e-20);
            if (w > 0) {
                msum += w;
                ++nm;
            }
        }
        double m1 = 0, m2 = 0;
        if (nm > 0)
            for (int o = 0; o < O; ++o) {
                bool m = ohe[k * O + o] > 0;
                float w32 = m ? oi[o] : oi[o] * 0.0f;
                double w = (double)w32 / ((double)ws + 1e-20);
                if (w > 0) {
                    double lw = w / msum;
                    m1 += omz[k * O + o] * lw;
                    m2 += ohe[k * O + o] * lw;
                }
            }
        mzmean[k] = m1;
        height[k] = m2;
        merr[k] = (m1 - (double)g_mz[k]) / (double)g_mz[k] * 1e6;
        int rk = 0;
        float ia = g_int[k];
        for (int b = 0; b < K; ++b) {
            float ib = g_int[b];
            rk += (ib > ia) || (ib == ia && b > k);
        }
        ord[rk] = k;
    }
    adh_wave_sync();
    if (caps.stop_phase == 46) return;

    Assemble asmv;
    asmv.run = nullptr;  // location features are float64 here, filled below
    asmv.rec = nullptr;
    asmv.featv = featv;
    asmv.iso_int = iso_int; asmv.iso_mz = iso_mz; asmv.spi = spi; asmv.oi = oi; asmv.tsum = tsum;
    asmv.rowsum = rowsum; asmv.g_fin = g_fin; asmv.g_int = g_int; asmv.obs_int = obs_int;
    asmv.corr = corr; asmv.ftc = ftc; asmv.fw = fw; asmv.medpk = medpk;
    asmv.omzp = omzp; asmv.hp = hp; asmv.ohe = ohe; asmv.area = area; asmv.height = height;
    asmv.merr = merr; asmv.kmap = kmap; asmv.ord = ord; asmv.g_type = g_type; asmv.g_pos = g_pos;
    asmv.n_present = n_present; asmv.K0 = K0; asmv.top3 = 0.0f;
    // Feature assembly.  The loop form (feat::assemble_*) is ~45 short sequential sums over fragments on one
    // lane, every term an LDS load away: a third of a millisecond per 60 000 candidates.  Up to 16 fragments:
    // lane k provides its term of every sum (a skipped term is +0, which leaves a sum as it is), lane j adds
    // up sum j in fragment order from registers and finishes the features that hang on it.
    const bool lanes_ok = K <= KR && I <= 3 && O <= 4 && caps.stop_phase != 21;
    double(*const t64)[6] = reinterpret_cast<double(*)[6]>(pool);                   // [KR][6]; the chunk lists are idle
    float(*const t32)[8] = reinterpret_cast<float(*)[8]>(pool + KR * 6 * 8);        // [KR][8]
    double *const red64 = reinterpret_cast<double *>(pool + KR * 6 * 8 + KR * 8 * 4);  // [16]
    static_assert(KR * 6 * 8 + KR * 8 * 4 + 16 * 8 <= ADH_IM_STATIC_LDS, "term matrices fit the chunk lists");
    if (lanes_ok && caps.stop_phase != 20) {
        const bool kl = lane < K;
        const int k = kl ? lane : 0;
        const double area_k = area[k], m2 = height[k], merr_k = merr[k];
        const float gfin = g_fin[k], oint = obs_int[k];
        const int type = g_type[k], pos = g_pos[k];
        bool hrow = false;
        for (int o = 0; o < O; ++o) hrow = hrow || ohe[k * O + o] > 0;
        const bool ipos = kl && oint > 0.0f, hpos = kl && m2 > 0.0;
        const bool isb = kl && type == 98, isy = kl && type == 121;
        const int n_int = __popcll(__ballot(ipos)), n_hei = __popcll(__ballot(hpos));
        const int n_hrows = __popcll(__ballot(kl && hrow));
        const int nb = __popcll(__ballot(isb)), ny = __popcll(__ballot(isy));
        int min_y = isy ? pos : 255, max_b = isb ? pos : 0;
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) {  // (fragments sit in lanes 0..15)
            min_y = min(min_y, __shfl_xor(min_y, m));
            max_b = max(max_b, __shfl_xor(max_b, m));
        }
        min_y = __shfl(min_y, 0);
        max_b = __shfl(max_b, 0);
        const bool ov = (isy && pos < max_b) || (isb && pos > min_y);
        const int n_ov = __popcll(__ballot(ov));
        const int n3 = min(K, 3);
        if (kl) {
            double *t = t64[k];
            t[0] = area_k;
            t[1] = m2;
            t[2] = (double)gfin;
            t[3] = merr_k;
            t[4] = ov ? area_k : 0.0;
            t[5] = ov ? merr_k : 0.0;
            // cosine_similarity_a1 (features_utils.py:40-47) of the observation sums
            float tn = 0.0f, fn = 0.0f, dot = 0.0f;
            const float *rs = rowsum + kmap[k] * O;
            for (int o = 0; o < O; ++o) tn += tsum[o] * tsum[o];
            tn = sqrtf(tn);
            for (int o = 0; o < O; ++o) fn += rs[o] * rs[o];
            fn = sqrtf(fn);
            for (int o = 0; o < O; ++o) dot += rs[o] * tsum[o];
            const float pr = fn * tn;
            const float score = (float)((double)dot / ((double)pr + 0.0001));
            float *u = t32[k];
            u[0] = ipos ? gfin : 0.0f;
            u[1] = hpos ? gfin : 0.0f;
            u[2] = ipos ? score : 0.0f;
            u[3] = isb ? oint : 0.0f;
            u[4] = isy ? oint : 0.0f;
        }
        adh_wave_sync();
        {
            // lane j < 6: float64 sum j; 6 <= j < 11: float32 sum j - 6; lane 11: mean_top3 mass error, by rank
            double s64 = 0.0;
            float s32 = 0.0f;
            {
                double v64[KR];
                float v32[KR];
                const int c64 = min(lane, 5), c32 = min(max(lane - 6, 0), 4);
#pragma unroll
                for (int j = 0; j < KR; ++j) {
                    v64[j] = t64[j][c64];
                    v32[j] = t32[j][c32];
                }
                if (lane == 11) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) v64[i] = merr[i < n3 ? ord[i] : 0];
                }
                const int n64 = lane == 11 ? n3 : K;
#pragma unroll
                for (int j = 0; j < KR; ++j) {
                    s64 += (j < n64) ? v64[j] : 0.0;
                    s32 += (j < K) ? v32[j] : 0.0f;
                }
            }
            // the quotients: lane -> (numerator, denominator)
            //   0-2 sums of area / height / intensity over K (np.corrcoef means), 3 mass error / K (42),
            //   4, 5 overlap area / mass error over n_ov (44, 45), 6 n_int / K (20), 7 n_hei / K (21),
            //   8 cosine sum / n_int (24), 11 top-3 mass error / n3 (41), 15 n_present / K0 (28, candidate.py:362)
            double num = s64, den = (double)K;
            if (lane == 4 || lane == 5) den = (double)n_ov;
            if (lane == 6) num = (double)n_int;
            if (lane == 7) num = (double)n_hei;
            if (lane == 8) num = (double)s32, den = (double)n_int;
            if (lane == 11) den = (double)n3;
            if (lane == 15) num = (double)n_present, den = (double)K0;
            const double quo = num / den;
            if (lane < 3) red64[lane] = quo;
            if (lane == 1) red64[3] = s64;  // (the sum of the heights decides whether feature 19 is taken)
            float *ft = featv;
            if (lane == 3) ft[42] = (float)quo;
            if ((lane == 4 || lane == 5) && nb > 0 && ny > 0) {
                if (lane == 4) ft[43] = (float)n_ov;
                ft[40 + lane] = n_ov > 0 ? (float)quo : (lane == 4 ? 0.0f : 15.0f);
            }
            if (lane == 6) ft[22] = s32, ft[20] = (float)quo;
            if (lane == 7) ft[23] = s32, ft[21] = (float)quo;
            if (lane == 8 && n_int > 0) ft[24] = (float)quo;
            if (lane == 9 || lane == 10) {
                const float lg = (float)log((double)s32 + 1.0);
                ft[16 + lane] = ((lane == 9 ? nb : ny) > 0) ? lg : 0.0f;
            }
            if (lane == 11) ft[41] = (float)quo;
            if (lane == 15) ft[28] = (float)quo, ft[17] = (float)O;
        }
        adh_wave_sync();
        {
            // np.corrcoef terms (feat::corrcoef01): area vs intensity, height vs intensity
            const double mx_a = red64[0], mx_h = red64[1], my = red64[2];
            if (kl) {
                const double a = area_k - mx_a, h = m2 - mx_h, b = (double)gfin - my;
                double *t = t64[k];
                t[0] = a * a;
                t[1] = b * b;
                t[2] = a * b;
                t[3] = h * h;
                t[4] = h * b;
            }
        }
        adh_wave_sync();
        {
            double v64[KR], s64 = 0.0;
            const int c64 = min(lane, 4);
#pragma unroll
            for (int j = 0; j < KR; ++j) v64[j] = t64[j][c64];
#pragma unroll
            for (int j = 0; j < KR; ++j) s64 += (j < K) ? v64[j] : 0.0;
            if (lane < 5) red64[7 + lane] = s64;
        }
        adh_wave_sync();
        if (lane < 2) {
            // lane 0: feature 18 (areas), lane 1: feature 19 (heights)
            const double fact = fmax((double)K - 1.0, 0.0);
            const double inv = 1.0 / fact;
            const double cxx = red64[lane ? 10 : 7] * inv, cyy = red64[8] * inv;
            const double cxy = red64[lane ? 11 : 9] * inv;
            const double s0 = sqrt(cxx), s1 = sqrt(cyy);
            double cc = cxy / s1 / s0;
            if (fabs(cc) > 1.0) cc = (cc > 0) ? 1.0 : -1.0;
            const bool on = lane ? (red64[3] > 0.0) : (n_hrows > 0);
            if (on) featv[18 + lane] = (float)cc;
        }
        if (lane == 2) featv[27] = featv[25] - featv[26];
        if (lane == 3) precursor_features_rows(featv, I, O, iso_int, iso_mz, spi, hp, omzp, oi);
    }
    if (lane == 0) {
        if (caps.stop_phase != 20 && !lanes_ok) feat::assemble_part1(asmv, I, O, K);  // (20: developer ablation, no assembly)
        // location_features.py:8-33 with float64 mobility / rt arrays
        featv[0] = (float)(run.mobility[r.scan_start] - run.mobility[r.scan_stop - 1]);
        featv[1] = (float)(run.rt[r.frame_stop - 1] - run.rt[r.frame_start]);
        featv[2] = (float)run.rt[r.frame_center];
        featv[3] = (float)run.mobility[r.scan_center];
    }

    if (caps.stop_phase == 5) return;
    // =========================== fragment_mobility_correlation (fragment_features.py:430-480)
    // centred scan profiles go to the second work buffer, centred frame profiles later
    adh_wave_sync();
    float *cen = work_b;
    {
        // fragments whose scan profiles hold any signal (fragment_features.py:447-452), in order
        int Km = 0;
        if (K <= ADH_WAVE) {  // one fragment per lane
            float so = 0;
            if (lane < K)
                for (int o = 0; o < O; ++o) so += osum<SR>(fsp + (lane * O + o) * S, 1, S);
            const unsigned long long keep = __ballot(lane < K && so > 0.0f);
            Km = __popcll(keep);
            if (lane < K && so > 0.0f) mkeep[__popcll(keep & ((1ull << lane) - 1ull))] = lane;
        } else {
            for (int k = 0; k < K; ++k) {
                float so = 0;
                for (int o = 0; o < O; ++o) {
                    float ss = 0;
                    for (int sc = 0; sc < S; ++sc) ss += fsp[(k * O + o) * S + sc];
                    so += ss;
                }
                if (so > 0.0f) {
                    if (lane == 0) mkeep[Km] = k;
                    ++Km;
                }
            }
        }
        adh_wave_sync();
        const bool rows = S <= SR;  // scan rows fit the registers
        if (Km >= 3) {
            float isum = 0;
            for (int a = 0; a < Km; ++a) isum += g_int[mkeep[a]];
            for (int a = lane; a < Km; a += ADH_WAVE) mnorm[a] = g_int[mkeep[a]] / isum;
            // centred rows + std per (a, o) over the scan axis (scoring/utils.py:545-559)
            if (rows) {
                // ... and, with the centred row still in registers, its correlation with the template's
                // scan profile (scoring/utils.py:574-647)
                for (int c = lane; c < Km * O; c += ADH_WAVE) {
                    const int a = c / O, o = c - a * O;
                    float pr[SR], py[SR];
                    load_row<SR>(pr, fsp + (mkeep[a] * O + o) * S, 1, S);
                    load_row<SR>(py, tsp + o * S, 1, S);
                    float mean, sd, ym, ysd;
                    row_moments<SR>(pr, S, mean, sd);
                    row_moments<SR>(py, S, ym, ysd);
                    float dot = 0.0f;
#pragma unroll
                    for (int i = 0; i < SR; ++i) {
                        float d = pr[i] - mean;
                        d = (i < S) ? d : 0.0f;
                        if (i < S) cen[c * S + i] = d;
                        dot += d * (py[i] - ym);
                    }
                    mfw[c] = sd;
                    const float cov = dot / (float)S;
                    const float sm = sd * ysd;
                    ftc[o * Km + a] = (float)((double)cov / ((double)sm + 1e-12));
                }
            } else {
            for (int c = lane; c < Km * O; c += ADH_WAVE) {
                int a = c / O, o = c - a * O;
                const float *p = fsp + (mkeep[a] * O + o) * S;
                float sm = 0;
                for (int sc = 0; sc < S; ++sc) sm += p[sc];
                float mean = sm / (float)S;
                float q = 0;
                for (int sc = 0; sc < S; ++sc) cen[c * S + sc] = p[sc] - mean;
                for (int sc = 0; sc < S; ++sc) q += cen[c * S + sc] * cen[c * S + sc];
                mfw[c] = sqrtf(q / (float)S);
            }
            }
            adh_wave_sync();
            // np.dot(profile_centered, profile_centered.T) over the scan axis (scoring/utils.py:559, BLAS
            // SGEMM in the reference): one MFMA tile per observation, as in adh_feature_kernel; more
            // than 16 fragments use ordered dot products
            if (Km <= 16) {
                float(*gram)[17] = reinterpret_cast<float(*)[17]>(pool);  // the chunk lists are idle here
                float(*redm)[17] = gram + 16;
                for (int c = lane; c < 16 * 16; c += ADH_WAVE) redm[c / 16][c % 16] = 0.0f;
                for (int o = 0; o < O; ++o) {
                    typedef float floatx4 __attribute__((ext_vector_type(4)));
                    floatx4 d = {0.0f, 0.0f, 0.0f, 0.0f};
                    const int i = lane & 15, kq = lane >> 4;
                    // (NOT unrolled with the operands loaded ahead: hipcc 7.2 then gives an MFMA of the chain a
                    // destination that overlaps the operand register of a later one - v_mfma v[2:5], v5, v5, v[16:19]
                    // - and the tile comes out different from run to run on ~1 of 2 000 candidates)
                    for (int s0 = 0; s0 < S; s0 += 4) {
                        const int sc = s0 + kq;
                        const float v = (i < Km && sc < S) ? cen[(i * O + o) * S + sc] : 0.0f;
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(v, v, d, 0, 0, 0);
                    }
                    adh_wave_sync();
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) gram[4 * kq + rr][i] = d[rr];
                    adh_wave_sync();
                    // one (a, b) pair per lane and round: every pair has its own float64 division
                    for (int pr = lane; pr < Km * Km; pr += ADH_WAVE) {
                        const int a = pr / Km, b = pr - a * Km;
                        float cov = gram[a][b] / (float)S;
                        float sm = mfw[a * O + o] * mfw[b * O + o];
                        float cm = (float)((double)cov / ((double)sm + 1e-12));
                        redm[a][b] += cm * oi[o];
                    }

                }
                adh_wave_sync();
                for (int a = lane; a < Km; a += ADH_WAVE) {
                    float ra[KR], rb[KR], acc = 0;
#pragma unroll
                    for (int b = 0; b < KR; ++b) {
                        ra[b] = redm[a][b];
                        rb[b] = (b < Km) ? mnorm[b] : 0.0f;
                    }
#pragma unroll
                    for (int b = 0; b < KR; ++b) acc += (b < Km) ? ra[b] * rb[b] : 0.0f;
                    mlist[a] = acc;
                }
            } else {
                for (int a = lane; a < Km; a += ADH_WAVE) {
                    float acc = 0;
                    for (int b = 0; b < Km; ++b) {
                        float red = 0;
                        for (int o = 0; o < O; ++o) {
                            float dot = 0;
                            for (int sc = 0; sc < S; ++sc)
                                dot += cen[(a * O + o) * S + sc] * cen[(b * O + o) * S + sc];
                            float cov = dot / (float)S;
                            float sm = mfw[a * O + o] * mfw[b * O + o];
                            float cm = (float)((double)cov / ((double)sm + 1e-12));
                            red += cm * oi[o];
                        }
                        acc += red * mnorm[b];
                    }
                    mlist[a] = acc;
                }
            }
            // fragment vs template scan correlation (scoring/utils.py:574-647)
            if (!rows)
            for (int c = lane; c < Km * O; c += ADH_WAVE) {
                int a = c / O, o = c - a * O;
                const float *py = tsp + o * S;
                float sy = 0;
                for (int sc = 0; sc < S; ++sc) sy += py[sc];
                float ym = sy / (float)S;
                float qy = 0;
                for (int sc = 0; sc < S; ++sc) {
                    float d = py[sc] - ym;
                    qy += d * d;
                }
                float ysd = sqrtf(qy / (float)S);
                float dot = 0;
                for (int sc = 0; sc < S; ++sc) dot += cen[c * S + sc] * (py[sc] - ym);
                float cov = dot / (float)S;
                float sm = mfw[c] * ysd;
                ftc[o * Km + a] = (float)((double)cov / ((double)sm + 1e-12));
            }
            adh_wave_sync();
            if (lane == 0) {
                float lsum = 0;
                for (int a = 0; a < Km; ++a) lsum += mlist[a];
                featv[29] = (float)((double)lsum / (double)Km);
                float dot = 0;
                for (int a = 0; a < Km; ++a) {
                    float rr = 0;
                    for (int o = 0; o < O; ++o) rr += ftc[o * Km + a] * oi[o];
                    dot += rr * mnorm[a];
                }
                featv[30] = dot;
            }
        }
    }
    adh_wave_sync();

    if (caps.stop_phase == 6) return;
    // =========================== profile features (profile_features.py:18-206)
    float *isl = bp, *nrm = work_b;  // (the quantification profiles are done with; work_b is free again once the scan correlation is done)
    if (cfg.experimental_xic) {
        for (int c = lane; c < K * F; c += ADH_WAVE) {
            int k = c / F, f = c - k * F;
            float a = 0;
            for (int o = 0; o < O; ++o) a += ffp[(k * O + o) * F + f];
            isl[c] = a;
        }
        adh_wave_sync();
        int cidx = F / 2, wa, wb;
        py_slice(cidx - 1, cidx + 2, F, wa, wb);
        for (int k = lane; k < K; k += ADH_WAVE) {
            float sm = 0;
            for (int i = wa; i < wb; ++i) sm += isl[k * F + i];
            double ci = (double)sm / (double)(wb - wa);
            for (int f = 0; f < F; ++f)
                nrm[k * F + f] = (ci > 0) ? (float)((double)isl[k * F + f] / ci) : 0.0f;
        }
        adh_wave_sync();
        for (int f = lane; f < F; f += ADH_WAVE) {
            float lo_v = 0, hi_v = 0;
            int r_lo = (K - 1) / 2, r_hi = K / 2;
            for (int a = 0; a < K; ++a) {
                float va = nrm[a * F + f];
                int rk = 0;
                for (int b = 0; b < K; ++b) {
                    float vb = nrm[b * F + f];
                    rk += (vb < va) || (vb == va && b < a);
                }
                if (rk == r_lo) lo_v = va;
                if (rk == r_hi) hi_v = va;
            }
            float m;
            if (K & 1) {
                m = hi_v;
            } else {
                float sm = lo_v + hi_v;
                m = (float)((double)sm / 2.0);
            }
            med[f] = m;
        }
        adh_wave_sync();
        float sx = 0;
        for (int f = 0; f < F; ++f) sx += med[f];
        float mx = (float)((double)sx / (double)F);
        for (int f = lane; f < F; f += ADH_WAVE) xm[f] = med[f] - mx;
        adh_wave_sync();
        float sxx = 0;
        for (int f = 0; f < F; ++f) sxx += xm[f] * xm[f];
        double var_x = (double)sxx / (double)F;
        for (int k = lane; k < K; k += ADH_WAVE) {
            float sy = 0;
            for (int f = 0; f < F; ++f) sy += isl[k * F + f];
            float my = (float)((double)sy / (double)F);
            float sxy = 0, syy = 0;
            for (int f = 0; f < F; ++f) sxy += xm[f] * (isl[k * F + f] - my);
            for (int f = 0; f < F; ++f) {
                float ym = isl[k * F + f] - my;
                syy += ym * ym;
            }
            double cov = (double)sxy / (double)F;
            double var_y = (double)syy / (double)F;
            double var_xy = var_x * var_y;
            corr[k] = (var_xy == 0) ? 0.0f : (float)(cov / sqrt(var_xy));
        }
    } else {
        for (int c = lane; c < K * O; c += ADH_WAVE) {
            const float *p = ffp + c * F;
            float sm = 0;
            for (int f = 0; f < F; ++f) sm += p[f];
            float mean = sm / (float)F;
            float q = 0;
            for (int f = 0; f < F; ++f) cen[c * F + f] = p[f] - mean;
            for (int f = 0; f < F; ++f) q += cen[c * F + f] * cen[c * F + f];
            fw[c] = sqrtf(q / (float)F);
        }
        adh_wave_sync();
        for (int a = lane; a < K; a += ADH_WAVE) {
            float acc = 0;
            for (int b = 0; b < K; ++b) {
                float red = 0;
                for (int o = 0; o < O; ++o) {
                    float dot = 0;
                    for (int f = 0; f < F; ++f)
                        dot += cen[(a * O + o) * F + f] * cen[(b * O + o) * F + f];
                    float cov = dot / (float)F;
                    float sm = fw[a * O + o] * fw[b * O + o];
                    float cm = (float)((double)cov / ((double)sm + 1e-12));
                    red += cm * oi[o];
                }
                acc += red * g_int[b];
            }
            corr[a] = acc;
        }
    }
    adh_wave_sync();
    if (caps.stop_phase == 71) return;
    float top3 = 0.0f;
    if (lane == 0) {
        int n3 = min(K, 3);
        if (cfg.experimental_xic) {
            float sm = 0;
            for (int i = 0; i < n3; ++i) sm += corr[ord[i]];
            top3 = (float)((double)sm / (double)n3);
        } else {
            float sm = 0;
            for (int i = 0; i < n3; ++i)
                for (int j = 0; j < n3; ++j) {
                    int a = ord[i], b = ord[j];
                    float red = 0;
                    for (int o = 0; o < O; ++o) {
                        float dot = 0;
                        for (int f = 0; f < F; ++f)
                            dot += cen[(a * O + o) * F + f] * cen[(b * O + o) * F + f];
                        float cov = dot / (float)F;
                        float sd = fw[a * O + o] * fw[b * O + o];
                        float cm = (float)((double)cov / ((double)sd + 1e-12));
                        red += cm * oi[o];
                    }
                    sm += red;
                }
            top3 = (float)((double)sm / (double)(n3 * n3));
        }
    }
    adh_wave_sync();
    const double rt_width = run.rt[r.frame_stop - 1] - run.rt[r.frame_start];
    const double mob_width = run.mobility[r.scan_start] - run.mobility[r.scan_stop - 1];
    const bool reg_rows = F <= FR && S <= SR;
    if (reg_rows) {
        // the same statistics with the rows in registers (see load_row)
        for (int c = lane; c < K * O; c += ADH_WAVE) {
            const int k = c / O, o = c - k * O;
            float px[FR], py[FR];
            load_row<FR>(px, ffp + c * F, 1, F);
            load_row<FR>(py, tfp + o * F, 1, F);
            float ym, ysd, xmn, xsd;
            row_moments<FR>(py, F, ym, ysd);
            row_moments<FR>(px, F, xmn, xsd);
            float dot = 0.0f;
#pragma unroll
            for (int i = 0; i < FR; ++i) {
                float dx = px[i] - xmn;
                dx = (i < F) ? dx : 0.0f;  // (one zero factor is enough)
                dot += dx * (py[i] - ym);
            }
            const float cov = dot / (float)F;
            const float sm = xsd * ysd;
            ftc[o * K + k] = (float)((double)cov / ((double)sm + 1e-12));
            float mxv = px[0];
            int am = 0;
#pragma unroll
            for (int i = 1; i < FR; ++i) {
                const bool up = i < F && px[i] > mxv;
                mxv = up ? px[i] : mxv;
                am = up ? i : am;
            }
            const double half = (double)mxv / 2.0;
            int n_above = 0;
#pragma unroll
            for (int i = 0; i < FR; ++i) n_above += (i < F && (double)px[i] > half) ? 1 : 0;
            const double frac = (double)n_above / (double)F;
            fpeak[c] = am;
            fw[c] = (float)(frac * rt_width);
            // mobility FWHM (profile_features.py:151-188)
            float ps[SR];
            load_row<SR>(ps, fsp + c * S, 1, S);
            float mxs = ps[0];
#pragma unroll
            for (int i = 1; i < SR; ++i) mxs = (i < S && ps[i] > mxs) ? ps[i] : mxs;
            const double halfs = (double)mxs / 2.0;
            int n_ab = 0;
#pragma unroll
            for (int i = 0; i < SR; ++i) n_ab += (i < S && (double)ps[i] > halfs) ? 1 : 0;
            const double fracs = (double)n_ab / (double)S;
            mfw[c] = (float)(fracs * mob_width);
        }
    } else
    for (int c = lane; c < K * O; c += ADH_WAVE) {
        int k = c / O, o = c - k * O;
        const float *px = ffp + c * F;
        const float *py = tfp + o * F;
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
        ftc[o * K + k] = (float)((double)cov / ((double)sm + 1e-12));
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
        fpeak[c] = am;
        fw[c] = (float)(frac * rt_width);
        // mobility FWHM (profile_features.py:151-188)
        const float *ps = fsp + c * S;
        float mxs = ps[0];
        for (int sc = 1; sc < S; ++sc) mxs = ps[sc] > mxs ? ps[sc] : mxs;
        double halfs = (double)mxs / 2.0;
        int n_ab = 0;
        for (int sc = 0; sc < S; ++sc) n_ab += ((double)ps[sc] > halfs);
        double fracs = (double)n_ab / (double)S;
        mfw[c] = (float)(fracs * mob_width);
    }
    adh_wave_sync();
    if (caps.stop_phase == 72) return;
    if (lane < O) {
        int o = lane;
        int lo_v = 0, hi_v = 0, r_lo = (K - 1) / 2, r_hi = K / 2;
        for (int a = 0; a < K; ++a) {
            int va = fpeak[a * O + o];
            int rk = 0;
            for (int b = 0; b < K; ++b) {
                int vb = fpeak[b * O + o];
                rk += (vb < va) || (vb == va && b < a);
            }
            if (rk == r_lo) lo_v = va;
            if (rk == r_hi) hi_v = va;
        }
        double m = (K & 1) ? (double)hi_v : (double)(lo_v + hi_v) / 2.0;
        medpk[o] = (float)m;
    }
    adh_wave_sync();
    if (lanes_ok) {
        // features 31-40 (feat::assemble_part2 is the loop form), lane-parallel as above
        const bool kl = lane < K;
        const int k = kl ? lane : 0;
        const int type = g_type[k];
        const bool isb = kl && type == 98, isy = kl && type == 121;
        const unsigned long long bb = __ballot(isb), by = __ballot(isy), below = (1ull << lane) - 1ull;
        const int nbi = __popcll(bb), nyi = __popcll(by);
        if (kl) {
            float rr = 0, ml = 0, mm = 0;
            for (int o = 0; o < O; ++o) rr += ftc[o * K + k] * oi[o];
            for (int o = 0; o < O; ++o) ml += fw[k * O + o] * oi[o];
            for (int o = 0; o < O; ++o) mm += mfw[k * O + o] * oi[o];
            const float gi = g_int[k], co = corr[ord[k]];
            float *u = t32[k];
            u[0] = corr[k];
            u[1] = rr * gi;
            u[2] = ml * gi;
            u[3] = mm * gi;
            // b / y: mask in original order applied to the sorted index array (profile_features.py:94-113)
            u[4] = (isb && __popcll(bb & below) < 3) ? co : 0.0f;
            u[5] = (isy && __popcll(by & below) < 3) ? co : 0.0f;
        }
        adh_wave_sync();
        {
            float v32[KR], s32 = 0.0f;
            const int c32 = min(lane, 5);
#pragma unroll
            for (int j = 0; j < KR; ++j) v32[j] = t32[j][c32];
#pragma unroll
            for (int j = 0; j < KR; ++j) s32 += (j < K) ? v32[j] : 0.0f;
            float *ft = featv;
            if (caps.stop_phase != 20) {
                if (lane == 0) ft[31] = (float)((double)s32 / (double)K), ft[32] = top3;
                if (lane == 1) ft[33] = s32;
                if (lane == 2) ft[38] = s32;
                if (lane == 4 && nbi > 0) ft[34] = (float)((double)s32 / (double)min(nbi, 3)), ft[35] = (float)nbi;
                if (lane == 5 && nyi > 0) ft[36] = (float)((double)s32 / (double)min(nyi, 3)), ft[37] = (float)nyi;
                if (lane == 6) {
                    double acc = 0;
                    for (int o = 0; o < O; ++o) {
                        double delta = (double)medpk[o] - floor((double)F / 2.0);
                        acc += delta * (double)oi[o];
                    }
                    ft[40] = (float)acc;
                }
            }
            if (lane == 3) ft[39] = s32;
        }
    } else if (lane == 0) {
        asmv.top3 = top3;
        if (caps.stop_phase != 20) feat::assemble_part2(asmv, O, K, F);
        float agg = 0;
        for (int k = 0; k < K; ++k) {
            float ml = 0;
            for (int o = 0; o < O; ++o) ml += mfw[k * O + o] * oi[o];
            agg += ml * g_int[k];
        }
        featv[39] = agg;
    }
    adh_wave_sync();
    if (caps.stop_phase == 73) return;

    if (lane < ADH_NUM_FEATURES) out.features[(int64_t)row * ADH_NUM_FEATURES + lane] = featv[lane];
    if (cfg.collect_fragments) {
        const int n = min(K, top_k);
        const int64_t base = (int64_t)row * top_k;
        for (int k = lane; k < n; k += ADH_WAVE) {
            if (out.fragment_precursor_idx) {  // (NULL: the columns that repeat ids / the library are rebuilt later)
                out.fragment_precursor_idx[base + k] = r.precursor_idx;
                out.fragment_rank[base + k] = r.rank;
                out.fragment_mz_library[base + k] = g_mzlib[k];
                out.fragment_mz[base + k] = g_mz[k];
                out.fragment_position[base + k] = g_pos[k];
                out.fragment_number[base + k] = g_number[k];
                out.fragment_type[base + k] = g_type[k];
                out.fragment_charge[base + k] = g_charge[k];
                out.fragment_loss_type[base + k] = g_loss[k];
            }
            out.fragment_mz_observed[base + k] = (float)mzmean[k];
            out.fragment_height[base + k] = (float)height[k];
            out.fragment_intensity[base + k] = (float)area[k];
            out.fragment_mass_error[base + k] = (float)merr[k];
            out.fragment_correlation[base + k] = corr[k];
            if (out.fragment_lib_slot) {
                const LibRec pick = reinterpret_cast<const LibRec *>(block + 32)[kmap[k]];
                out.fragment_lib_slot[base + k] = (uint16_t)(1 + pick.pad0 + 256 * pick.pad1);
            }
        }
    }
    if (lane == 0) out.valid[row] = 1;
}

template <class LAY, bool SPLIT = false>
__global__ __launch_bounds__(ADH_WAVE, 3) void adh_feature_im_kernel(
    DevTims run, const CandRecIM *__restrict__ plan, const float *__restrict__ iso_table,
    int32_t n_iso_cols, adh_scoring_config_t cfg, const unsigned char *__restrict__ scratch,
    DevOut out, Caps caps, unsigned char *__restrict__ prof = nullptr) {
    adh_feature_im_body<LAY, SPLIT>((int)blockIdx.x, run, plan, iso_table, n_iso_cols, cfg, scratch, out, caps, prof);
}
