// adh_features_fast.hip - register-resident feature kernel for the common candidate shape.
//
// Same arithmetic as adh_feature_kernel (adh_features.hip) - which in turn restates
// Candidate.process after get_dense (alphadia/search/scoring/containers/candidate.py:278-481)
// - but organised around the register file instead of LDS:
//
//   * four candidates per 64-lane wavefront, 16 lanes each; lane = fragment (K <= 16)
//   * a lane keeps the XIC row of its fragment in VGPRs, *centred*: register r holds cycle
//     f = r - FM/2 + F/2, so the apex sits in register FM/2 for every F <= FM and all loops over
//     cycles are fully unrolled with constant register indices; cells outside [0, F) hold 0,
//     which leaves float32 sums unchanged, so most reductions need no predication.  The kernel
//     is instantiated for FM = 8, 12, ..., 32 (one observation) and 16, 24, 32 (two)
//   * the three isotope rows use the same registers in an earlier phase
//   * LDS carries only what crosses lanes (template, weight tables, per-fragment results,
//     a 16x16 transpose buffer for the per-cycle median): 3.8 KB per candidate
//   * float32 reductions keep the reference's sequential order -> bit-identical to the
//     generic kernel and the CPU oracle
//
//   * a precursor that overlaps two isolation windows (O == 2) runs the per-observation part
//     twice over the same registers: rows of observation o are loaded, their row sums,
//     weighted centre means and frame-profile statistics are parked in LDS ([fragment][o]),
//     and the rows are added into the fragment's summed profile
//
// Eligibility (decided by the host plan): O <= ADH_FAST_OMAX observations (and quant_all when
// O > 1), 3 <= F <= 32, k_cap <= 16, I <= 4, experimental_xic = True.
//
// The WIDE forms (template parameter GS = 32 / 64, round 5): the same code with two candidates per wavefront and 32
// fragment lanes each, or ONE candidate and 64 lanes, for candidates that keep 17 ... 32 / 33 ... 64 fragments - transfer-library requantification scores with
// top_k_fragments = 9999 against libraries of 20 - 40 fragments per precursor.  Only the per-cycle median differs:
// a lane per cycle sorts the column of all K fragments in registers (bitonic network of 32 or 64 inputs).
// Everything else runs through adh_feature_kernel.
#include "adh_device.h"
#include "adh_feature_common.h"

#define ADH_FMAX 32  // largest cycle count handled by the register kernels
#define ADH_GS 16
#define ADH_FAST_OMAX 2  // observations handled by the register kernels

namespace fast {

using feat::Assemble;

__device__ __forceinline__ double logistic(double x, double mu, double sigma) {
    double a = (x - mu) / sigma;
    return 1.0 / (1.0 + exp(-a));
}

template <int FM, int NO, int GS = 16>
struct __attribute__((aligned(16))) GroupLds {
    union {
        double dT[4][FM];    // isotope contributions to the template of one observation
        float nrmT[GS == 16 ? 16 : FM][GS + 1];  // transpose buffer for the per-cycle median (padded rows)
        struct {                   // per-fragment terms of the feature sums: [fragment][sum]
            double t64[GS][6];
            float t32[GS][6];
        } at;
    } u;
    double wt[2][FM];        // exp weights around the template centre of one observation, centred index
    double merr[GS];
    double ohe[GS][NO], omz[GS][NO];  // [fragment lane][observation]
    double hp[4], omzp[4], qtf[4][NO];
    double red64[12];        // results of the float64 sums
    float tpl[NO][FM], tfp[FM], frt[FM], med[FM];  // centred index
    float g_int[GS], g_fin[GS], corr[GS];
    float rowsum[GS][NO], ftc[GS][NO], fw[GS][NO];  // [fragment lane][o]
    int fpeak[GS][NO];
    float iso_mz[4], iso_int[4], spi[4];
    float oi[NO], tsum[NO], qmask[NO];
    float red32[8];          // results of the float32 sums
    float feat[ADH_NUM_FEATURES + 2];
    int ord[GS];
    int medlo[NO], medhi[NO];
};

// Make a register value opaque to the optimiser (no instruction is emitted): stops LICM from
// hoisting the 64 float->double conversions of a row out of the two-trip scan loop, which
// would keep 128 extra VGPRs live.
#define OPAQUE(x) __asm__ volatile("" : "+v"(x))
#define FOR_R _Pragma("unroll") for (int r = 0; r < FM; ++r)
// Compiler-only fence every 8 unrolled iterations: keeps hipcc from hoisting all 32 (or 64)
// table loads of an unrolled loop to its top, which would double the live register set.
#define R_FENCE(r)                                        \
    do {                                                  \
        if ((((r)) & 7) == 7) __asm__ volatile("" ::: "memory"); \
    } while (0)

// center_envelope_1d (fragment_features.py:71-159) on a centred register row
template <int FM>
__device__ __forceinline__ void center_envelope(float (&x)[FM], int F) {
    constexpr int RC = FM / 2;
    const int c = F / 2;
    if (F & 1) {
        double left = (double)(x[RC - 1] + x[RC]) * 0.5;
        double right = (double)(x[RC + 1] + x[RC]) * 0.5;
#pragma unroll
        for (int i = 1; i <= RC - 1; ++i) {
            if (i <= c) {
                x[RC - i] = (float)fmin(left, (double)x[RC - i]);
                left = (double)(x[RC - i] + x[RC - i + 1]) * 0.5;
                x[RC + i] = (float)fmin(right, (double)x[RC + i]);
                right = (double)(x[RC + i] + x[RC + i - 1]) * 0.5;
            }
        }
    } else {
        // cl = register 15, cr = register 16
        double left = x[RC - 1], right = x[RC];
#pragma unroll
        for (int i = 1; i <= RC - 1; ++i) {
            if (i <= c - 1) {
                x[RC - 1 - i] = (float)fmin(left, (double)x[RC - 1 - i]);
                left = (double)(x[RC - 1 - i] + x[RC - i]) * 0.5;
                x[RC + i] = (float)fmin(right, (double)x[RC + i]);
                right = (double)(x[RC + i] + x[RC + i - 1]) * 0.5;
            }
        }
    }
}

// the 12-input sorting network as a list of compare-exchanges (layer by layer)
#define ADH_SORT12_NETWORK(CE)                                                              \
    CE(0, 1) CE(2, 3) CE(4, 5) CE(6, 7) CE(8, 9) CE(10, 11)                                  \
    CE(1, 3) CE(5, 7) CE(9, 11) CE(0, 2) CE(4, 6) CE(8, 10)                                  \
    CE(1, 2) CE(5, 6) CE(9, 10) CE(0, 4) CE(7, 11)                                           \
    CE(1, 5) CE(6, 10) CE(3, 7) CE(4, 8)                                                     \
    CE(5, 9) CE(2, 6) CE(0, 4) CE(7, 11) CE(3, 8)                                            \
    CE(1, 5) CE(6, 10) CE(2, 3) CE(8, 9)                                                     \
    CE(1, 4) CE(7, 10) CE(3, 5) CE(6, 8)                                                     \
    CE(2, 4) CE(7, 9) CE(5, 6)                                                               \
    CE(3, 4) CE(7, 8)

// ascending bitonic sort of 16 registers
__device__ __forceinline__ void sort16(float (&v)[16]) {
#pragma unroll
    for (int k = 2; k <= 16; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int l = i ^ j;
                if (l > i) {
                    float a = v[i], b = v[l];
                    float lo = fminf(a, b), hi = fmaxf(a, b);
                    bool up = (i & k) == 0;
                    v[i] = up ? lo : hi;
                    v[l] = up ? hi : lo;
                }
            }
        }
    }
}

// ascending bitonic sort of N registers (N a power of two)
template <int N, int M>
__device__ __forceinline__ void sort_pow2(float (&v)[M]) {
    static_assert(N <= M, "sorts the first N of M registers");
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                int l = i ^ j;
                if (l > i) {
                    float a = v[i], b = v[l];
                    float lo = fminf(a, b), hi = fmaxf(a, b);
                    bool up = (i & k) == 0;
                    v[i] = up ? lo : hi;
                    v[l] = up ? hi : lo;
                }
            }
        }
    }
}

// the lanes of a candidate's group that answer `p`, bit 0 = the group's first lane
template <int GS>
__device__ __forceinline__ unsigned long long group_ballot(bool p, unsigned gsh) {
    const unsigned long long b = __ballot(p);
    if constexpr (GS == 64) return b;
    else return (b >> gsh) & ((1ull << GS) - 1ull);
}

// ascending sort of v[0..11] (39 compare-exchanges in 9 layers, the optimal 12-input network; checked against all
// 4096 zero-one inputs by tests/test_host_logic.py); v[12..15] are left alone - callers hold +inf there
__device__ __forceinline__ void sort_first12(float (&v)[16]) {
#define ADH_CE(a, b)                                  \
    {                                                 \
        const float lo_ = fminf(v[a], v[b]);          \
        const float hi_ = fmaxf(v[a], v[b]);          \
        v[a] = lo_;                                   \
        v[b] = hi_;                                   \
    }
    ADH_SORT12_NETWORK(ADH_CE)
#undef ADH_CE
}

// frame-profile statistics of one (fragment, observation) row against the template frame
// profile of that observation: fragment-vs-template correlation (scoring/utils.py:574-647),
// FWHM in RT (profile_features.py:117-146) and apex (profile_features.py:192-193)
template <int FM>
__device__ __forceinline__ void profile_stats(const float (&P)[FM], const float *tfp, int F, int shift,
                                              float rt_width, float &ftc, float &fw, int &fpeak) {
    float syt = 0.0f;
    FOR_R {
        syt += tfp[r];
        R_FENCE(r);
    }
    const float ym = syt / (float)F;
    float sy = 0.0f;
    FOR_R sy += P[r];
    const float xmn = sy / (float)F;
    // the three sums of squares and products in ONE pass (each keeps its own order; round 6: the deviations were
    // computed twice and the template row read three times)
    float qy = 0.0f, qx = 0.0f, dot = 0.0f;
    FOR_R {
        int f = r + shift;
        bool ok = f >= 0 && f < F;
        float dx = ok ? P[r] - xmn : 0.0f;
        float dy = ok ? tfp[r] - ym : 0.0f;
        qy += dy * dy;
        qx += dx * dx;
        dot += dx * dy;
        R_FENCE(r);
    }
    const float ysd = sqrtf(qy / (float)F);
    const float xsd = sqrtf(qx / (float)F);
    const float cv = dot / (float)F;
    const float smm = xsd * ysd;
    ftc = (float)((double)cv / ((double)smm + 1e-12));
    float mxv = 0.0f;
    int am = 0;
    bool first = true;
    FOR_R {
        int f = r + shift;
        bool ok = f >= 0 && f < F;
        if (ok && (first || P[r] > mxv)) {
            mxv = P[r];
            am = f;
            first = false;
        }
    }
    const double half_max = (double)mxv / 2.0;
    int n_above = 0;
    FOR_R {
        int f = r + shift;
        bool ok = f >= 0 && f < F;
        n_above += (ok && (double)P[r] > half_max);
    }
    const double frac = (double)n_above / (double)F;
    fw = (float)(frac * (double)rt_width);
    fpeak = am;
}

}  // namespace fast

// precursor weight table exp(-0.1 * sqrt((s - 2)^2 + (f - 1)^2)), s in {0,1}, f < 64:
// the "expected centre" (S, 1) of precursor_features.py:52-57 does not depend on the candidate
__global__ void adh_wtp_table_kernel(double *table) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * 64) return;
    int sc = i / 64, f = i - sc * 64;
    double ds = (double)(sc - 2), df = (double)(f - 1);
    table[i] = exp(-0.1 * sqrt(ds * ds + df * df));
}

// LDS of one wavefront: the groups' blocks and the precursor weight table
template <int FM, int NO, int GS>
constexpr size_t adh_fast_lds_bytes() {
    return sizeof(fast::GroupLds<FM, NO, GS>) * (ADH_WAVE / GS) + 2 * FM * sizeof(double);
}

// one wavefront = ADH_WAVE / GS candidates of a class; `wave_idx` counts the wavefronts of the class
template <int FM, int NO, int GS>
__device__ __forceinline__ void adh_fast_body(const DevRun &run, const CandRec *__restrict__ plan, int32_t n_cand, int32_t wave_idx,
                                              const float *__restrict__ iso_table, int32_t n_iso_cols,
                                              const adh_scoring_config_t &cfg, const unsigned char *__restrict__ scratch,
                                              const double *__restrict__ wtp_table, const DevOut &out, int32_t stop_phase,
                                              unsigned char *smem) {
    using namespace fast;
    constexpr int RC = FM / 2;
    constexpr int O = NO;  // every candidate of this launch has NO observations (host plan)
    static_assert(GS == 16 || GS == 32 || GS == 64, "16 or 32 lanes per candidate, or the whole wavefront");
    static_assert(sizeof(GroupLds<FM, NO, GS>) % 16 == 0, "the blocks of the groups follow each other");
    GroupLds<FM, NO, GS> *const lds = reinterpret_cast<GroupLds<FM, NO, GS> *>(smem);
    double(*const wtp_s)[FM] = reinterpret_cast<double(*)[FM]>(smem + sizeof(GroupLds<FM, NO, GS>) * (ADH_WAVE / GS));
    const int lane = threadIdx.x;
    // precursor weight table exp(-0.1 * sqrt((s - 2)^2 + (f - 1)^2)), f < F <= FM: the "expected
    // centre" (S, 1) of precursor_features.py:52-57 does not depend on the candidate
    // (requested here, stored behind the plan record and the header: one round trip instead of two before the tile)
    double wtp_l0 = 0.0, wtp_l1 = 0.0;
    if (lane < FM) {
        wtp_l0 = wtp_table[lane];
        wtp_l1 = wtp_table[64 + lane];
    }
    const int g = lane / GS, sub = lane % GS;
    const unsigned gsh = (unsigned)(g * GS);
    GroupLds<FM, NO, GS> &L = lds[g];
    const int ci = wave_idx * (ADH_WAVE / GS) + g;
    bool alive = ci < n_cand;
    const CandRec &rec = plan[alive ? ci : 0];
    alive = alive && !(rec.flags & ADH_FLAG_SKIP);
    const unsigned char *block = scratch + rec.scratch_off;
    const uint32_t *header = reinterpret_cast<const uint32_t *>(block);
    int K0 = alive ? (int)header[0] : 0;
    alive = alive && K0 != 0;
    if (lane < FM) {
        wtp_s[0][lane] = wtp_l0;
        wtp_s[1][lane] = wtp_l1;
    }
    __syncthreads();
    if (__ballot(alive) == 0ull) return;  // (every lane of a dead candidate walks the whole kernel masked: four dead ones need not)
    const uint32_t row = rec.row;
    const int Lc = run.cycle_len;
    const int c0 = rec.frame_start / Lc;
    const int F = alive ? rec.frame_stop / Lc - c0 : 0;
    const int c = F / 2;
    const int shift = c - RC;  // f = r + shift
    const int I = alive ? min(n_iso_cols, (int)cfg.top_k_isotopes) : 0;
    const int top_k = out.top_k;
    const float rt_width = alive ? run.rt[rec.frame_stop - 1] - run.rt[rec.frame_start] : 0.0f;
    if (alive && sub == 0 && out.stat_matched_peaks) out.stat_matched_peaks[row] = header[1];
#pragma unroll
    for (int j = 0; j < (ADH_NUM_FEATURES + 2 + GS - 1) / GS; ++j)
        if (sub + GS * j < ADH_NUM_FEATURES + 2) L.feat[sub + GS * j] = 0.0f;
    // location features (location_features.py:8-33) right away: their four table look-ups would
    // otherwise sit, exposed, in the middle of the single-lane feature assembly
    float loc = 0.0f;
    if (alive && sub < 4) {
        loc = sub == 0   ? run.mobility[rec.scan_start] - run.mobility[rec.scan_stop - 1]
              : sub == 1 ? rt_width
              : sub == 2 ? run.rt[rec.frame_center]
                         : run.mobility[rec.scan_center];
    }

    float A[FM], B[FM];
    // The fragment rows of the first observation are requested HERE, ahead of the precursor phase that does not use the
    // registers: their round trip runs beside it instead of behind it (two wavefronts per SIMD hide nothing; round 6).
    // The quadrupole mask (a result of the precursor phase) is applied where the rows are used.
    float frt_l[(FM + GS - 1) / GS];  // frame RTs of the cycles sub, sub + GS, ...
#pragma unroll
    for (int pass = 0; pass < (FM + GS - 1) / GS; ++pass) {
        int r = min(sub + GS * pass, FM - 1);  // (duplicates of the last row hold the same value)
        int f = r + shift;
        bool ok = alive && f >= 0 && f < F;
        frt_l[pass] = ok ? run.rt[rec.frame_start + f * Lc] : 0.0f;
    }
    const bool frag_lane0 = alive && sub < K0;
    {
        const float2 *fcells = reinterpret_cast<const float2 *>(block + adh_scratch_frag_off(rec.k_cap));
        FOR_R {
            int f = r + shift;
            bool ok = frag_lane0 && f >= 0 && f < F;
            float2 v = fcells[ok ? f * K0 + sub : 0];  // branch-free: clamp the index, mask the value
            A[r] = ok ? v.x : 0.0f;
            B[r] = ok ? v.y : 0.0f;
        }
    }

    // ================= precursor phase =================
    // Three steps since round 6 (before: lane i < I held isotope row i in registers and walked it alone, 3 of a candidate's
    // 32 or 64 lanes busy for a quarter of the wide kernels' time):
    //  1. lanes 0..I-1: the isotope's m/z, library intensity and quadrupole transfer (one logistic pair per observation)
    //  2. lane = cycle: the I cells of the cycle from the scratch block (coalesced), kept in LDS, and the template row
    //     of every observation (the sum over the isotopes, in isotope order)
    //  3. lane = (isotope, sum): the four sequential float64 sums of an isotope row - value and weight of the intensity
    //     and of the m/z channel (precursor_features.py:52-66) - are independent chains: one lane each, side by side
    const bool iso_lane = alive && sub < I;
    {
        const float2 *pcells =
            reinterpret_cast<const float2 *>(block + adh_scratch_prec_off(rec.k_cap, O, F));
        // (the cells of step 2 are requested first: their round trip runs beside the logistic pairs of step 1)
        constexpr int PP = (FM + GS - 1) / GS;
        float2 pv[PP][4];
#pragma unroll
        for (int pass = 0; pass < PP; ++pass) {
            const int f = sub + GS * pass + shift;
            const bool ok = alive && sub + GS * pass < FM && f >= 0 && f < F;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool oki = ok && i < I;
                const float2 v = pcells[oki ? i * F + f : 0];  // branch-free: clamp the index, mask the value
                pv[pass][i] = oki ? v : make_float2(0.0f, 0.0f);
            }
        }
        if (iso_lane) {
            const float iso_int_l = iso_table[(int64_t)row * n_iso_cols + sub];
            double off = (double)sub * 1.0033548350700006 / (double)rec.charge;  // candidate.py:158-163
            const float iso_mz_l = (float)off + rec.precursor_mz;
            // quadrupole_transfer_function_single (quadrupole.py:261-301), n_scans == 1
            double x = (double)iso_mz_l;
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const double *cy = run.cycle + 2 * ((int64_t)rec.obs[o] * run.cycle_scans + rec.scan_start);
                const QuadParams qp = adh_quad_params(cfg);
                L.qtf[sub][o] = logistic(x, cy[0] + qp.delta_lo, qp.sigma_lo) - logistic(x, cy[1] + qp.delta_hi, qp.sigma_hi);
            }
            L.iso_mz[sub] = iso_mz_l;
            L.iso_int[sub] = iso_int_l;
        }
        __syncthreads();
        // (the isotope rows take the bytes of the template contributions they replace)
        static_assert(sizeof(L.u.dT) >= 2 * 4 * FM * sizeof(float), "isotope rows: intensity and m/z of four isotopes");
        float(*const PA)[FM] = reinterpret_cast<float(*)[FM]>(&L.u.dT[0][0]);  // [4][FM] intensity, centred index
        float(*const PB)[FM] = PA + 4;                                          // [4][FM] m/z
#pragma unroll
        for (int pass = 0; pass < PP; ++pass) {
            const int r = sub + GS * pass;
            if (r < FM) {
                double acc[NO];
#pragma unroll
                for (int o = 0; o < NO; ++o) acc[o] = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float a = pv[pass][i].x;
                    PA[i][r] = a;
                    PB[i][r] = pv[pass][i].y;
                    if (i < I) {
                        // template rows of every observation (quadrupole.py:304-324): zero outside [0, F)
                        const float ai = a * L.iso_int[i];
#pragma unroll
                        for (int o = 0; o < NO; ++o) acc[o] += (double)ai * L.qtf[i][o];
                    }
                }
#pragma unroll
                for (int o = 0; o < NO; ++o) L.tpl[o][r] = (float)acc[o];
            }
        }
        __syncthreads();
        // weighted centre means around (S, 1) (precursor_features.py:52-66): lane 4 i + c walks isotope row i for
        // sum c (0: intensity x weight, 1: weight where the intensity is positive, 2 / 3: the same of the m/z channel).
        // A skipped term is an added + 0.0 (the sums start at + 0.0 and their terms are positive: never - 0.0), a
        // weight sum adds 1.0 * w = w.
        {
            const int ci_ = sub >> 2, ch = sub & 3;
            const bool chain = alive && ci_ < I;  // (4 I <= 16 <= GS lanes)
            const float *X = (ch < 2 ? PA : PB)[ci_ & 3];
            const bool is_value = (ch & 1) == 0;
            double acc = 0.0;
            float sfa = 0.0f;
#pragma unroll 1
            for (int sc = 0; sc < 2; ++sc) {
#pragma unroll 4
                for (int r = 0; r < FM; ++r) {
                    const float x = X[r];  // (zero outside the candidate's cycles: a positive cell is a cell of the window)
                    const double w = wtp_s[sc][min(max(r + shift, 0), FM - 1)];
                    const double m = is_value ? (double)x : 1.0;
                    acc += x > 0.0f ? m * w : 0.0;
                    sfa += sc == 0 ? x : 0.0f;  // the row sum over the cycles (the second scan slot doubles it below)
                }
            }
            const double wsum = __shfl_down(acc, 1);  // the weight sum of the lane's value sum
            if (chain && ch == 0) {
                L.spi[ci_] = sfa + sfa;
                L.hp[ci_] = wsum > 0 ? acc / wsum : 0.0;
            }
            if (chain && ch == 2) L.omzp[ci_] = wsum > 0 ? acc / wsum : 0.0;
        }
        // the qtf mask (candidate.py:287-289)
        if (sub == 0) {
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                double qs = 0;
                for (int i = 0; i < I; ++i) qs += L.qtf[i][o];
                L.qmask[o] = (I > 0) ? (float)(qs / (double)I) : 0.0f;
            }
        }
    }
    __syncthreads();
    if (stop_phase == 31) return;
    // observation importance (quadrupole.py:327-335)
    if (sub < O) {
        float st = 0.0f;
        FOR_R {
            st += L.tpl[sub][r];
            R_FENCE(r);
        }
        L.tsum[sub] = st + st;
    }
    // frame RTs (requested ahead of the precursor phase)
#pragma unroll
    for (int pass = 0; pass < (FM + GS - 1) / GS; ++pass) L.frt[min(sub + GS * pass, FM - 1)] = frt_l[pass];
    __syncthreads();
    {
        float tot = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) tot += L.tsum[o];
        if (sub < O) L.oi[sub] = (tot == 0.0f) ? 1.0f / (float)O : L.tsum[sub] / tot;
    }
    if (stop_phase == 32) return;

    // ================= fragment phase: lane = fragment, one pass per observation =================
    float P[FM];  // frame profile summed over observations (frame_profile_2d + sum over o)
    FOR_R P[r] = 0.0f;
    float so = 0.0f;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        // ---- tables of this observation: template centre of mass (fragment_features.py:20-68;
        // every lane computes it), template frame profile, weights around the centre
        double esc, efc;
        {
            // The three sequential float64 sums (intensity; scan slot x intensity; cycle x intensity) are independent
            // chains: lanes 4 j, 4 j + 1, 4 j + 2 of every four walk one each (lane 4 j + 3 repeats the first) and the
            // four share the results - a third of the instructions of every lane walking all three (round 6).  A cell
            // that is not positive enters as 0 (its term +-0 changes no sum: they start at + 0.0); 1.0 * v = v.
            const int ch = sub & 3;
            double acc = 0.0;
#pragma unroll 1  // a real loop: unrolling lets hipcc keep 2 x 32 converted values live
            for (int sc = 0; sc < 2; ++sc) {
                double mr = ch == 2 ? (double)shift : (ch == 1 ? (double)sc : 1.0);  // factor of cycle r: f, sc or 1
                const double inc = ch == 2 ? 1.0 : 0.0;
                FOR_R {
                    const float v = L.tpl[o][r];
                    acc += mr * (double)__builtin_fmaxf(v, 0.0f);  // (the reference skips v <= 0: its term here is +-0)
                    mr += inc;
                    R_FENCE(r);
                }
            }
            const int l4 = lane & ~3;
            const double isum = __shfl(acc, l4), ssum = __shfl(acc, l4 + 1), fsum = __shfl(acc, l4 + 2);
            esc = isum > 0 ? ssum / isum : 0.0;  // (isum > 0 exactly when a cell is positive)
            efc = isum > 0 ? fsum / isum : 0.0;
        }
        __syncthreads();  // the previous observation's tables were consumed
        // template frame profile with or_envelope (scoring/utils.py:46-53)
#pragma unroll
        for (int pass = 0; pass < (FM + GS - 1) / GS; ++pass) {
            int r = min(sub + GS * pass, FM - 1);  // (duplicates of the last row write the same value)
            int f = r + shift;
            bool ok = alive && f >= 0 && f < F;
            float x = L.tpl[o][r] + L.tpl[o][r];
            float rr = x;
            if (ok && f >= 1 && f < F - 1) {
                float xl = L.tpl[o][r - 1] + L.tpl[o][r - 1];
                float xr = L.tpl[o][r + 1] + L.tpl[o][r + 1];
                if (x < xl || x < xr) {
                    float sm = xl + xr;
                    rr = (float)((double)sm / 2.0);
                }
            }
            L.tfp[r] = ok ? rr : 0.0f;
        }
        // weight table around the template centre (features_utils.py:9-25), centred index
#pragma unroll
        for (int pass = 0; pass < (2 * FM + GS - 1) / GS; ++pass) {
            int idx = min(sub + GS * pass, 2 * FM - 1);
            int sc = idx / FM, r = idx - sc * FM;
            int f = r + shift;
            bool ok = alive && f >= 0 && f < F;
            double w = 0.0;
            if (ok) {
                double ds = (double)sc - esc, df = (double)f - efc;
                w = exp(-0.1 * sqrt(ds * ds + df * df));
            }
            L.wt[sc][r] = w;
        }
        __syncthreads();

        // ---- rows of this observation
        {
            const float2 *fcells =
                reinterpret_cast<const float2 *>(block + adh_scratch_frag_off(rec.k_cap));
            const float qmask = L.qmask[o];
            if (o == 0) {  // (requested ahead of the precursor phase)
                FOR_R {
                    int f = r + shift;
                    bool ok = frag_lane0 && f >= 0 && f < F;
                    A[r] = ok ? A[r] * qmask : 0.0f;  // candidate.py:290
                }
            } else {
                FOR_R {
                    int f = r + shift;
                    bool ok = frag_lane0 && f >= 0 && f < F;
                    float2 v = fcells[ok ? (o * F + f) * K0 + sub : 0];
                    A[r] = ok ? v.x * qmask : 0.0f;  // candidate.py:290
                    B[r] = ok ? v.y : 0.0f;
                }
            }
        }
        // presence (candidate.py:319-329): row sum over the two identical scan slots
        float sf = 0.0f;
        FOR_R sf += A[r];
        const float ss = sf + sf;
        L.rowsum[sub][o] = ss;
        so += ss;
        // weighted centre means of both channels (features_utils.py:9-37)
        {
            double vo = 0, wo = 0, vm = 0, wm = 0;
#pragma unroll 1  // a real loop: unrolling lets hipcc keep 2 x 32 converted values live
            for (int sc = 0; sc < 2; ++sc) {
                FOR_R {
                    double w = L.wt[sc][r];
                    float a = A[r], b = B[r];
                    OPAQUE(a);
                    OPAQUE(b);
                    // A skipped cell (value <= 0, features_utils.py:9-37) enters as 0: its product + 0 changes no sum.  Its
                    // weight stays out: w x 1.0 + wo rounds once, like wo + w; w x 0.0 adds + 0 (one select on the high
                    // word of the indicator instead of two on the weight; as in adh_fused.hip)
                    vo += (double)__builtin_fmaxf(a, 0.0f) * w;
                    wo = __builtin_fma(w, __hiloint2double(a > 0.0f ? 0x3ff00000 : 0, 0), wo);
                    vm += (double)__builtin_fmaxf(b, 0.0f) * w;
                    wm = __builtin_fma(w, __hiloint2double(b > 0.0f ? 0x3ff00000 : 0, 0), wm);
                    R_FENCE(r);
                }
            }
            // (the weights of the window's cells are positive: a weight sum is positive exactly when a cell counted)
            L.ohe[sub][o] = wo > 0 ? vo / wo : 0.0;
            L.omz[sub][o] = wm > 0 ? vm / wm : 0.0;
        }
        if (NO > 1) {
            // per-observation frame profile (frame_profile_2d): statistics against this
            // observation's template.  With one observation they are taken after the envelope
            // step, whose in-place edit they must see when quant_all is off.
            FOR_R A[r] = A[r] + A[r];
            float ftc_o, fw_o;
            int fpeak_o;
            profile_stats<FM>(A, L.tfp, F, shift, rt_width, ftc_o, fw_o, fpeak_o);
            L.ftc[sub][o] = ftc_o;
            L.fw[sub][o] = fw_o;
            L.fpeak[sub][o] = fpeak_o;
            FOR_R P[r] += A[r];
        } else {
            FOR_R P[r] += A[r] + A[r];
        }
    }
    if (stop_phase == 33) return;
    LibRec lrec;
    if (frag_lane0) lrec = reinterpret_cast<const LibRec *>(block + 32)[sub];
    bool present = frag_lane0 && so > 0.0f;
    const unsigned long long gm = group_ballot<GS>(present, gsh);
    int K = __popcll(gm);
    const int kk = __popcll(gm & ((1ull << sub) - 1ull));
    const int n_present = K;
    if (K < 2) {  // candidate.py:323
        alive = false;
        present = false;
        K = 0;
    }
    if (present) {
        L.g_fin[kk] = lrec.intensity;  // raw intensity, normalised below
    }
    __syncthreads();
    // fragment intensities: apply_mask renormalisation + the second one of fragment_features.py:218
    float g_int_l = 0.0f, g_fin_l = 0.0f;
    {
        float sum1 = 0.0f;
#pragma unroll 4
        for (int j = 0; j < K; ++j) sum1 += L.g_fin[j];
        if (present) {
            g_int_l = lrec.intensity / sum1;
            L.g_int[kk] = g_int_l;
        }
    }
    __syncthreads();
    {
        float sum2 = 0.0f;
#pragma unroll 4
        for (int j = 0; j < K; ++j) sum2 += L.g_int[j];
        if (present) g_fin_l = g_int_l / sum2;
    }
    if (present) L.g_fin[kk] = g_fin_l;  // raw values were consumed before the last barrier
    if (stop_phase == 3 || stop_phase == 4) return;

    // ---- envelope, quantification (fragment_features.py:240-273)
    double area = 0.0;
    float obs_int = 0.0f;
    {
        float E[FM];  // np.sum(axis=1) made a copy: with quant_all the profile itself is untouched
        FOR_R E[r] = P[r];
        center_envelope<FM>(E, F);
        const int qw = min(c - 1, (int)cfg.quant_window);
        double ar = 0.0;
#pragma unroll
        for (int r = 1; r < FM - 1; ++r) {
            if (r >= RC - qw && r + 1 <= RC + qw) {
                float sm = E[r + 1] + E[r];
                float drt = L.frt[r + 1] - L.frt[r];
                float m = sm * drt;
                ar += (double)m * 0.5;
            }
        }
        area = ar * (double)qw;
        FOR_R {
            if (r >= RC - qw && r <= RC + qw) obs_int += E[r];
        }
        if (NO == 1 && !cfg.quant_all) {
            FOR_R P[r] = E[r];  // a VIEW of the best observation's profile: edited in place
        }
    }
    double m1 = 0.0, m2 = 0.0, merr_l = 0.0;
    bool hrow = false;
    if (present) {
        // importance-weighted means over observations (fragment_features.py:311-336)
        float ws = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            bool m = L.ohe[sub][o] > 0;
            hrow = hrow || m;
            float w32 = m ? L.oi[o] : L.oi[o] * 0.0f;
            ws += w32;
        }
        double msum = 0.0;
        int nm = 0;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            bool m = L.ohe[sub][o] > 0;
            float w32 = m ? L.oi[o] : L.oi[o] * 0.0f;
            double w = (double)w32 / ((double)ws + 1e-20);
            if (w > 0) {
                msum += w;
                ++nm;
            }
        }
        if (nm > 0) {
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                bool m = L.ohe[sub][o] > 0;
                float w32 = m ? L.oi[o] : L.oi[o] * 0.0f;
                double w = (double)w32 / ((double)ws + 1e-20);
                if (w > 0) {
                    double lw = w / msum;
                    m1 += L.omz[sub][o] * lw;
                    m2 += L.ohe[sub][o] * lw;
                }
            }
        }
        merr_l = (m1 - (double)lrec.mz) / (double)lrec.mz * 1e6;  // fragment_features.py:387
        L.merr[kk] = merr_l;
        int rk = 0;
#pragma unroll 4
        for (int j = 0; j < K; ++j) {
            float ib = L.g_int[j];
            rk += (int)((ib > g_int_l) | ((ib == g_int_l) & (j > kk)));
        }
        L.ord[rk] = kk;  // position in argsort(intensity)[::-1]
    }
    __syncthreads();
    if (stop_phase == 5) return;

    if (alive && sub < 4) L.feat[sub] = loc;
    if (alive && sub == 0) {
        Assemble asmv;
        asmv.run = nullptr;  // features 0-3 are in place
        asmv.rec = &rec;
        asmv.featv = L.feat;
        asmv.iso_int = L.iso_int; asmv.iso_mz = L.iso_mz; asmv.spi = L.spi; asmv.oi = L.oi;
        asmv.omzp = L.omzp; asmv.hp = L.hp;
        asmv.n_present = n_present; asmv.K0 = K0;
        feat::assemble_precursor_regs<NO>(asmv, I, O);  // (I <= 4: the host plan)
    }
    // ---- fragment features 17-27, 41-45 (fragment_features.py:198-427; the scalar form is
    // feat::assemble_fragments).  Every sum over fragments keeps the reference's order
    // (k ascending) but all sums advance together: lane k provides its term of every sum,
    // then lane j adds up sum j.  Skipped terms are added as +0, which leaves a sum unchanged.
    const bool ipos = present && obs_int > 0.0f;
    const bool hpos = present && m2 > 0.0;
    const bool isb = present && lrec.type == 98, isy = present && lrec.type == 121;
    const unsigned long long b_isb = group_ballot<GS>(isb, gsh);
    const unsigned long long b_isy = group_ballot<GS>(isy, gsh);
    const int n_int = __popcll(group_ballot<GS>(ipos, gsh));
    const int n_hei = __popcll(group_ballot<GS>(hpos, gsh));
    const int n_hrows = __popcll(group_ballot<GS>(present && hrow, gsh));
    const int nb = __popcll(b_isb), ny = __popcll(b_isy);
    int min_y = isy ? (int)lrec.position : 255, max_b = isb ? (int)lrec.position : 0;
#pragma unroll
    for (int m = GS / 2; m > 0; m >>= 1) {
        min_y = min(min_y, __shfl_xor(min_y, m, GS));
        max_b = max(max_b, __shfl_xor(max_b, m, GS));
    }
    const bool ov = (isy && (int)lrec.position < max_b) || (isb && (int)lrec.position > min_y);
    const int n_ov = __popcll(group_ballot<GS>(ov, gsh));
    const int n3 = min(K, 3);
    if (present) {
        double *t = L.u.at.t64[kk];
        t[0] = area;
        t[1] = m2;
        t[2] = (double)g_fin_l;
        t[3] = merr_l;
        t[4] = ov ? area : 0.0;
        t[5] = ov ? merr_l : 0.0;
        // cosine_similarity_a1 (features_utils.py:40-47) of the observation sums
        float tn = 0.0f, fn = 0.0f, dot = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) tn += L.tsum[o] * L.tsum[o];
        tn = sqrtf(tn);
#pragma unroll
        for (int o = 0; o < NO; ++o) fn += L.rowsum[sub][o] * L.rowsum[sub][o];
        fn = sqrtf(fn);
#pragma unroll
        for (int o = 0; o < NO; ++o) dot += L.rowsum[sub][o] * L.tsum[o];
        const float pr = fn * tn;
        const float score = (float)((double)dot / ((double)pr + 0.0001));
        float *u = L.u.at.t32[kk];
        u[0] = ipos ? g_fin_l : 0.0f;
        u[1] = hpos ? g_fin_l : 0.0f;
        u[2] = ipos ? score : 0.0f;
        u[3] = isb ? obs_int : 0.0f;
        u[4] = isy ? obs_int : 0.0f;
    }
    __syncthreads();
    {
        double s64 = 0.0;
        float s32 = 0.0f;
        if (sub < 6) {
#pragma unroll 4
            for (int k = 0; k < K; ++k) s64 += L.u.at.t64[k][sub];
            L.red64[sub] = s64;
        } else if (sub < 11) {
#pragma unroll 4
            for (int k = 0; k < K; ++k) s32 += L.u.at.t32[k][sub - 6];
            L.red32[sub - 6] = s32;
        } else if (sub == 11) {
            for (int i = 0; i < n3; ++i) s64 += L.merr[L.ord[i]];  // mean_top3 mass error, by rank
            L.red64[6] = s64;
        }
    }
    __syncthreads();
    {
        // np.corrcoef terms (feat::corrcoef01): area vs intensity, height vs intensity
        const double mx_a = L.red64[0] / (double)K, mx_h = L.red64[1] / (double)K;
        const double my = L.red64[2] / (double)K;
        if (present) {
            const double a = area - mx_a, h = m2 - mx_h, b = (double)g_fin_l - my;
            double *t = L.u.at.t64[kk];
            t[0] = a * a;
            t[1] = b * b;
            t[2] = a * b;
            t[3] = h * h;
            t[4] = h * b;
        }
    }
    __syncthreads();
    if (sub < 5) {
        double s64 = 0.0;
#pragma unroll 4
        for (int k = 0; k < K; ++k) s64 += L.u.at.t64[k][sub];
        L.red64[7 + sub] = s64;
    }
    __syncthreads();
    if (alive && sub < 2) {
        // lane 0: feature 18 (areas), lane 1: feature 19 (heights)
        const double fact = fmax((double)K - 1.0, 0.0);
        const double inv = 1.0 / fact;
        const double cxx = L.red64[sub ? 10 : 7] * inv, cyy = L.red64[8] * inv;
        const double cxy = L.red64[sub ? 11 : 9] * inv;
        const double s0 = sqrt(cxx), s1 = sqrt(cyy);
        double cc = cxy / s1 / s0;
        if (fabs(cc) > 1.0) cc = (cc > 0) ? 1.0 : -1.0;
        const bool on = sub ? (L.red64[1] > 0.0) : (n_hrows > 0);
        if (on) L.feat[18 + sub] = (float)cc;
    }
    if (alive && sub == 0) {
        float *ft = L.feat;
        ft[17] = (float)O;
        ft[20] = (float)((double)n_int / (double)K);
        ft[21] = (float)((double)n_hei / (double)K);
        ft[22] = L.red32[0];
        ft[23] = L.red32[1];
        if (n_int > 0) ft[24] = (float)((double)L.red32[2] / (double)n_int);
        ft[25] = nb > 0 ? (float)log((double)L.red32[3] + 1.0) : 0.0f;
        ft[26] = ny > 0 ? (float)log((double)L.red32[4] + 1.0) : 0.0f;
        ft[27] = ft[25] - ft[26];
        ft[41] = (float)(L.red64[6] / (double)n3);
        ft[42] = (float)(L.red64[3] / (double)K);
        if (nb > 0 && ny > 0) {
            ft[43] = (float)n_ov;
            if (n_ov > 0) {
                ft[44] = (float)(L.red64[4] / (double)n_ov);
                ft[45] = (float)(L.red64[5] / (double)n_ov);
            } else {
                ft[44] = 0.0f;
                ft[45] = 15.0f;
            }
        }
    }
    if (stop_phase == 6) return;

    // ================= profile features (profile_features.py:18-206), experimental_xic =======
    // intensity_slice = frame profile summed over the observations = P
    {
        // normalize_profiles (scoring_utils.py:71-117): centre +- 1 are registers RC-1, RC, RC+1
        float sm = 0.0f;
        sm += P[RC - 1];
        sm += P[RC];
        sm += P[RC + 1];
        const double cn = (double)sm / 3.0;
        // median over fragments per cycle (scoring_utils.py:120-152): 16x16 transposes via LDS
        if constexpr (GS > 16) {
            // wide forms: the whole [cycle][fragment] table at once, then lane r < FM sorts column r
            __syncthreads();  // previous users of the union are done
            if (present) {
                FOR_R {
                    float x = P[r];
                    L.u.nrmT[r][kk] = (cn > 0) ? (float)((double)x / cn) : 0.0f;
                    if ((r & 3) == 3) __asm__ volatile("" ::: "memory");  // bound the in-flight divisions
                }
            }
            __syncthreads();
            const int r_lo = (K - 1) / 2, r_hi = K / 2;
            float lo_v = 0.0f, hi_v = 0.0f;
            if (sub < FM) {
                float v[GS];
                if (GS == 32 || K <= 32) {  // (GS = 64: one candidate per wavefront, the branch is uniform)
#pragma unroll
                    for (int j = 0; j < 32; ++j) v[j] = (j < K) ? L.u.nrmT[sub][j] : INFINITY;
                    sort_pow2<32>(v);
#pragma unroll
                    for (int j = 0; j < 32; ++j) {
                        lo_v = (j == r_lo) ? v[j] : lo_v;
                        hi_v = (j == r_hi) ? v[j] : hi_v;
                    }
                } else if constexpr (GS == 64) {
#pragma unroll
                    for (int j = 0; j < 64; ++j) v[j] = (j < K) ? L.u.nrmT[sub][j] : INFINITY;
                    sort_pow2<64>(v);
#pragma unroll
                    for (int j = 0; j < 64; ++j) {
                        lo_v = (j == r_lo) ? v[j] : lo_v;
                        hi_v = (j == r_hi) ? v[j] : hi_v;
                    }
                }
                float m;
                if (K & 1) {
                    m = hi_v;
                } else {
                    float s2 = lo_v + hi_v;
                    m = (float)((double)s2 / 2.0);
                }
                const int f = sub + shift;
                L.med[sub] = (alive && f >= 0 && f < F) ? m : 0.0f;
            }
        } else {
#pragma unroll
        for (int half = 0; half < (FM + 15) / 16; ++half) {
            __syncthreads();  // previous users of the union are done
            if (present) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    if (half * 16 + t >= FM) break;
                    float x = P[half * 16 + t];
                    L.u.nrmT[t][kk] = (cn > 0) ? (float)((double)x / cn) : 0.0f;
                    if ((t & 3) == 3) __asm__ volatile("" ::: "memory");  // bound the in-flight divisions
                }
            }
            __syncthreads();
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = (j < K) ? L.u.nrmT[sub][j] : INFINITY;
            sort16(v);
            const int r_lo = (K - 1) / 2, r_hi = K / 2;
            float lo_v = 0.0f, hi_v = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                lo_v = (j == r_lo) ? v[j] : lo_v;
                hi_v = (j == r_hi) ? v[j] : hi_v;
            }
            float m;
            if (K & 1) {
                m = hi_v;
            } else {
                float s2 = lo_v + hi_v;
                m = (float)((double)s2 / 2.0);
            }
            const int r = half * 16 + sub;
            const int f = r + shift;
            if (r < FM) L.med[r] = (alive && f >= 0 && f < F) ? m : 0.0f;
        }
        }
    }
    __syncthreads();
    if (stop_phase == 61) return;
    float corr_l = 0.0f;
    {
        // correlation_coefficient (scoring_utils.py:14-68)
        float sx = 0.0f;
        FOR_R {
            sx += L.med[r];
            R_FENCE(r);
        }
        const float mx = (float)((double)sx / (double)F);
        float sxx = 0.0f, sy = 0.0f;
        FOR_R sy += P[r];
        const float my = (float)((double)sy / (double)F);
        float sxy = 0.0f, syy = 0.0f;
        FOR_R {  // (one pass for the three sums, each in its own order)
            int f = r + shift;
            bool ok = f >= 0 && f < F;
            float xm = ok ? L.med[r] - mx : 0.0f;
            float ym = ok ? P[r] - my : 0.0f;
            sxx += xm * xm;
            sxy += xm * ym;
            syy += ym * ym;
            R_FENCE(r);
        }
        const double var_x = (double)sxx / (double)F;
        const double cov = (double)sxy / (double)F;
        const double var_y = (double)syy / (double)F;
        const double var_xy = var_x * var_y;
        corr_l = (var_xy == 0) ? 0.0f : (float)(cov / sqrt(var_xy));
    }
    if (NO == 1) {
        float ftc_o, fw_o;
        int fpeak_o;
        profile_stats<FM>(P, L.tfp, F, shift, rt_width, ftc_o, fw_o, fpeak_o);
        L.ftc[sub][0] = ftc_o;
        L.fw[sub][0] = fw_o;
        L.fpeak[sub][0] = fpeak_o;
    }
    if (stop_phase == 62) return;
    if (present) L.corr[kk] = corr_l;
    __syncthreads();
    // ---- features 31-38, 40 (profile_features.py:70-113,141-146,196-204; the scalar form is
    // feat::assemble_part2), sums organised as above
    {
        const int r_lo = (K - 1) / 2, r_hi = K / 2;
        // median apex per observation (profile_features.py:196-198): rank of this fragment's apex
        int apex[NO], apex_rk[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const int va = L.fpeak[sub][o];
            int rk = 0;
            if constexpr (GS > 16) {
                // wide form: a real loop over the lanes that hold a fragment; lane b's apex is a lane read of the
                // register it sits in (b is uniform), not an LDS read whose latency every trip would wait for (round 6)
                const int kmax = __builtin_amdgcn_readfirstlane(GS == 64 ? K0 : max(K0, __shfl_xor(K0, 32)));
                for (int b = 0; b < kmax; ++b) {
                    int vb = __builtin_amdgcn_readlane(va, b);
                    if constexpr (GS == 32) vb = g ? __builtin_amdgcn_readlane(va, 32 + b) : vb;
                    const bool pb = (gm >> b) & 1ull;  // (no bit at or above K0)
                    rk += (int)(pb & ((vb < va) | ((vb == va) & (b < sub))));
                }
            } else {
#pragma unroll
                for (int b = 0; b < GS; ++b) {
                    if (!((gm >> b) & 1ull)) continue;
                    int vb = L.fpeak[b][o];
                    rk += (int)((vb < va) | ((vb == va) & (b < sub)));
                }
            }
            apex[o] = va;
            apex_rk[o] = rk;
        }
        if (present) {
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                if (apex_rk[o] == r_lo) L.medlo[o] = apex[o];
                if (apex_rk[o] == r_hi) L.medhi[o] = apex[o];
            }
            const float cr = L.corr[L.ord[kk]];  // correlation of the fragment with intensity rank kk
            // b / y: mask in original order applied to the sorted index array (profile_features.py:94-113)
            const bool b3 = isb && __popcll(b_isb & ((1ull << sub) - 1ull)) < 3;
            const bool y3 = isy && __popcll(b_isy & ((1ull << sub) - 1ull)) < 3;
            float rr = 0.0f, ml = 0.0f;
#pragma unroll
            for (int o = 0; o < NO; ++o) rr += L.ftc[sub][o] * L.oi[o];
#pragma unroll
            for (int o = 0; o < NO; ++o) ml += L.fw[sub][o] * L.oi[o];
            float *u = L.u.at.t32[kk];
            u[0] = corr_l;
            u[1] = rr * g_int_l;
            u[2] = ml * g_int_l;
            u[3] = b3 ? cr : 0.0f;
            u[4] = y3 ? cr : 0.0f;
            u[5] = (kk < n3) ? cr : 0.0f;
        }
    }
    __syncthreads();
    if (sub < 6) {
        float s32 = 0.0f;
#pragma unroll 4
        for (int k = 0; k < K; ++k) s32 += L.u.at.t32[k][sub];
        L.red32[sub] = s32;
    }
    __syncthreads();
    if (alive && sub == 0) {
        float *ft = L.feat;
        ft[31] = (float)((double)L.red32[0] / (double)K);
        ft[32] = (float)((double)L.red32[5] / (double)n3);
        ft[33] = L.red32[1];
        if (nb > 0) {
            ft[34] = (float)((double)L.red32[3] / (double)min(nb, 3));
            ft[35] = (float)nb;
        }
        if (ny > 0) {
            ft[36] = (float)((double)L.red32[4] / (double)min(ny, 3));
            ft[37] = (float)ny;
        }
        ft[38] = L.red32[2];
        double acc = 0.0;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const double med = (K & 1) ? (double)L.medhi[o] : (double)(L.medlo[o] + L.medhi[o]) / 2.0;
            const float medpk = (float)med;
            acc += ((double)medpk - floor((double)F / 2.0)) * (double)L.oi[o];
        }
        ft[40] = (float)acc;
    }
    __syncthreads();

    // ---- output row (candidate.py:403-481)
    if (alive) {
#pragma unroll
        for (int j = 0; j < (ADH_NUM_FEATURES + GS - 1) / GS; ++j) {
            int idx = sub + GS * j;
            if (idx < ADH_NUM_FEATURES) out.features[(int64_t)row * ADH_NUM_FEATURES + idx] = L.feat[idx];
        }
        if (cfg.collect_fragments && present && kk < top_k) {
            const int64_t o = (int64_t)row * top_k + kk;
            if (out.fragment_precursor_idx) {  // (NULL: the columns that repeat ids / the library are rebuilt later)
                out.fragment_precursor_idx[o] = rec.precursor_idx;
                out.fragment_rank[o] = rec.rank;
                out.fragment_mz_library[o] = lrec.mz_library;
                out.fragment_mz[o] = lrec.mz;
                out.fragment_position[o] = lrec.position;
                out.fragment_number[o] = lrec.number;
                out.fragment_type[o] = lrec.type;
                out.fragment_charge[o] = lrec.charge;
                out.fragment_loss_type[o] = lrec.loss_type;
            }
            out.fragment_mz_observed[o] = (float)m1;
            out.fragment_height[o] = (float)m2;
            out.fragment_intensity[o] = (float)area;
            out.fragment_mass_error[o] = (float)merr_l;
            out.fragment_correlation[o] = corr_l;
            if (out.fragment_lib_slot) out.fragment_lib_slot[o] = (uint16_t)(1 + lrec.pad0 + 256 * lrec.pad1);
        }
        if (sub == 0) out.valid[row] = 1;
    }
}

template <int FM, int NO, int GS = 16>
__global__ __launch_bounds__(ADH_WAVE) void adh_feature_fast_kernel(
    DevRun run, const CandRec *__restrict__ plan, int32_t n_cand, const float *__restrict__ iso_table,
    int32_t n_iso_cols, adh_scoring_config_t cfg, const unsigned char *__restrict__ scratch,
    const double *__restrict__ wtp_table, DevOut out, int32_t stop_phase) {
    __shared__ __align__(16) unsigned char smem[adh_fast_lds_bytes<FM, NO, GS>()];
    adh_fast_body<FM, NO, GS>(run, plan, n_cand, (int32_t)blockIdx.x, iso_table, n_iso_cols, cfg, scratch, wtp_table, out,
                              stop_phase, smem);
}

// The wide classes of one batch (17 ... 64 fragments kept) in TWO launches: a chunk of the host -> host pipeline holds a
// few thousand candidates of each (observations, cycles, lanes) class, a wavefront lives ~50 us, and twelve launches
// of two rounds each spent most of their time filling and draining the GPU (round 5: one launch per observation
// count).  Round 6: one kernel runs at the register count and LDS block of its LARGEST body, and the wide kernels wait
// for dependent round trips - two wavefronts per SIMD hide little.  The bodies that fit three wavefronts per SIMD (16
// cycles with either lane count; 24 cycles on 64 lanes for one observation: <= 157 registers, <= 12.7 KB) of BOTH
// observation counts go out as one launch behind the heavy ones of both.  ADH_WIDE_SPLIT=0: everything in one launch.
// A kind is (FM - 16) / 8 + 3 * (lanes per candidate == 32) + 6 * (observations == 2).
#define ADH_WIDE_MAX_CLASSES 12
struct WideClasses {
    int32_t n;                                      // classes in this launch
    int32_t first_block[ADH_WIDE_MAX_CLASSES + 1];  // first wavefront of class i; [n] = all
    int32_t first_cand[ADH_WIDE_MAX_CLASSES];       // first candidate of the class, counted from `plan`
    int32_t n_cand[ADH_WIDE_MAX_CLASSES];
    int32_t kind[ADH_WIDE_MAX_CLASSES];
};
#ifndef ADH_WIDE_SPLIT
#define ADH_WIDE_SPLIT 1
#endif
#define ADH_WIDE_ALL 0xFFF
#ifndef ADH_WIDE_LIGHT_SET
#define ADH_WIDE_LIGHT_SET (0x00B | (0x009 << 6))
#endif
#ifndef ADH_WIDE_LIGHT_WAVES
#define ADH_WIDE_LIGHT_WAVES 1
#endif
#define ADH_WIDE_LIGHT (ADH_WIDE_SPLIT ? ADH_WIDE_LIGHT_SET : 0)  // kinds {0, 1, 3} of one, {0, 3} of two observations
#define ADH_WIDE_HEAVY (ADH_WIDE_ALL & ~ADH_WIDE_LIGHT)
template <int KINDS>
constexpr size_t adh_wide_lds_bytes() {
    size_t b = 0;
    auto take = [&](int kind, size_t bytes) {
        if (((KINDS >> kind) & 1) && bytes > b) b = bytes;
    };
    take(0, adh_fast_lds_bytes<16, 1, 64>());
    take(1, adh_fast_lds_bytes<24, 1, 64>());
    take(2, adh_fast_lds_bytes<32, 1, 64>());
    take(3, adh_fast_lds_bytes<16, 1, 32>());
    take(4, adh_fast_lds_bytes<24, 1, 32>());
    take(5, adh_fast_lds_bytes<32, 1, 32>());
    take(6, adh_fast_lds_bytes<16, 2, 64>());
    take(7, adh_fast_lds_bytes<24, 2, 64>());
    take(8, adh_fast_lds_bytes<32, 2, 64>());
    take(9, adh_fast_lds_bytes<16, 2, 32>());
    take(10, adh_fast_lds_bytes<24, 2, 32>());
    take(11, adh_fast_lds_bytes<32, 2, 32>());
    return b ? b : 16;  // (no bodies: the empty second launch of a build without the split)
}
// The arguments travel as ONE struct read through the kernel-argument segment where they are needed (round 6, as
// adh_fused_kernel since round 5): as formal parameters the ~120 dwords (DevRun, the class table, the config, the output
// pointers) are loaded in the prologue of a kernel that holds six bodies, do not fit the scalar registers beside the
// rest - 848 spilled scalar registers - and come back one v_readlane (a vector-ALU slot) per dword and use.
struct WideArgs {
    DevRun run;
    const CandRec *plan;
    WideClasses wc;
    const float *iso_table;
    int32_t n_iso_cols;
    adh_scoring_config_t cfg;
    const unsigned char *scratch;
    const double *wtp_table;
    DevOut out;
    int32_t stop_phase;
};
// Two wavefronts per SIMD as the register budget of the heavy launch: left alone the two-observation bodies of 24 cycles
// take 256 + 6 registers and run ONE wavefront per SIMD (round 6: 364 -> 274 us per launch of the
// transfer-requantification leg).
#define ADH_WIDE_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(KINDS == ADH_WIDE_LIGHT ? ADH_WIDE_LIGHT_WAVES : 2)))
template <int KINDS>
__global__ __launch_bounds__(ADH_WAVE) ADH_WIDE_WAVES_ATTR void adh_feature_wide_kernel(WideArgs formal_args_not_read) {
    const WideArgs &A = *(const WideArgs *)__builtin_amdgcn_kernarg_segment_ptr();  // (the only argument: offset 0)
    const DevRun &run = A.run;
    const CandRec *__restrict__ plan = A.plan;
    const WideClasses &wc = A.wc;
    const float *__restrict__ iso_table = A.iso_table;
    const int32_t n_iso_cols = A.n_iso_cols;
    const adh_scoring_config_t &cfg = A.cfg;
    const unsigned char *__restrict__ scratch = A.scratch;
    const double *__restrict__ wtp_table = A.wtp_table;
    const DevOut &out = A.out;
    const int32_t stop_phase = A.stop_phase;
    __shared__ __align__(16) unsigned char smem[adh_wide_lds_bytes<KINDS>()];
    const int32_t b = (int32_t)blockIdx.x;
    int c = 0;
    while (c + 1 < wc.n && b >= wc.first_block[c + 1]) ++c;
    const CandRec *recs = plan + wc.first_cand[c];
    const int32_t n = wc.n_cand[c], blk = b - wc.first_block[c];
#define ADH_WIDE_CASE(KIND, NO, FM, GS)                                                                                    \
    if constexpr ((KINDS >> KIND) & 1) {                                                                                   \
        if (wc.kind[c] == KIND) {                                                                                          \
            static_assert(adh_fast_lds_bytes<FM, NO, GS>() <= adh_wide_lds_bytes<KINDS>(), "the largest body's block");    \
            adh_fast_body<FM, NO, GS>(run, recs, n, blk, iso_table, n_iso_cols, cfg, scratch, wtp_table, out, stop_phase,  \
                                      smem);                                                                               \
            return;                                                                                                        \
        }                                                                                                                  \
    }
    ADH_WIDE_CASE(0, 1, 16, 64)
    ADH_WIDE_CASE(1, 1, 24, 64)
    ADH_WIDE_CASE(2, 1, 32, 64)
    ADH_WIDE_CASE(3, 1, 16, 32)
    ADH_WIDE_CASE(4, 1, 24, 32)
    ADH_WIDE_CASE(5, 1, 32, 32)
    ADH_WIDE_CASE(6, 2, 16, 64)
    ADH_WIDE_CASE(7, 2, 24, 64)
    ADH_WIDE_CASE(8, 2, 32, 64)
    ADH_WIDE_CASE(9, 2, 16, 32)
    ADH_WIDE_CASE(10, 2, 24, 32)
    ADH_WIDE_CASE(11, 2, 32, 32)
#undef ADH_WIDE_CASE
}
