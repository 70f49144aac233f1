// adh_features_fast.hip - register-resident feature kernel for the common candidate shape.
//
// Same arithmetic as adh_feature_kernel (adh_features.hip) - which in turn restates
// Candidate.process after get_dense (alphadia/search/scoring/containers/candidate.py:278-481)
// - but organised around the register file instead of LDS:
//
//   * four candidates per 64-lane wavefront, 16 lanes each; lane = fragment (K <= 16)
//   * a lane keeps the XIC row of its fragment in VGPRs, *centred*: register r holds cycle
//     f = r - 16 + F/2, so the apex sits in register 16 for every F <= 32 and all loops over
//     cycles are fully unrolled with constant register indices; cells outside [0, F) hold 0,
//     which leaves float32 sums unchanged, so most reductions need no predication
//   * the three isotope rows use the same registers in an earlier phase
//   * LDS carries only what crosses lanes (template, weight tables, per-fragment results,
//     a 16x16 transpose buffer for the per-cycle median): 3.8 KB per candidate
//   * float32 reductions keep the reference's sequential order -> bit-identical to the
//     generic kernel and the CPU oracle
//
// Eligibility (decided by the host plan): one observation (O == 1), 3 <= F <= 32,
// k_cap <= 16, I <= 4, experimental_xic = True, quant_window <= 15.  Everything else runs
// through adh_feature_kernel.
#include "adh_device.h"
#include "adh_feature_common.h"

#define ADH_FMAX 32  // largest cycle count handled by the register kernels
#define ADH_GS 16

namespace fast {

using feat::Assemble;

__device__ __forceinline__ double logistic(double x, double mu, double sigma) {
    double a = (x - mu) / sigma;
    return 1.0 / (1.0 + exp(-a));
}

template <int FM>
struct __attribute__((aligned(16))) GroupLds {
    union {
        double dT[4][FM];    // isotope contributions to the template
        float nrmT[16][17];        // transpose buffer for the per-cycle median (padded rows)
    } u;
    double wt[2][FM];        // exp weights around the template centre, centred index
    double mzmean[16], height[16], area[16], merr[16], ohe[16];
    double hp[4], omzp[4], qtf[4];
    float tpl[FM], tfp[FM], frt[FM], med[FM];  // centred index
    float g_mzlib[16], g_mz[16], g_int[16], g_fin[16], obs_int[16], corr[16], fw[16], ftc[16],
        rowsum[16];
    float iso_mz[4], iso_int[4], spi[4];
    float oi[1], tsum[1], medpk[1], pad0;
    float feat[ADH_NUM_FEATURES + 2];
    int fpeak[16], ord[16], idmap[16];
    uint8_t g_type[16], g_loss[16], g_charge[16], g_number[16], g_pos[16];
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

template <int FM>
__global__ __launch_bounds__(ADH_WAVE) void adh_feature_fast_kernel(
    DevRun run, const CandRec *__restrict__ plan, int32_t n_cand, const float *__restrict__ iso_table,
    int32_t n_iso_cols, adh_scoring_config_t cfg, const unsigned char *__restrict__ scratch,
    const double *__restrict__ wtp_table, DevOut out, int32_t stop_phase) {
    using namespace fast;
    constexpr int RC = FM / 2;
    __shared__ GroupLds<FM> lds[ADH_WAVE / ADH_GS];
    __shared__ double wtp_s[2][64];
    const int lane = threadIdx.x;
    (void)wtp_table;
    {
        // precursor weight table exp(-0.1 * sqrt((s - 2)^2 + (f - 1)^2)): the "expected centre"
        // (S, 1) of precursor_features.py:52-57 does not depend on the candidate
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int idx = lane + 64 * j;
            int sc = idx / 64, f = idx - sc * 64;
            double ds = (double)(sc - 2), df = (double)(f - 1);
            wtp_s[sc][f] = exp(-0.1 * sqrt(ds * ds + df * df));
        }
    }
    const int g = lane / ADH_GS, sub = lane % ADH_GS;
    GroupLds<FM> &L = lds[g];
    const int ci = blockIdx.x * (ADH_WAVE / ADH_GS) + g;
    bool alive = ci < n_cand;
    const CandRec &rec = plan[alive ? ci : 0];
    alive = alive && !(rec.flags & ADH_FLAG_SKIP);
    const unsigned char *block = scratch + rec.scratch_off;
    const uint32_t *header = reinterpret_cast<const uint32_t *>(block);
    int K0 = alive ? (int)header[0] : 0;
    alive = alive && K0 != 0;
    const uint32_t row = rec.row;
    const int Lc = run.cycle_len;
    const int c0 = rec.frame_start / Lc;
    const int F = alive ? rec.frame_stop / Lc - c0 : 0;
    const int c = F / 2;
    const int shift = c - RC;  // f = r + shift
    const int I = alive ? min(n_iso_cols, (int)cfg.top_k_isotopes) : 0;
    const int top_k = out.top_k;
    if (alive && sub == 0 && out.stat_matched_peaks) out.stat_matched_peaks[row] = header[1];
    L.idmap[sub] = sub;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (sub + 16 * j < ADH_NUM_FEATURES + 2) L.feat[sub + 16 * j] = 0.0f;

    float A[FM], B[FM];

    // ================= precursor phase: lanes 0..I-1 hold one isotope row each =================
    const bool iso_lane = alive && sub < I;
    {
        const float2 *pcells =
            reinterpret_cast<const float2 *>(block + adh_scratch_prec_off(rec.k_cap, 1, F));
        FOR_R {
            int f = r + shift;
            bool ok = iso_lane && f >= 0 && f < F;
            float2 v = pcells[ok ? sub * F + f : 0];  // branch-free: clamp the index, mask the value
            A[r] = ok ? v.x : 0.0f;
            B[r] = ok ? v.y : 0.0f;
        }
        float iso_int_l = 0.0f, iso_mz_l = 0.0f;
        double q = 0.0;
        if (iso_lane) {
            iso_int_l = iso_table[(int64_t)row * n_iso_cols + sub];
            double off = (double)sub * 1.0033548350700006 / (double)rec.charge;  // candidate.py:158-163
            iso_mz_l = (float)off + rec.precursor_mz;
            // quadrupole_transfer_function_single (quadrupole.py:261-301), n_scans == 1
            const double *cy = run.cycle + 2 * ((int64_t)rec.obs[0] * run.cycle_scans + rec.scan_start);
            double x = (double)iso_mz_l;
            q = logistic(x, cy[0], 0.2) - logistic(x, cy[1], 0.2);
            L.qtf[sub] = q;
            L.iso_mz[sub] = iso_mz_l;
            L.iso_int[sub] = iso_int_l;
        }
        float sf = 0.0f;
        FOR_R sf += A[r];
        // template contributions (quadrupole.py:304-324)
        FOR_R {
            float a = A[r] * iso_int_l;
            if (iso_lane) L.u.dT[sub][r] = (double)a * q;
        }
        // weighted centre means around (S, 1) (precursor_features.py:52-66)
        double vh = 0, wh = 0, vm = 0, wm = 0;
        bool anyh = false, anym = false;
#pragma unroll 1  // a real loop: unrolling lets hipcc keep 2 x 32 converted values live
        for (int sc = 0; sc < 2; ++sc) {
            FOR_R {
                int f = r + shift;
                bool ok = iso_lane && f >= 0 && f < F;
                double w = ok ? wtp_s[sc][f] : 0.0;
                float a = A[r], b = B[r];
                OPAQUE(a);
                OPAQUE(b);
                if (ok && a > 0.0f) {
                    anyh = true;
                    vh += (double)a * w;
                    wh += w;
                }
                if (ok && b > 0.0f) {
                    anym = true;
                    vm += (double)b * w;
                    wm += w;
                }
                R_FENCE(r);
            }
        }
        if (iso_lane) {
            L.spi[sub] = sf + sf;
            L.hp[sub] = (anyh && wh > 0) ? vh / wh : 0.0;
            L.omzp[sub] = (anym && wm > 0) ? vm / wm : 0.0;
        }
    }
    __syncthreads();
    // qtf mask (candidate.py:287-289) and template rows
    float qmask = 0.0f;
    {
        double qs = 0;
        for (int i = 0; i < I; ++i) qs += L.qtf[i];
        if (I > 0) qmask = (float)(qs / (double)I);
#pragma unroll
        for (int pass = 0; pass < (FM + 15) / 16; ++pass) {
            int r = sub + 16 * pass;
            if (r < FM) {
                double acc = 0;
                for (int i = 0; i < I; ++i) acc += L.u.dT[i][r];
                L.tpl[r] = (float)acc;  // zero outside [0, F)
            }
        }
    }
    __syncthreads();

    // ---- per-candidate tables first, so that the m/z row can die right after its only use
    // template centre of mass (fragment_features.py:20-68); every lane computes it
    double esc, efc;
    {
        double isum = 0, ssum = 0, fsum = 0;
        bool any = false;
#pragma unroll 1  // a real loop: unrolling lets hipcc keep 2 x 32 converted values live
        for (int sc = 0; sc < 2; ++sc) {
            FOR_R {
                float v = L.tpl[r];
                if (v > 0.0f) {
                    any = true;
                    isum += (double)v;
                    ssum += (double)sc * (double)v;
                    fsum += (double)(r + shift) * (double)v;
                }
                R_FENCE(r);
            }
        }
        esc = (any && isum > 0) ? ssum / isum : 0.0;
        efc = (any && isum > 0) ? fsum / isum : 0.0;
    }
    // observation importance (quadrupole.py:327-335), O == 1
    float tsum, oi;
    {
        float st = 0.0f;
        FOR_R {
            st += L.tpl[r];
            R_FENCE(r);
        }
        tsum = st + st;
        float tot = 0.0f + tsum;
        oi = (tot == 0.0f) ? 1.0f / 1.0f : tsum / tot;
    }
    if (sub == 0) {
        L.oi[0] = oi;
        L.tsum[0] = tsum;
    }
    // template frame profile with or_envelope (scoring/utils.py:46-53), frame RTs
#pragma unroll
    for (int pass = 0; pass < (FM + 15) / 16; ++pass) {
        int r = min(sub + 16 * pass, FM - 1);  // (duplicates of the last row write the same value)
        int f = r + shift;
        bool ok = alive && f >= 0 && f < F;
        float x = L.tpl[r] + L.tpl[r];
        float rr = x;
        if (ok && f >= 1 && f < F - 1) {
            float xl = L.tpl[r - 1] + L.tpl[r - 1];
            float xr = L.tpl[r + 1] + L.tpl[r + 1];
            if (x < xl || x < xr) {
                float sm = xl + xr;
                rr = (float)((double)sm / 2.0);
            }
        }
        L.tfp[r] = ok ? rr : 0.0f;
        L.frt[r] = ok ? run.rt[rec.frame_start + f * Lc] : 0.0f;
    }
    // weight table around the template centre (features_utils.py:9-25), centred index
#pragma unroll
    for (int pass = 0; pass < (2 * FM + 15) / 16; ++pass) {
        int idx = min(sub + 16 * pass, 2 * FM - 1);
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

    // ================= fragment phase: lane = fragment =================
    const bool frag_lane = alive && sub < K0;
    {
        const float2 *fcells =
            reinterpret_cast<const float2 *>(block + adh_scratch_frag_off(rec.k_cap));
        FOR_R {
            int f = r + shift;
            bool ok = frag_lane && f >= 0 && f < F;
            float2 v = fcells[ok ? f * K0 + sub : 0];
            A[r] = ok ? v.x * qmask : 0.0f;  // candidate.py:290
            B[r] = ok ? v.y : 0.0f;
        }
    }
    LibRec lrec;
    if (frag_lane) lrec = reinterpret_cast<const LibRec *>(block + 32)[sub];
    // presence (candidate.py:319-329); O == 1 so the sum over observations is the row sum
    float sf = 0.0f;
    FOR_R sf += A[r];
    const float ss = sf + sf;
    bool present = frag_lane && ss > 0.0f;
    const unsigned long long bal = __ballot(present);
    const unsigned gm = (unsigned)((bal >> (g * ADH_GS)) & 0xFFFFull);
    int K = __popc(gm);
    const int kk = __popc(gm & ((1u << sub) - 1u));
    const int n_present = K;
    if (K < 2) {  // candidate.py:323
        alive = false;
        present = false;
        K = 0;
    }
    // ---- weighted centre means of both channels (features_utils.py:9-37)
    double ohe, omz;
    {
        double vo = 0, wo = 0, vm = 0, wm = 0;
        bool anyo = false, anym = false;
#pragma unroll 1  // a real loop: unrolling lets hipcc keep 2 x 32 converted values live
        for (int sc = 0; sc < 2; ++sc) {
            FOR_R {
                double w = L.wt[sc][r];
                float a = A[r], b = B[r];
                OPAQUE(a);
                OPAQUE(b);
                if (a > 0.0f) {
                    anyo = true;
                    vo += (double)a * w;
                    wo += w;
                }
                if (b > 0.0f) {
                    anym = true;
                    vm += (double)b * w;
                    wm += w;
                }
                R_FENCE(r);
            }
        }
        ohe = (anyo && wo > 0) ? vo / wo : 0.0;
        omz = (anym && wm > 0) ? vm / wm : 0.0;
    }
    if (present) {
        L.g_mzlib[kk] = lrec.mz_library;
        L.g_mz[kk] = lrec.mz;
        L.g_fin[kk] = lrec.intensity;  // raw intensity, normalised below
        L.g_type[kk] = lrec.type;
        L.g_loss[kk] = lrec.loss_type;
        L.g_charge[kk] = lrec.charge;
        L.g_number[kk] = lrec.number;
        L.g_pos[kk] = lrec.position;
        L.rowsum[kk] = ss;
    }
    __syncthreads();
    // fragment intensities: apply_mask renormalisation + the second one of fragment_features.py:218
    float g_int_l = 0.0f, g_fin_l = 0.0f;
    {
        float sum1 = 0.0f;
        for (int j = 0; j < K; ++j) sum1 += L.g_fin[j];
        if (present) {
            g_int_l = lrec.intensity / sum1;
            L.g_int[kk] = g_int_l;
        }
    }
    __syncthreads();
    {
        float sum2 = 0.0f;
        for (int j = 0; j < K; ++j) sum2 += L.g_int[j];
        if (present) g_fin_l = g_int_l / sum2;
    }
    if (present) L.g_fin[kk] = g_fin_l;  // raw values were consumed before the last barrier
    if (stop_phase == 3 || stop_phase == 4) return;

    // ---- frame profile row, envelope, quantification (fragment_features.py:240-273)
    float P[FM];
    FOR_R P[r] = A[r] + A[r];  // frame_profile_2d: sum over the two identical scan slots
    double area = 0.0;
    float obs_int = 0.0f;
    {
        center_envelope<FM>(P, F);
        const int qw = min(c - 1, (int)cfg.quant_window);
        double ar = 0.0;
#pragma unroll
        for (int r = 1; r < FM - 1; ++r) {
            if (r >= RC - qw && r + 1 <= RC + qw) {
                float sm = P[r + 1] + P[r];
                float drt = L.frt[r + 1] - L.frt[r];
                float m = sm * drt;
                ar += (double)m * 0.5;
            }
        }
        area = ar * (double)qw;
        FOR_R {
            if (r >= RC - qw && r <= RC + qw) obs_int += P[r];
        }
        if (cfg.quant_all) {
            FOR_R P[r] = A[r] + A[r];  // np.sum(axis=1) made a copy: the profile itself is untouched
        }
    }
    if (present) {
        // importance-weighted means, O == 1 (fragment_features.py:311-336)
        bool m = ohe > 0;
        float w32 = m ? oi : oi * 0.0f;
        float ws = 0.0f + w32;
        double w = (double)w32 / ((double)ws + 1e-20);
        double m1 = 0, m2 = 0;
        if (w > 0) {
            double msum = 0.0 + w;
            double lw = w / msum;
            m1 = omz * lw;
            m2 = ohe * lw;
        }
        L.ohe[kk] = ohe;
        L.mzmean[kk] = m1;
        L.height[kk] = m2;
        L.merr[kk] = (m1 - (double)lrec.mz) / (double)lrec.mz * 1e6;  // fragment_features.py:387
        L.area[kk] = area;
        L.obs_int[kk] = obs_int;
        int rk = 0;
        for (int j = 0; j < K; ++j) {
            float ib = L.g_int[j];
            rk += (ib > g_int_l) || (ib == g_int_l && j > kk);
        }
        L.ord[rk] = kk;  // position in argsort(intensity)[::-1]
    }
    __syncthreads();
    if (stop_phase == 5) return;

    Assemble asmv;
    asmv.run = &run;
    asmv.rec = &rec;
    asmv.featv = L.feat;
    asmv.iso_int = L.iso_int; asmv.iso_mz = L.iso_mz; asmv.spi = L.spi; asmv.oi = L.oi;
    asmv.tsum = L.tsum; asmv.rowsum = L.rowsum; asmv.g_fin = L.g_fin; asmv.g_int = L.g_int;
    asmv.obs_int = L.obs_int; asmv.corr = L.corr; asmv.ftc = L.ftc; asmv.fw = L.fw;
    asmv.medpk = L.medpk; asmv.omzp = L.omzp; asmv.hp = L.hp; asmv.ohe = L.ohe; asmv.area = L.area;
    asmv.height = L.height; asmv.merr = L.merr; asmv.kmap = L.idmap; asmv.ord = L.ord;
    asmv.g_type = L.g_type; asmv.g_pos = L.g_pos;
    asmv.n_present = n_present; asmv.K0 = K0; asmv.top3 = 0.0f;
    if (alive && sub == 0) feat::assemble_part1(asmv, I, 1, K);
    if (stop_phase == 6) return;

    // ================= profile features (profile_features.py:18-206), experimental_xic =======
    // intensity_slice = frame profile summed over the single observation = P
    {
        // normalize_profiles (scoring_utils.py:71-117): centre +- 1 are registers 15, 16, 17
        float sm = 0.0f;
        sm += P[RC - 1];
        sm += P[RC];
        sm += P[RC + 1];
        const double cn = (double)sm / 3.0;
        // median over fragments per cycle (scoring_utils.py:120-152): 16x16 transposes via LDS
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
    __syncthreads();
    float corr_l = 0.0f, ftc_l = 0.0f, fw_l = 0.0f;
    int fpeak_l = 0;
    {
        // correlation_coefficient (scoring_utils.py:14-68)
        float sx = 0.0f;
        FOR_R {
            sx += L.med[r];
            R_FENCE(r);
        }
        const float mx = (float)((double)sx / (double)F);
        float sxx = 0.0f, sy = 0.0f;
        FOR_R {
            int f = r + shift;
            bool ok = f >= 0 && f < F;
            float xm = ok ? L.med[r] - mx : 0.0f;
            sxx += xm * xm;
            R_FENCE(r);
        }
        const double var_x = (double)sxx / (double)F;
        FOR_R sy += P[r];
        const float my = (float)((double)sy / (double)F);
        float sxy = 0.0f, syy = 0.0f;
        FOR_R {
            int f = r + shift;
            bool ok = f >= 0 && f < F;
            float xm = ok ? L.med[r] - mx : 0.0f;
            float ym = ok ? P[r] - my : 0.0f;
            sxy += xm * ym;
            R_FENCE(r);
        }
        FOR_R {
            int f = r + shift;
            bool ok = f >= 0 && f < F;
            float ym = ok ? P[r] - my : 0.0f;
            syy += ym * ym;
        }
        const double cov = (double)sxy / (double)F;
        const double var_y = (double)syy / (double)F;
        const double var_xy = var_x * var_y;
        corr_l = (var_xy == 0) ? 0.0f : (float)(cov / sqrt(var_xy));

        // fragment vs template frame correlation (scoring/utils.py:574-647)
        float syt = 0.0f;
        FOR_R {
            syt += L.tfp[r];
            R_FENCE(r);
        }
        const float ym = syt / (float)F;
        float qy = 0.0f;
        FOR_R {
            int f = r + shift;
            bool ok = f >= 0 && f < F;
            float d = ok ? L.tfp[r] - ym : 0.0f;
            qy += d * d;
            R_FENCE(r);
        }
        const float ysd = sqrtf(qy / (float)F);
        const float xmn = sy / (float)F;
        float qx = 0.0f, dot = 0.0f;
        FOR_R {
            int f = r + shift;
            bool ok = f >= 0 && f < F;
            float d = ok ? P[r] - xmn : 0.0f;
            qx += d * d;
        }
        const float xsd = sqrtf(qx / (float)F);
        FOR_R {
            int f = r + shift;
            bool ok = f >= 0 && f < F;
            float dx = ok ? P[r] - xmn : 0.0f;
            float dy = ok ? L.tfp[r] - ym : 0.0f;
            dot += dx * dy;
            R_FENCE(r);
        }
        const float cv = dot / (float)F;
        const float smm = xsd * ysd;
        ftc_l = (float)((double)cv / ((double)smm + 1e-12));

        // FWHM in RT (profile_features.py:117-146) and apex (profile_features.py:192-193)
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
        const float rt_width = alive ? run.rt[rec.frame_stop - 1] - run.rt[rec.frame_start] : 0.0f;
        fw_l = (float)(frac * (double)rt_width);
        fpeak_l = am;
    }
    if (present) {
        L.corr[kk] = corr_l;
        L.ftc[kk] = ftc_l;
        L.fw[kk] = fw_l;
        L.fpeak[kk] = fpeak_l;
    }
    __syncthreads();
    if (alive && sub == 0) {
        // median apex (profile_features.py:196-198)
        int lo_v = 0, hi_v = 0, r_lo = (K - 1) / 2, r_hi = K / 2;
        for (int a = 0; a < K; ++a) {
            int va = L.fpeak[a];
            int rk = 0;
            for (int b = 0; b < K; ++b) {
                int vb = L.fpeak[b];
                rk += (vb < va) || (vb == va && b < a);
            }
            if (rk == r_lo) lo_v = va;
            if (rk == r_hi) hi_v = va;
        }
        double m = (K & 1) ? (double)hi_v : (double)(lo_v + hi_v) / 2.0;
        L.medpk[0] = (float)m;
        int n3 = min(K, 3);
        float sm = 0;
        for (int i = 0; i < n3; ++i) sm += L.corr[L.ord[i]];
        asmv.top3 = (float)((double)sm / (double)n3);
        feat::assemble_part2(asmv, 1, K, F);
    }
    __syncthreads();

    // ---- output row (candidate.py:403-481)
    if (alive) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int idx = sub + 16 * j;
            if (idx < ADH_NUM_FEATURES) out.features[(int64_t)row * ADH_NUM_FEATURES + idx] = L.feat[idx];
        }
        if (cfg.collect_fragments && present && kk < top_k) {
            const int64_t o = (int64_t)row * top_k + kk;
            out.fragment_precursor_idx[o] = rec.precursor_idx;
            out.fragment_rank[o] = rec.rank;
            out.fragment_mz_library[o] = lrec.mz_library;
            out.fragment_mz[o] = lrec.mz;
            out.fragment_mz_observed[o] = (float)L.mzmean[kk];
            out.fragment_height[o] = (float)L.height[kk];
            out.fragment_intensity[o] = (float)area;
            out.fragment_mass_error[o] = (float)L.merr[kk];
            out.fragment_correlation[o] = corr_l;
            out.fragment_position[o] = lrec.position;
            out.fragment_number[o] = lrec.number;
            out.fragment_type[o] = lrec.type;
            out.fragment_charge[o] = lrec.charge;
            out.fragment_loss_type[o] = lrec.loss_type;
        }
        if (sub == 0) out.valid[row] = 1;
    }
}
