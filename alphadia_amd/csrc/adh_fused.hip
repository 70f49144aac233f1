// adh_fused.hip - gather + features in ONE kernel for the common candidate shape.
//
// The two-kernel path (adh_gather_kernel -> scratch block in HBM -> adh_feature_fast_kernel) writes
// every XIC tile to HBM (zero fill + cells) and reads it back: half of the traffic of a scoring pass
// existed only because gather and features were two kernels.  Here a candidate never leaves the
// compute unit between
//   FragmentContainer.slice / filter / sort   alphadia/search/jitclasses/fragment_container.py:56-102
//   AlphaRawJIT.get_dense (x2)                alphadia/search/jitclasses/alpharaw_jit.py:208-337
//   MS1 observation collapse                  alphadia/search/scoring/containers/candidate.py:248-269
// and Candidate.process after get_dense       alphadia/search/scoring/containers/candidate.py:278-481
//
//   * four candidates per 64-lane wavefront, 16 lanes each.  Lane k < 12 owns fragment k (ascending
//     m/z), lane 12 + i owns isotope i: all K + I m/z windows of a candidate are gathered at once and
//     the per-row work of the precursor features runs in the same instructions as the fragments'
//   * a lane folds the peaks of its window into a private column of an LDS tile [FM][16] (the
//     reference's running intensity-weighted m/z, alpharaw_jit.py:299-335, in the reference's order:
//     see adh_gather.hip), stored CENTRED - row r holds cycle f = r - FM/2 + F/2 - so that the
//     rows then move into VGPRs with constant offsets and no masks; cells outside [0, F) stay zero
//   * from there on the arithmetic is that of adh_feature_fast_kernel (same summation order, float64
//     where Numba types the expression as float64): bit-identical rows.  Differences are of form
//     only: branch-free envelope recurrences (a min with an untouched zero cell is a no-op), cycle
//     masks only on the outer registers of a class (3 <= FM - F < 4 for every candidate of a launch),
//     float64 divisions by a common divisor through one shared reciprocal refinement
//   * LDS is one union per candidate: selection scratch -> tile -> feature arrays
//
// Eligibility (adh_plan_rec_kernel): one or two observations (both settings of quant_all), 3 <= F <= 32, k_cap <= 12,
// I <= 4, library slice of at most 64 fragments, experimental_xic = True, a single MS1 row per cycle.  Everything
// else runs through the two-kernel path: in particular candidates that keep 13 or more fragments - a group has 16
// lanes, 12 + 4 is the split that covers the reference's defaults (top_k_fragments 12, default.yaml:185; 3 or 4
// isotopes); longer lists only occur when a transfer library with more than 12 fragments per precursor is
// requantified with top_k_fragments = 9999 (transfer_library_requantification_handler.py:117-124).
#include "adh_device.h"
#include "adh_feature_common.h"

#include <type_traits>

#define ADH_FUSED_ISO0 12      // first isotope lane of a 16-lane group
#ifndef ADH_FUSED_TW3
#define ADH_FUSED_TW3 15       // tile columns of the three-isotope launches (A/B of round 6: 17)
#endif
#define ADH_FUSED_NLIB 64      // longest library slice handled here
#ifndef ADH_FUSED_WAVES
#define ADH_FUSED_WAVES 3      // wavefronts per SIMD the register budget is held to
#endif
#ifndef ADH_FUSED_FM3
#define ADH_FUSED_FM3 24       // longest one-observation rows that run at ADH_FUSED_WAVES; longer ones at ADH_FUSED_WAVES2
#endif
#ifndef ADH_FUSED_WAVES2
#define ADH_FUSED_WAVES2 2     // ... of the two-observation kernels (3: spills, 15 % slower; same-box A/B)
#endif
#ifndef ADH_FUSED_SCALAR
#define ADH_FUSED_SCALAR 1     // developer switch: 0 skips the feature assembly of single lanes (wrong results; what it costs)
#endif
#ifndef ADH_FUSED_AHEAD
#define ADH_FUSED_AHEAD 0      // 1: the entries of the step after next are requested while a step is folded; 0: step by step
#endif
#ifndef ADH_FUSED_NPAIR
#define ADH_FUSED_NPAIR 2      // m/z bins of a window whose table words and entries are requested together
#endif
#ifndef ADH_FUSED_EB
#define ADH_FUSED_EB 4         // entries per step of a gather task (even)
#endif

namespace fused {

constexpr int GS = 16;
// Tile columns TW (a template parameter): 12 fragment lanes + the isotope lanes.  15 for up to three isotopes
// (the search: extraction_handler.py:372) - an odd row stride, and lane 15 has no window; 16 for four (round 4:
// top_k_isotopes = 4 is the class default, config.py:78, which multiplex / transfer-library requantification
// score with).  One width for both cost the three-isotope candidates of the headline 4 % (same-box A/B: the
// even stride of 32 words puts the scattered cell updates of the gather on two sets of banks).
constexpr int ISO0 = ADH_FUSED_ISO0;
constexpr int NLIB = ADH_FUSED_NLIB;

// cycles F of a launch with FM registers: FMIN <= F <= FM; registers [R0, R1) hold a valid cycle for
// every such F (f = r - FM/2 + F/2)
template <int FM>
struct Rng {
    static constexpr int FMIN = FM == 8 ? 3 : FM - 3;
    static constexpr int R0 = FM / 2 - FMIN / 2;
    static constexpr int R1 = FM / 2 + (FMIN + 1) / 2;
};

// a library record (LibRec, 32 bytes) as the registers it travels in: a = (mz_library, mz, intensity,
// type | loss_type << 8 | charge << 16 | number << 24), b = position | cardinality << 8 | (index inside the
// library slice) << 16
struct __attribute__((aligned(16))) RawRec {
    uint4 a;
    uint32_t b, pad[3];
};
static_assert(sizeof(RawRec) == sizeof(LibRec), "RawRec mirrors LibRec");
__device__ __forceinline__ RawRec load_rec(const LibRec *p) {
    RawRec r;
    r.a = *reinterpret_cast<const uint4 *>(p);
    r.b = reinterpret_cast<const uint32_t *>(p)[4] & 0xFFFFu;
    return r;
}
__device__ __forceinline__ float rec_mz(const RawRec &r) { return __uint_as_float(r.a.y); }
__device__ __forceinline__ float rec_intensity(const RawRec &r) { return __uint_as_float(r.a.z); }
__device__ __forceinline__ int rec_type(const RawRec &r) { return (int)(r.a.w & 0xFFu); }
__device__ __forceinline__ int rec_position(const RawRec &r) { return (int)(r.b & 0xFFu); }
__device__ __forceinline__ int rec_cardinality(const RawRec &r) { return (int)((r.b >> 8) & 0xFFu); }

// What a candidate keeps in LDS across the gather of its second observation (which takes the tile = the
// union below): with one observation these arrays live inside the union's feature part, at no cost
template <int FM, int NO>
struct __attribute__((aligned(16))) Keep {
    double hp[4], omzp[4];
    float frt[FM];             // frame RTs, centred
    float tpl_next[FM];        // template row of the observation still to come
    float iso_mz[4], iso_int[4], spi[4];
};
struct __attribute__((aligned(16))) NoKeep {};

template <int FM, int NO, int TW>
struct __attribute__((aligned(16))) GroupLds {
    union {
        struct {  // fragment selection
            float l_int[NLIB], l_mz[NLIB];
            int l_ok[NLIB], l_rank[NLIB];
            RawRec sel[ISO0];
        } s;
        float2 tile[FM][TW];  // gathered cells, [centred cycle][lane]
        struct {
            union {
                double dT[4][FM];    // isotope contributions to the template
                double wti[2][FM];   // precursor weights (centred, masked); written once dT is consumed
                float nrmT[16][17];  // transpose buffer for the per-cycle median (padded rows)
                struct {             // per-fragment terms of the feature sums: [fragment][sum]
                    double t64[16][6];
                    float t32[16][6];
                } at;
            } u;
            double wt[2][FM];  // [scan slot][centred cycle]: exp weights around the template centre
            double merr[16];
            double red64[14];
            float tpl[FM], tfp[FM], med[FM];
            float g_int[16], g_fin[16], corr[16];
            int fpeak[16][NO];
            float oi[NO];
            float red32[8];
            float feat[ADH_NUM_FEATURES + 2];
            int ord[16];
            int medlo[NO], medhi[NO];
            typename std::conditional<NO == 1, Keep<FM, NO>, NoKeep>::type keep1;
        } f;
    } u;
    typename std::conditional<NO == 1, NoKeep, Keep<FM, NO>>::type keep2;
    __device__ __forceinline__ Keep<FM, NO> &keep() {
        if constexpr (NO == 1) return u.f.keep1;
        else return keep2;
    }
};

#define FU_OPAQUE(x) __asm__ volatile("" : "+v"(x))
#define FU_FOR_R _Pragma("unroll") for (int r = 0; r < FM; ++r)
#define FU_FENCE(r)                                               \
    do {                                                          \
        if ((((r)) & 7) == 7) __asm__ volatile("" ::: "memory"); \
    } while (0)
// does register r hold a cycle of this candidate?  (compile-time true for the inner registers)
#define FU_OK(r) (((r) >= Rng<FM>::R0 && (r) < Rng<FM>::R1) ? true : ((unsigned)((r) + shift) < (unsigned)F))

// value of the lane `n` lanes down in the same 16-lane row; `fill` where there is none
template <int N>
__device__ __forceinline__ float row_shr(float fill, float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x110 + N, 0xF, 0xF, false));
}
// maximum over all lower lanes of the row (-inf for lane 0)
__device__ __forceinline__ float row_prefix_max_excl(float x) {
    float v = row_shr<1>(-INFINITY, x);
    v = fmaxf(v, row_shr<1>(-INFINITY, v));
    v = fmaxf(v, row_shr<2>(-INFINITY, v));
    v = fmaxf(v, row_shr<4>(-INFINITY, v));
    v = fmaxf(v, row_shr<8>(-INFINITY, v));
    return v;
}

// value of another lane of the same 16-lane row: rotation by N lanes (whichever lane it is, the lane
// index travels with it, so callers never depend on the direction)
template <int N>
__device__ __forceinline__ int row_ror(int x) {
    return __builtin_amdgcn_mov_dpp(x, 0x120 + N, 0xF, 0xF, false);  // (every lane has a source: no `old` value to set up)
}
#define FU_FOR_OTHER_LANES(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

__device__ __forceinline__ double logistic(double x, double mu, double sigma) {
    double a = (x - mu) / sigma;
    return 1.0 / (1.0 + exp(-a));
}

// x / d for many x and one d > 0: the reciprocal refinement of the IEEE division sequence
// (v_rcp_f64 + two Newton steps) is shared, the per-quotient part (q0 = x r, e = x - d q0,
// q = q0 + e r) is what the compiler emits for x / d when no operand needs rescaling
struct Recip {
    double d, r;
    __device__ __forceinline__ explicit Recip(double dd) : d(dd) {
        double y = __builtin_amdgcn_rcp(dd);
        double e = __builtin_fma(-dd, y, 1.0);
        y = __builtin_fma(y, e, y);
        e = __builtin_fma(-dd, y, 1.0);
        y = __builtin_fma(y, e, y);
        r = y;
    }
    __device__ __forceinline__ double div(double x) const {
        double q = x * r;
        double e = __builtin_fma(-d, q, x);
        return __builtin_fma(e, r, q);
    }
};

// center_envelope_1d (fragment_features.py:71-159) on a centred register row of non-negative
// values.  (a) every step is min(x, mean of the two inner neighbours): float32 additions of the
// reference, the halving and the float64 minimum are exact, so float32 arithmetic gives the same
// bits; (b) a step on a register outside [0, F) is min(0, >= 0) = 0: no predicate is needed beyond
// the parity of F, which decides where the left chain starts.
template <int FM>
__device__ __forceinline__ void center_envelope(float (&x)[FM], int F) {
    constexpr int RC = FM / 2;
    const bool odd = F & 1;
    // right chain: registers RC + i, i >= 1, seeded with (x[RC + 1] + x[RC]) / 2 (odd) or x[RC] (even)
    {
        float right = odd ? (x[RC + 1] + x[RC]) * 0.5f : x[RC];
#pragma unroll
        for (int i = 1; RC + i < FM; ++i) {
            x[RC + i] = (right < x[RC + i]) ? right : x[RC + i];  // (min of two non-negative numbers: no NaN to quiet)
            right = (x[RC + i] + x[RC + i - 1]) * 0.5f;
        }
    }
    // left chain: odd F walks RC - i from the centre register RC, even F walks RC - 1 - i from RC - 1.  In place:
    // step i reads and writes the register of the lane's parity, the other parity's register passes through
    {
        float prev = odd ? x[RC] : x[RC - 1];
        float left = odd ? (x[RC - 1] + x[RC]) * 0.5f : x[RC - 1];
#pragma unroll
        for (int i = 1; i <= RC; ++i) {
            const float xe = (RC - 1 - i >= 0) ? x[RC - 1 - i > 0 ? RC - 1 - i : 0] : 0.0f;
            const float cur = odd ? x[RC - i] : xe;
            const float nv = (left < cur) ? left : cur;
            left = (nv + prev) * 0.5f;
            prev = nv;
            x[RC - i] = odd ? nv : x[RC - i];
            if (RC - 1 - i >= 0) x[RC - 1 - i] = odd ? x[RC - 1 - i] : nv;
        }
    }
}

// frame-profile statistics of one fragment row against the template frame profile: fragment-vs-
// template correlation (scoring/utils.py:574-647), FWHM in RT (profile_features.py:117-146) and apex
// (profile_features.py:192-193)
template <int FM>
__device__ __forceinline__ void profile_stats(const float (&P)[FM], const float *tfp, int F, int shift,
                                              float rt_width, float &ftc, float &fw, int &fpeak) {
    const float Ff = (float)F;
    float syt = 0.0f;
    FU_FOR_R {
        syt += tfp[r];
        FU_FENCE(r);
    }
    const float ym = syt / Ff;
    float qy = 0.0f;
    FU_FOR_R {
        float d = tfp[r] - ym;
        d = FU_OK(r) ? d : 0.0f;
        qy += d * d;
        FU_FENCE(r);
    }
    const float ysd = sqrtf(qy / Ff);
    float sy = 0.0f;
    FU_FOR_R sy += P[r];
    const float xmn = sy / Ff;
    float qx = 0.0f, dot = 0.0f;
    FU_FOR_R {
        float d = P[r] - xmn;
        d = FU_OK(r) ? d : 0.0f;
        qx += d * d;
    }
    const float xsd = sqrtf(qx / Ff);
    FU_FOR_R {
        float dx = P[r] - xmn;
        float dy = tfp[r] - ym;
        dx = FU_OK(r) ? dx : 0.0f;  // (one zero factor is enough)
        dot += dx * dy;
        FU_FENCE(r);
    }
    const float cv = dot / Ff;
    const float smm = xsd * ysd;
    ftc = (float)((double)cv / ((double)smm + 1e-12));
    // first maximum over the valid cycles; values are non-negative and cells outside [0, F) are 0,
    // so "first strictly greater" started from the first valid cycle is all that is needed
    float mxv = -1.0f;
    int am = 0;
    FU_FOR_R {
        const bool up = FU_OK(r) && P[r] > mxv;
        mxv = up ? P[r] : mxv;
        am = up ? r : am;
    }
    const float half_max = mxv * 0.5f;  // exact halving; (double)P > (double)mxv / 2 is the same comparison
    int n_above = 0;
    FU_FOR_R n_above += (FU_OK(r) && P[r] > half_max) ? 1 : 0;
    const double frac = (double)n_above / (double)F;
    fw = (float)(frac * (double)rt_width);
    fpeak = am + shift;
}

// One gather task = one m/z bin of a window inside one GROUP of cycle blocks (adh_device.h): the entries of
// the blocks a candidate spans are one contiguous run there, sorted by (cycle, m/z), and their first / last
// table words sit next to each other.  A candidate of up to 32 cycles meets one or two groups, a window one or
// two bins (rarely more): per window ~2 table lines and ~2 entry lines travel, where the block-by-block
// layout needed ~5.  A scattered load costs the memory system one line per lane: that, not arithmetic or the
// length of the dependency chain, is what bounds the gather (measured: coarser bins, larger blocks, more
// loads in flight and fewer instructions all left its time where it was or made it worse).
constexpr int EB = ADH_FUSED_EB;  // entries fetched per step
constexpr int NPAIR = ADH_FUSED_NPAIR;  // bins of a window in flight together
struct __attribute__((packed, aligned(4))) Tab4 {
    uint32_t x, y, z, w;
};
struct __attribute__((packed, aligned(8))) Ent2 {
    uint2 a, b;
};
struct Task {
    uint32_t idx, end;
    uint32_t bin_bits;        // (bin0 + bin) << ADH_BIN_SHIFT: the upper bits of every m/z of the bin
    int grp_base;             // first cycle of the group
    uint32_t f_lo, nf;        // group-relative cycles [f_lo, f_lo + nf)
    bool first_bin;           // no earlier bin of the window has touched the cells
    uint2 e[EB], n[EB];       // entries of the current and of the next step
};
// the window as the gather sees it: bins and the float32 bounds as bit patterns (positive floats order
// like their patterns; m/z > excl is pattern >= pattern(excl) + 1)
struct WinBits {
    int b_lo, b_hi;
    uint32_t lo_u, hi_u;
};
__device__ __forceinline__ WinBits win_bits(const gather::Window &w) {
    WinBits q;
    q.b_lo = w.b_lo;
    q.b_hi = w.b_hi;
    const uint32_t lo = (w.lo > 0.0f) ? __float_as_uint(w.lo) : 0u;
    const uint32_t ex = (w.excl > 0.0f) ? __float_as_uint(w.excl) + 1u : 0u;  // (-inf: no earlier window)
    q.lo_u = max(lo, ex);
    q.hi_u = (w.hi > 0.0f) ? __float_as_uint(w.hi) : 0u;
    return q;
}

// entries of the blocks sb0 .. sb1 of bin b of (group, row): [t[b * ADH_SUB + sb0], t[b * ADH_SUB + sb1 + 1])
__device__ __forceinline__ void task_begin(const DevRun &run, bool on, int row, int grp, int sb0, int sb1, int b,
                                           int b_lo, int c0, int F, Task &k) {
    const int gs = run.block_shift + ADH_SUB_SHIFT;
    k.grp_base = grp << gs;
    const int f_lo = max(c0, k.grp_base) - k.grp_base;
    const int f_hi = min(c0 + F, k.grp_base + (1 << gs)) - k.grp_base;
    k.f_lo = (uint32_t)f_lo;
    k.nf = (uint32_t)max(f_hi - f_lo, 0);
    k.bin_bits = (uint32_t)(run.bin0 + b) << ADH_BIN_SHIFT;
    k.first_bin = b == b_lo;
    k.idx = 0;
    k.end = 0;
    if (on) {
        const uint32_t *t = adh_tab_row(run, row, grp << ADH_SUB_SHIFT) + (int64_t)b * ADH_SUB + sb0;
        const int d = sb1 + 1 - sb0;  // 1 .. ADH_SUB
        if (d <= 3) {
            const Tab4 v = *reinterpret_cast<const Tab4 *>(t);  // words sb0 .. sb0 + 3 (the table has 4 spare words)
            k.idx = v.x;
            k.end = d == 1 ? v.y : (d == 2 ? v.z : v.w);
        } else {
            k.idx = t[0];
            k.end = t[d];
        }
    }
}

__device__ __forceinline__ void load2(const uint2 *ent, uint32_t i, uint2 &a, uint2 &b) {
    const Ent2 p = *reinterpret_cast<const Ent2 *>(ent + i);
    a = p.a;
    b = p.b;
}

__device__ __forceinline__ void task_fetch(const uint2 *ent, Task &k) {
    const uint32_t n = k.end - k.idx;  // (idx <= end)
#pragma unroll
    for (int u = 0; u < EB; u += 2) {
        if (n > (uint32_t)u) load2(ent, k.idx + u, k.e[u], k.e[u + 1]);
    }
#if ADH_FUSED_AHEAD
#pragma unroll
    for (int u = 0; u < EB; u += 2) {
        if (n > (uint32_t)(EB + u)) load2(ent, k.idx + EB + u, k.n[u], k.n[u + 1]);
    }
#endif
}

// cells: the lane's column of the tile, cells[r * TW] = centred row r; roff = FM/2 - F/2 - c0
template <int TW>
__device__ __forceinline__ void task_run(const uint2 *ent, const WinBits &w, Task &k, float2 *cells, int roff,
                                         uint32_t &hits) {
    uint32_t idx = k.idx;
    const uint32_t end = k.end;
    if (idx >= end) return;
    int cur = -1;  // open cell (group-relative cycle), -1: none
    float acc_i = 0.0f, acc_m = 0.0f;
    while (idx < end) {
#if ADH_FUSED_AHEAD
        // this step's entries were requested two steps ago; the step after next is requested now
        uint2 e[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) {
            e[u] = k.e[u];
            k.e[u] = k.n[u];
        }
        {
            const uint32_t left = end - idx;
#pragma unroll
            for (int u = 0; u < EB; u += 2) {
                if (left > (uint32_t)(2 * EB + u)) load2(ent, idx + 2 * EB + u, k.n[u], k.n[u + 1]);
            }
        }
#else
        uint2 e[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) e[u] = k.e[u];
        {
            const uint32_t left = end - idx;
#pragma unroll
            for (int u = 0; u < EB; u += 2) {
                if (left > (uint32_t)(EB + u)) load2(ent, idx + EB + u, k.e[u], k.e[u + 1]);
            }
        }
#endif
#pragma unroll
        for (int u = 0; u < EB; ++u) {
            const uint32_t i = idx + (uint32_t)u;
            if (i >= end) break;
            const uint32_t cyc = e[u].x >> ADH_BIN_SHIFT;
            const uint32_t bits = k.bin_bits | (e[u].x & ((1u << ADH_BIN_SHIFT) - 1u));
            if (cyc - k.f_lo >= k.nf || bits < w.lo_u || bits > w.hi_u) continue;
            if ((int)cyc != cur) {
                if (cur >= 0) cells[(cur + k.grp_base + roff) * TW] = make_float2(acc_i, acc_m);
                float2 v = make_float2(0.0f, 0.0f);
                if (!k.first_bin) v = cells[((int)cyc + k.grp_base + roff) * TW];  // continue from an earlier bin
                acc_i = v.x;
                acc_m = v.y;
                cur = (int)cyc;
            }
            gather::fold(acc_i, acc_m, __uint_as_float(bits), __uint_as_float(e[u].y));
            ++hits;
        }
        idx += EB;
    }
    if (cur >= 0) cells[(cur + k.grp_base + roff) * TW] = make_float2(acc_i, acc_m);
}

// Precursor features 4-16 (precursor_features.py:13-102; feat::assemble_precursor is the scalar form over LDS
// arrays): the per-isotope values are read once, every loop over the (at most three) isotopes is unrolled
// with `i < I` masks - same terms, same order.  wme_term[i] = mass error of isotope i x its intensity, or 0
// where the isotope was not observed (computed by the isotope's own lane).
template <int NO>
__device__ __forceinline__ void precursor_features(float *ft, int I, const float *iso_int_p, const float *iso_mz_p,
                                                   const float *spi_p, const double *hp_p, const double *wme_term_p,
                                                   const float (&oi)[NO]) {
    constexpr int NI = 4;
    float ii[NI], mz[NI], spi[NI];
    double hp[NI], wt[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        ii[i] = iso_int_p[i];
        mz[i] = iso_mz_p[i];
        spi[i] = spi_p[i];
        hp[i] = hp_p[i];
        wt[i] = wme_term_p[i];
    }
    int amax = 0;
#pragma unroll
    for (int i = 1; i < NI; ++i)
        if (i < I && ii[i] > (amax == 0 ? ii[0] : (amax == 1 ? ii[1] : ii[2]))) amax = i;
    float w4 = 0, w5 = 0, f6 = 0, f7 = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (i < I) {
            float a = 0;
#pragma unroll
            for (int o = 0; o < NO; ++o) a += spi[i] * oi[o];
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
    for (int i = 0; i < NI; ++i)
        if (i < I) wme += wt[i];
    ft[8] = (float)wme;
    ft[9] = (float)fabs(wme);
    ft[10] = (float)((double)mz[0] + wme * 1e-6 * (double)mz[0]);
    ft[11] = (float)hp[0];
    ft[12] = (float)(amax == 0 ? hp[0] : (amax == 1 ? hp[1] : (amax == 2 ? hp[2] : hp[3])));
    {
        double a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i < I) a += hp[i];
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i < I) b += hp[i] * (double)ii[i];
        ft[13] = (float)a;
        ft[14] = (float)b;
    }
    {
        // save_corrcoeff (scoring/utils.py:478-510): (f32, f32) and (f32, f64)
        float sx = 0, sy = 0;
        double sh = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i < I) sx += ii[i];
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i < I) sy += spi[i];
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i < I) sh += hp[i];
        const Recip rI((double)max(I, 1));
        const float xb = (float)rI.div((double)sx), yb = (float)rI.div((double)sy);
        const double hb = rI.div(sh);
        float num = 0, sxx = 0, syy = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i < I) num += (ii[i] - xb) * (spi[i] - yb);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i < I) sxx += (ii[i] - xb) * (ii[i] - xb);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i < I) syy += (spi[i] - yb) * (spi[i] - yb);
        const float den = sqrtf(sxx * syy);
        ft[15] = (float)((double)num / ((double)den + 1e-12));
        double numd = 0, shh = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i < I) numd += (double)(ii[i] - xb) * (hp[i] - hb);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i < I) shh += (hp[i] - hb) * (hp[i] - hb);
        const double dend = sqrt((double)sxx * shh);
        ft[16] = (float)(numd / (dend + 1e-12));
    }
}

// one gather pass of a candidate: every lane with a window folds the peaks of its (window, cycle row) into
// its column of the zeroed tile, cycle block after cycle block
template <int FM, int TW>
__device__ __forceinline__ void gather_pass(const DevRun &run, const WinBits &wb, bool task_on, int task_row, bool alive,
                                            int c0, int F, float2 (*tile)[TW], int sub, uint32_t &hits) {
    {
        float4 *z = reinterpret_cast<float4 *>(&tile[0][0]);
        constexpr int N4 = FM * TW / 2;
#pragma unroll
        for (int j = 0; j < (N4 + GS - 1) / GS; ++j)
            if (j * GS + sub < N4) z[j * GS + sub] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const int bs = run.block_shift;
    const int blk0 = c0 >> bs, blk1 = alive ? (c0 + F - 1) >> bs : blk0 - 1;  // first / last cycle block
    float2 *cells = &tile[0][min(sub, TW - 1)];  // (TW = 15: lane 15 never has a window)
    const int roff = FM / 2 - F / 2 - c0;       // centred row of absolute cycle x: x + roff
    const int n_bins_l = task_on ? wb.b_hi - wb.b_lo + 1 : 0;
    for (int grp = blk0 >> ADH_SUB_SHIFT; grp <= (blk1 >> ADH_SUB_SHIFT); ++grp) {  // (wave-uniform only per group of lanes:
        // lanes of a candidate that has no such group idle, see `on`)
        const int sb0 = max(blk0 - (grp << ADH_SUB_SHIFT), 0), sb1 = min(blk1 - (grp << ADH_SUB_SHIFT), ADH_SUB - 1);
        const uint2 *ent = adh_group_entries(run, grp << ADH_SUB_SHIFT);  // (table words count from the group's first entry)
        // two bins of the window at a time: their table words, then their entries, are in flight together
        // (one dependent round trip per pair instead of per bin; a window rarely has a third bin)
        for (int j = 0; __any(j < n_bins_l); j += NPAIR) {
            Task t[NPAIR];
#pragma unroll
            for (int u = 0; u < NPAIR; ++u)
                task_begin(run, j + u < n_bins_l && sb1 >= sb0, task_row, grp, sb0, max(sb1, sb0), wb.b_lo + j + u, wb.b_lo, c0,
                           F, t[u]);
#pragma unroll
            for (int u = 0; u < NPAIR; ++u) task_fetch(ent, t[u]);
#pragma unroll
            for (int u = 0; u < NPAIR; ++u) task_run<TW>(ent, wb, t[u], cells, roff, hits);
        }
    }
}

}  // namespace fused

// one wavefront = four candidates of a class with FM registers; `block` counts the wavefronts of the class
template <int FM, int NO, int TW>
__device__ __forceinline__ void fused_body(const DevRun &run, const LibRec *__restrict__ lib, const CandRec *__restrict__ plan,
                                           int32_t n_cand, int32_t block, const float *__restrict__ iso_table,
                                           int32_t n_iso_cols, const adh_scoring_config_t &cfg,
                                           const double *__restrict__ wtp_table, const DevOut &out, int32_t stop_phase,
                                           unsigned char *smem) {
    using namespace fused;
    using feat::Assemble;
    constexpr int RC = FM / 2;
    constexpr int KMAX = ISO0;  // fragments a candidate of this kernel keeps at most
    GroupLds<FM, NO, TW> *lds = reinterpret_cast<GroupLds<FM, NO, TW> *>(smem);
    const int lane = threadIdx.x;
    const int g = lane / GS, sub = lane % GS;
    const unsigned gsh = (unsigned)(g * GS);
    GroupLds<FM, NO, TW> &L = lds[g];
    Keep<FM, NO> &KP = L.keep();
    const int ci = block * (ADH_WAVE / GS) + g;
    bool alive = ci < n_cand;
    const CandRec &rec = plan[alive ? ci : 0];
    alive = alive && !(rec.flags & ADH_FLAG_SKIP);
    const uint32_t row = rec.row;
    if (alive && sub == 0 && out.precursor_idx) {
        out.precursor_idx[row] = rec.precursor_idx;  // candidate.py:175-176
        out.rank[row] = rec.rank;
    }
    const int Lc = run.cycle_len;
    const int c0 = rec.frame_start / Lc;
    const int F = alive ? rec.frame_stop / Lc - c0 : 0;
    const int c = F / 2;
    const int shift = c - RC;  // f = r + shift
    const int I = alive ? min(min(n_iso_cols, (int)cfg.top_k_isotopes), TW - ISO0) : 0;  // (the host picks TW for I)
    const int top_k = out.top_k;
    if (stop_phase == 19) {  // developer ablation (ADH_DEBUG_STOP_PHASE): launch + candidate record only
        if (F == -12345) out.valid[row] = 2;
        return;
    }

    // ================= fragments: slice, cardinality filter, top-k by intensity, sort by m/z =========
    // (fragment_container.py:56-102; the tie rules of argsort()[::-1][:k] and of the stable argsort(mz))
    const int64_t frag_start = rec.frag_start;
    const int n_lib = alive ? (int)(rec.frag_stop - rec.frag_start) : 0;
    RawRec mine;
    mine.a = make_uint4(0u, 0u, 0u, 0u);
    mine.b = 0u;
    if (sub < n_lib) mine = load_rec(lib + frag_start + sub);
    const float mine_int = rec_intensity(mine), mine_mz = rec_mz(mine);
    // loads that do not depend on the selection: issued now, consumed after the gather
    const float rt_first = alive ? run.rt[rec.frame_start] : 0.0f;
    const float rt_last = alive ? run.rt[max(rec.frame_stop - 1, 0)] : 0.0f;
    float loc = 0.0f;  // location features (location_features.py:8-33)
    if (alive && (sub == 0 || sub == 2 || sub == 3)) {
        loc = sub == 0   ? run.mobility[rec.scan_start] - run.mobility[rec.scan_stop - 1]
              : sub == 2 ? run.rt[rec.frame_center]
                         : run.mobility[rec.scan_center];
    }
    float frt_l[(FM + 15) / 16];  // frame RTs of the registers sub, sub + 16
#pragma unroll
    for (int pass = 0; pass < (FM + 15) / 16; ++pass) {
        const int r = min(sub + 16 * pass, FM - 1);
        const int f = r + shift;
        const bool ok = alive && f >= 0 && f < F;
        frt_l[pass] = ok ? run.rt[rec.frame_start + f * Lc] : 0.0f;
    }
    const bool iso_lane = alive && sub >= ISO0 && sub - ISO0 < I;
    const int il = iso_lane ? sub - ISO0 : 0;
    float iso_int_l = 0.0f;
    if (iso_lane) iso_int_l = iso_table[(int64_t)row * n_iso_cols + il];
    // quadrupole transfer function (quadrupole.py:261-301, n_scans == 1): a logistic term per (isotope,
    // observation, window edge) - at most 4 x 2 x 2 = 16: lane e = i + 4 o + 4 NO edge evaluates one, so that the
    // candidate costs ONE float64 exp and division instead of two per observation on three lanes
    const int q_i = sub & 3, q_o = (sub >> 2) % NO, q_edge = sub / (4 * NO);
    const bool q_lane = alive && sub < 8 * NO && q_i < I;
    double q_mu = 0.0;
    if (q_lane) q_mu = run.cycle[2 * ((int64_t)rec.obs[q_o] * run.cycle_scans + rec.scan_start) + q_edge];
    const int ms1_row = run.ms1_obs[0];
    if (stop_phase == 20) {  // ... + library records and the other first-round loads
        if (mine_int + rt_first + rt_last + loc + frt_l[0] + iso_int_l + (float)q_mu + (float)ms1_row == -12345.5f) out.valid[row] = 2;
        return;
    }
    int K0 = 0;
    if (!__any(n_lib > GS)) {
        // ---- every fragment of the slice has its lane: ranks by comparison with the 15 other lanes of
        // the row, values moved by DPP rotations (no LDS round trips in the dependency chain)
        const bool ok_l = sub < n_lib && !(cfg.exclude_shared_ions && rec_cardinality(mine) > 1);
        // (a lane that is out travels as -inf: never above, never equal to a library intensity)
        const int ia = __float_as_int(ok_l ? mine_int : -INFINITY);
        int rk = 0;
#define FU_RANK_STEP(N)                                                                                \
    {                                                                                                  \
        const int b = row_ror<N>(sub);                                                                 \
        const float ib = __int_as_float(row_ror<N>(ia));                                               \
        rk += (int)((ib > mine_int) | ((ib == mine_int) & (b > sub))); /* (no short circuit: no branches) */ \
    }
        FU_FOR_OTHER_LANES(FU_RANK_STEP)  // position in argsort()[::-1]
#undef FU_RANK_STEP
        if (!ok_l || rk >= (int)cfg.top_k_fragments) rk = -1;
        const unsigned selm = (unsigned)((__ballot(rk >= 0) >> gsh) & 0xFFFFull);
        K0 = __popc(selm);
        if (K0 <= 3) {  // candidate.py:190,230
            K0 = 0;
            alive = false;
        }
        if (__ballot(alive) == 0ull) return;
        int slot = 0;
        // (a lane that is not kept travels as +inf: never below, never equal to a library m/z)
        const int ma = __float_as_int(rk >= 0 ? mine_mz : INFINITY);
#define FU_SLOT_STEP(N)                                                                           \
    {                                                                                             \
        const int rb = row_ror<N>(rk);                                                            \
        const float mb = __int_as_float(row_ror<N>(ma));                                          \
        slot += (int)((mb < mine_mz) | ((mb == mine_mz) & (rb < rk)));                            \
    }
        FU_FOR_OTHER_LANES(FU_SLOT_STEP)  // stable argsort(mz) of the top-k list
#undef FU_SLOT_STEP
        if (rk >= 0 && alive) {
            L.u.s.sel[slot].a = mine.a;
            L.u.s.sel[slot].b = mine.b | ((uint32_t)sub << 16);  // position inside the library slice (adh_output_t.fragment_lib_slot)
        }
    } else {
        // ---- longer slices (up to 64): the same counting through LDS
        for (int j = sub; j < n_lib; j += GS) {
            RawRec lr = mine;
            if (j != sub) lr = load_rec(lib + frag_start + j);
            L.u.s.l_int[j] = rec_intensity(lr);
            L.u.s.l_mz[j] = rec_mz(lr);
            L.u.s.l_ok[j] = !(cfg.exclude_shared_ions && rec_cardinality(lr) > 1);
        }
        adh_wave_sync();
        for (int a = sub; a < n_lib; a += GS) {
            int rk = -1;
            if (L.u.s.l_ok[a]) {
                rk = 0;
                const float ia = L.u.s.l_int[a];
                for (int b = 0; b < n_lib; ++b) {
                    const float ib = L.u.s.l_int[b];
                    rk += (L.u.s.l_ok[b] != 0) && ((ib > ia) || (ib == ia && b > a));  // position in argsort()[::-1]
                }
                if (rk >= (int)cfg.top_k_fragments) rk = -1;
            }
            L.u.s.l_rank[a] = rk;
        }
        adh_wave_sync();
        for (int a = 0; a < n_lib; ++a) K0 += L.u.s.l_rank[a] >= 0;
        if (K0 <= 3) {  // candidate.py:190,230
            K0 = 0;
            alive = false;
        }
        if (__ballot(alive) == 0ull) return;
        for (int a = sub; a < n_lib; a += GS) {
            const int ra = L.u.s.l_rank[a];
            if (ra < 0 || !alive) continue;
            const float ma = L.u.s.l_mz[a];
            int slot = 0;
            for (int b = 0; b < n_lib; ++b) {
                const int rb = L.u.s.l_rank[b];
                const float mb = L.u.s.l_mz[b];
                slot += (rb >= 0) && ((mb < ma) || (mb == ma && rb < ra));  // stable argsort(mz) of the top-k list
            }
            RawRec pick = mine;
            if (a != sub) pick = load_rec(lib + frag_start + a);
            L.u.s.sel[slot].a = pick.a;
            L.u.s.sel[slot].b = pick.b | ((uint32_t)a << 16);  // position inside the library slice (adh_output_t.fragment_lib_slot)
        }
    }
    adh_wave_sync();
    const bool frag_lane0 = alive && sub < K0;
    RawRec lrec;
    lrec.a = make_uint4(0u, 0u, 0u, 0u);
    lrec.b = 0u;
    if (frag_lane0) {
        lrec.a = L.u.s.sel[sub].a;
        lrec.b = L.u.s.sel[sub].b;
    }
    const float lrec_mz = rec_mz(lrec), lrec_int = rec_intensity(lrec);
    adh_wave_sync();  // the selection arrays are dead: the tile takes their place
    if (stop_phase == 21) {
        if (lrec_mz == -1.5f) out.valid[row] = 2;
        return;
    }

    // ================= windows (jitclasses/utils.py:15-20: float32 throughout) =================
    gather::Window w;
    w.lo = 0.0f;
    w.hi = -INFINITY;
    float iso_mz_l = 0.0f;
    if (frag_lane0) {
        const float ma = lrec_mz;
        const float t = cfg.fragment_mz_tolerance * ma;
        const float q = t / 1000000.0f;
        w.lo = ma - q;
        w.hi = ma + q;
    }
    if (iso_lane) {  // isotope m/z (candidate.py:151-163)
        const double off = (double)il * 1.0033548350700006 / (double)rec.charge;
        iso_mz_l = (float)off + rec.precursor_mz;
        const float t = cfg.precursor_mz_tolerance * iso_mz_l;
        const float q = t / 1000000.0f;
        w.lo = iso_mz_l - q;
        w.hi = iso_mz_l + q;
    }
    const bool win_lane = frag_lane0 || iso_lane;
    {
        // the reference's monotone cursor: window k starts above max(hi_j, j < k) (adh_gather.hip)
        const float ef = row_prefix_max_excl(frag_lane0 ? w.hi : -INFINITY);
        const float ei = row_prefix_max_excl(iso_lane ? w.hi : -INFINITY);
        w.excl = iso_lane ? ei : ef;
    }
    gather::bins_of(run, w);
    const bool task_on = win_lane && w.b_hi >= w.b_lo;
    const int task_row = iso_lane ? ms1_row : (int)rec.obs[0];
    const WinBits wb = win_bits(w);

    // ================= gather: every lane its window, into its column of the tile =================
    uint32_t hits = 0;
    gather_pass<FM, TW>(run, wb, task_on, task_row, alive, c0, F, L.u.tile, sub, hits);
    // quadrupole_transfer_function_single (quadrupole.py:261-301): logistic(x, lower edge) - logistic(x, upper edge)
    double q_term = 0.0;
    {
        // isotope m/z of this lane's term (candidate.py:151-163), as the isotope lane computes it
        const double off = (double)q_i * 1.0033548350700006 / (double)rec.charge;
        const float mzq = (float)off + rec.precursor_mz;
        // (a fitted calibration shifts the edges and has its own widths: SimpleQuadrupoleJit.predict, quadrupole.py:94-113)
        const QuadParams qp = adh_quad_params(cfg);
        if (q_lane) q_term = logistic((double)mzq, q_mu + (q_edge ? qp.delta_hi : qp.delta_lo), q_edge ? qp.sigma_hi : qp.sigma_lo);
    }
    double q_io;  // lanes i + 4 o: the transfer function of (isotope i, observation o)
    {
        const double upper = __shfl(q_term, (int)gsh + ((sub + 4 * NO) & 15));
        q_io = q_term - upper;
    }
    // the isotope lanes take theirs; qtf mask of the fragment tile (candidate.py:287-289): mean over the
    // isotopes, in order
    double qtf_l[NO];
    float qmask[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        const double mine_q = __shfl(q_io, (int)gsh + 4 * o + (il & 3));
        qtf_l[o] = iso_lane ? mine_q : 0.0;
        double qs = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double qi = __shfl(q_io, (int)gsh + 4 * o + i);
            if (i < I) qs += qi;
        }
        qmask[o] = (I > 0) ? (float)(qs / (double)I) : 0.0f;
    }
    if (stop_phase == 2) return;

    float A[FM], B[FM];
    float P[FM];  // frame profile summed over the observations (frame_profile_2d + sum over o)
    auto &Q = L.u.f;
    const float rt_width = rt_last - rt_first;
    if (sub == 1) loc = rt_width;
    float tsum[NO], rowsum_l[NO], ftc_l[NO], fw_l[NO];
    double ohe_l[NO], omz_l[NO];
    int fpeak_l[NO];
    float so = 0.0f;
    // quantification of an enveloped profile (fragment_features.py:252-273): trapezoid area x window and the
    // intensity inside the window
    double area = 0.0;
    float obs_int = 0.0f;
    const int qw = min(c - 1, (int)cfg.quant_window);
    auto quantify = [&](const float (&E)[FM]) {
        double ar = 0.0;
#pragma unroll
        for (int r = 1; r < FM - 1; ++r) {
            const bool in = r >= RC - qw && r + 1 <= RC + qw;
            const float sm = E[r + 1] + E[r];
            const float drt = KP.frt[r + 1] - KP.frt[r];
            const float m = sm * drt;
            const float mi = in ? m : 0.0f;  // (outside the window the term is +0, as if skipped)
            ar += (double)mi * 0.5;
        }
        area = ar * (double)qw;
        float oi_sum = 0.0f;
        FU_FOR_R oi_sum += (r >= RC - qw && r <= RC + qw) ? E[r] : 0.0f;
        obs_int = oi_sum;
    };
    // without quant_all the profile of the most important observation is quantified, and - a view in the
    // reference (fragment_features.py:240-250) - its envelope edit stays in fragments_frame_profile
    int best_obs = 0;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        if (o > 0) {
            // the next observation: only the fragment lanes gather (cycle row obs[o]); the tile takes the
            // place of the feature arrays, what has to survive sits in KP
            adh_wave_sync();
            gather_pass<FM, TW>(run, wb, task_on && !iso_lane, (int)rec.obs[o], alive, c0, F, L.u.tile, sub, hits);
        }
        // ================= rows into registers (the tile is dead afterwards) =================
        {
            if (o == 0) {
                // MS1 observation collapse (candidate.py:248-269) with ONE MS1 row per cycle: the sum over the
                // single observation is the value itself, the mean m/z is y / (count + 1e-6) with count = 1
                // where y > 0 (y is a weighted mean of m/z values: positive or an untouched 0 = 0 / 1e-6).
                // In place on the isotope columns of the tile, the I x FM cells dealt to all 16 lanes of the
                // group (three isotope lanes dividing FM cells each kept the other thirteen waiting).
                const Recip one(1.0 + 1e-6);
                constexpr int NISO = TW - ISO0;
                adh_wave_sync();  // the columns were written by their own lanes
#pragma unroll
                for (int pass = 0; pass < (NISO * FM + 15) / 16; ++pass) {
                    const int idx = sub + 16 * pass;
                    const int i = idx / FM, r = idx - i * FM;
                    if (i < I) {
                        float2 &cell = L.u.tile[r][ISO0 + i];
                        const float y = cell.y;
                        if (y > 0.0f) cell.y = (float)one.div((double)y);
                    }
                }
                adh_wave_sync();
            }
            const float2 *col = &L.u.tile[0][min(sub, TW - 1)];
            const float mq = iso_lane ? 1.0f : qmask[o];  // candidate.py:290 (fragments only)
            FU_FOR_R {
                const float2 v = col[r * TW];
                A[r] = v.x * mq;
                B[r] = v.y;
            }
        }
        adh_wave_sync();
        if (o == 0) {
#pragma unroll
            for (int pass = 0; pass < (FM + 15) / 16; ++pass) {
                const int r = min(sub + 16 * pass, FM - 1);  // (duplicates of the last row write the same value)
                KP.frt[r] = frt_l[pass];
            }
            // ---- isotope lanes: template rows of every observation (quadrupole.py:304-324)
            if (iso_lane) {
                KP.iso_mz[il] = iso_mz_l;
                KP.iso_int[il] = iso_int_l;
            }
#pragma unroll
            for (int oo = 0; oo < NO; ++oo) {
                if (iso_lane) {
                    FU_FOR_R {
                        const float a = A[r] * iso_int_l;
                        Q.u.dT[il][r] = (double)a * qtf_l[oo];
                    }
                }
                adh_wave_sync();
#pragma unroll
                for (int pass = 0; pass < (FM + 15) / 16; ++pass) {
                    const int r = sub + 16 * pass;
                    if (r < FM) {
                        double acc = 0;
                        for (int i = 0; i < I; ++i) acc += Q.u.dT[i][r];
                        if (oo == 0) Q.tpl[r] = (float)acc;  // zero outside [0, F)
                        else KP.tpl_next[r] = (float)acc;
                    }
                }
                adh_wave_sync();
            }
            // precursor weights exp(-0.1 * sqrt((s - 2)^2 + (f - 1)^2)) around the expected centre (S, 1) of
            // precursor_features.py:52-57, centred and masked like the fragments' table below
#pragma unroll
            for (int pass = 0; pass < (2 * FM + 15) / 16; ++pass) {
                const int idx = min(sub + 16 * pass, 2 * FM - 1);
                const int sc = idx / FM, r = idx - sc * FM;
                const int f = r + shift;
                const bool ok = alive && f >= 0 && f < F;
                Q.u.wti[sc][r] = ok ? wtp_table[sc * 64 + f] : 0.0;
            }
        } else {
#pragma unroll
            for (int pass = 0; pass < (FM + 15) / 16; ++pass) {
                const int r = min(sub + 16 * pass, FM - 1);
                Q.tpl[r] = KP.tpl_next[r];
            }
            adh_wave_sync();
        }
        if (stop_phase == 31) return;
        // observation importance (quadrupole.py:327-335): template sums - all of them in the first round, the
        // rows of the later observations wait in KP.tpl_next (NO <= 2)
        if (o == 0) {
            float st = 0.0f;
            FU_FOR_R {
                st += Q.tpl[r];
                FU_FENCE(r);
            }
            tsum[0] = st + st;
            if (NO > 1) {
                float st1 = 0.0f;
                FU_FOR_R {
                    st1 += KP.tpl_next[r];
                    FU_FENCE(r);
                }
                tsum[NO - 1] = st1 + st1;
                float tot = 0.0f;
#pragma unroll
                for (int oo = 0; oo < NO; ++oo) tot += tsum[oo];
                float best_v = (tot == 0.0f) ? 1.0f / (float)NO : tsum[0] / tot;
#pragma unroll
                for (int oo = 1; oo < NO; ++oo) {  // np.argmax: the first maximum
                    const float v = (tot == 0.0f) ? 1.0f / (float)NO : tsum[oo] / tot;
                    if (v > best_v) best_v = v, best_obs = oo;
                }
            }
        }
        // ---- template centre of mass (fragment_features.py:20-68; every lane computes it), template frame
        // profile, weights around the centre
        double esc, efc;
        {
            double isum = 0, ssum = 0, fsum = 0;
#pragma unroll 1  // a real loop: unrolling lets hipcc keep 2 x 32 converted values live
            for (int sc = 0; sc < 2; ++sc) {
                // the cycle coordinate f = r + shift as a running float64 (exact) that starts from a value the
                // optimiser cannot see through: left to itself it converts all FM coordinates ahead of the two-trip
                // loop and keeps them in 2 FM registers next to the rows
                double fd = (double)shift;
                __asm__ volatile("" : "+v"(fd));
                FU_FOR_R {
                    float v = Q.tpl[r];
                    v = __builtin_fmaxf(v, 0.0f);  // the reference skips v <= 0; adding +0 is the same
                    const double vd = (double)v;
                    isum += vd;
                    fsum += fd * vd;
                    fd += 1.0;
                    FU_FENCE(r);
                }
                // sum of scan * v: scan 0 adds +0 FM times, scan 1 adds the row from zero in the order isum took
                // during scan 0 - the same additions, the same bits
                if (sc == 0) ssum = isum;
            }
            esc = (isum > 0) ? ssum / isum : 0.0;
            efc = (isum > 0) ? fsum / isum : 0.0;
        }
        // template frame profile with or_envelope (scoring/utils.py:46-53)
#pragma unroll
        for (int pass = 0; pass < (FM + 15) / 16; ++pass) {
            const int r = min(sub + 16 * pass, FM - 1);  // (duplicates of the last row write the same value)
            const int f = r + shift;
            const bool ok = alive && f >= 0 && f < F;
            const float x = Q.tpl[r] + Q.tpl[r];
            float rr = x;
            if (ok && f >= 1 && f < F - 1) {
                const float xl = Q.tpl[r - 1] + Q.tpl[r - 1];
                const float xr = Q.tpl[r + 1] + Q.tpl[r + 1];
                if (x < xl || x < xr) {
                    const float sm = xl + xr;
                    rr = sm * 0.5f;  // (float)((double)sm / 2): exact either way
                }
            }
            Q.tfp[r] = ok ? rr : 0.0f;
        }
        // weight table around the template centre (features_utils.py:9-25), centred index
        // A centre at scan 0.5 exactly (the two scan slots hold the same row: whenever the float64 sums above are
        // exact) gives both slots the same weights, (-0.5)^2 = 0.5^2: one evaluation serves both.
        const bool twin = __all(esc == 0.5 || !alive);
        const int n_wt = twin ? FM : 2 * FM;
#pragma unroll 1  // (pass after pass: interleaved float64 exp() evaluations cost registers the rows need)
        for (int pass = 0; pass * 16 < n_wt; ++pass) {
            const int idx = min(sub + 16 * pass, n_wt - 1);
            const int sc = idx >= FM ? 1 : 0, r = idx - sc * FM;
            const int f = r + shift;
            const bool ok = alive && f >= 0 && f < F;
            double wv = 0.0;
            if (ok) {
                const double ds = (double)sc - esc, df = (double)f - efc;
                wv = exp(-0.1 * sqrt(ds * ds + df * df));
            }
            Q.wt[sc][r] = wv;
            if (twin) Q.wt[1][r] = wv;
        }
        adh_wave_sync();
        if (stop_phase == 32) return;

        // ================= row sums and weighted centre means: fragments and isotopes together =========
        // presence (candidate.py:319-329) / sum_precursor_intensity: row sum over the two identical scan slots
        float sf = 0.0f;
        FU_FOR_R sf += A[r];
        const float ss = sf + sf;
        // weighted centre means of both channels (features_utils.py:9-37; precursor_features.py:52-66).
        // A skipped cell (value <= 0) adds nothing: its product is +0, only its weight has to stay out.
        double m_int, m_mz;
        {
            const double *wrow = (o == 0 && iso_lane) ? &Q.u.wti[0][0] : &Q.wt[0][0];
            double vo = 0, wo = 0, vm = 0, wm = 0;
#pragma unroll 1  // a real loop: unrolling lets hipcc keep 2 x 32 converted values live
            for (int sc = 0; sc < 2; ++sc) {
                FU_FOR_R {
                    const double wv = wrow[sc * FM + r];
                    // (opaque copies, not the rows themselves: marking A[r] / B[r] saves the two moves but costs
                    // 44 bytes of scratch per lane - 25 more spill instructions and 1.2 GB of write traffic per step)
                    float a = A[r], b = B[r];
                    FU_OPAQUE(a);
                    FU_OPAQUE(b);
                    // the weight of a cell that holds something: wv x 1.0 + wo rounds once, like wo + wv; wv x 0.0
                    // adds +0 (one select on the high word of the indicator instead of two on the weight)
                    vo += (double)a * wv;
                    wo = __builtin_fma(wv, __hiloint2double(a > 0.0f ? 0x3ff00000 : 0, 0), wo);
                    vm += (double)b * wv;
                    wm = __builtin_fma(wv, __hiloint2double(b > 0.0f ? 0x3ff00000 : 0, 0), wm);
                    FU_FENCE(r);
                }
            }
            m_int = (wo > 0) ? vo / wo : 0.0;
            m_mz = (wm > 0) ? vm / wm : 0.0;
        }
        if (o == 0 && iso_lane) {
            KP.spi[il] = ss;
            KP.hp[il] = m_int;
            KP.omzp[il] = m_mz;
        }
        ohe_l[o] = frag_lane0 ? m_int : 0.0;
        omz_l[o] = frag_lane0 ? m_mz : 0.0;
        rowsum_l[o] = frag_lane0 ? ss : 0.0f;
        so += rowsum_l[o];
        if (NO > 1) {
            // per-observation frame profile (frame_profile_2d): statistics against this observation's
            // template.  With one observation they are taken after the envelope step, whose in-place edit
            // they must see when quant_all is off.
            FU_FOR_R A[r] = frag_lane0 ? A[r] + A[r] : 0.0f;
            if (!cfg.quant_all && o == best_obs) {  // (the branch is per candidate: a group's lanes agree)
                center_envelope<FM>(A, F);
                quantify(A);
            }
            profile_stats<FM>(A, Q.tfp, F, shift, rt_width, ftc_l[o], fw_l[o], fpeak_l[o]);
            if (o == 0) {
                FU_FOR_R P[r] = A[r];  // (0 + x for x >= +0)
            } else {
                FU_FOR_R P[r] += A[r];
            }
        } else {
            FU_FOR_R P[r] = frag_lane0 ? A[r] + A[r] : 0.0f;
        }
    }
    // (nothing of Q written above is read below: from here on the feature arrays are rebuilt)
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) hits += __shfl_xor(hits, m, GS);
    if (alive && sub == 0 && out.stat_matched_peaks) out.stat_matched_peaks[row] = hits;
    adh_wave_sync();
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (sub + 16 * j < ADH_NUM_FEATURES + 2) Q.feat[sub + 16 * j] = 0.0f;
    // observation importance (quadrupole.py:327-335)
    float oi[NO];
    {
        float tot = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) tot += tsum[o];
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            oi[o] = (tot == 0.0f) ? 1.0f / (float)NO : tsum[o] / tot;
            if (sub == 0) Q.oi[o] = oi[o];
        }
    }
    if (stop_phase == 33) return;
    bool present = frag_lane0 && so > 0.0f;
    const unsigned long long bal = __ballot(present);
    const unsigned gm = (unsigned)((bal >> gsh) & 0xFFFFull);
    int K = __popc(gm);
    const int kk = __popc(gm & ((1u << sub) - 1u));
    const int n_present = K;
    if (K < 2) {  // candidate.py:323
        alive = false;
        present = false;
        K = 0;
    }
    if (present) Q.g_fin[kk] = lrec_int;  // raw intensity, normalised below
    adh_wave_sync();
    // fragment intensities: apply_mask renormalisation + the second one of fragment_features.py:218
    float g_int_l = 0.0f, g_fin_l = 0.0f;
    {
        float sum1 = 0.0f;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) sum1 += (j < K) ? Q.g_fin[j] : 0.0f;
        if (present) {
            g_int_l = lrec_int / sum1;
            Q.g_int[kk] = g_int_l;
        }
    }
    adh_wave_sync();
    {
        float sum2 = 0.0f;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) sum2 += (j < K) ? Q.g_int[j] : 0.0f;
        if (present) g_fin_l = g_int_l / sum2;
    }
    adh_wave_sync();
    if (present) Q.g_fin[kk] = g_fin_l;
    if (stop_phase == 3 || stop_phase == 4) return;

    // ---- envelope, quantification (fragment_features.py:240-273); several observations without quant_all
    // were quantified inside the loop above
    if (NO == 1 || cfg.quant_all) {
        float E[FM];  // np.sum(axis=1) made a copy: with quant_all the profile itself is untouched
        FU_FOR_R E[r] = P[r];
        center_envelope<FM>(E, F);
        quantify(E);
        if (NO == 1 && !cfg.quant_all) {
            FU_FOR_R P[r] = E[r];  // a VIEW of the best observation's profile: edited in place
        }
    }
    double m1 = 0.0, m2 = 0.0, merr_l = 0.0;
    bool hrow = false;
    if (present) {
        // importance-weighted means over observations (fragment_features.py:311-336)
        if (NO == 1 && oi[0] == 1.0f) {
            // one observation of importance 1: w = 1 / (1 + 1e-20) = 1, local weight 1 / 1: the means are the
            // observation's own values (x * 1.0 and 0.0 + x are exact)
            hrow = ohe_l[0] > 0;
            if (hrow) {
                m1 = omz_l[0];
                m2 = ohe_l[0];
            }
        } else {
            float ws = 0.0f;
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const bool m = ohe_l[o] > 0;
                hrow = hrow || m;
                const float w32 = m ? oi[o] : oi[o] * 0.0f;
                ws += w32;
            }
            double msum = 0.0;
            int nm = 0;
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const bool m = ohe_l[o] > 0;
                const float w32 = m ? oi[o] : oi[o] * 0.0f;
                const double wd = (double)w32 / ((double)ws + 1e-20);
                if (wd > 0) {
                    msum += wd;
                    ++nm;
                }
            }
            if (nm > 0) {
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    const bool m = ohe_l[o] > 0;
                    const float w32 = m ? oi[o] : oi[o] * 0.0f;
                    const double wd = (double)w32 / ((double)ws + 1e-20);
                    if (wd > 0) {
                        const double lw = wd / msum;
                        m1 += omz_l[o] * lw;
                        m2 += ohe_l[o] * lw;
                    }
                }
            }
        }
        merr_l = (m1 - (double)lrec_mz) / (double)lrec_mz * 1e6;  // fragment_features.py:387
        Q.merr[kk] = merr_l;
        int rk = 0;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {  // (all reads first: one LDS round trip, not K)
            const float ib = Q.g_int[j];
            rk += (int)((j < K) & ((ib > g_int_l) | ((ib == g_int_l) & (j > kk))));
        }
        Q.ord[rk] = kk;  // position in argsort(intensity)[::-1]
    }
    adh_wave_sync();
    if (stop_phase == 5) return;

    if (alive && sub < 4) Q.feat[sub] = loc;
    // ---- fragment features 17-27, 41-45 (fragment_features.py:198-427; the scalar form is
    // feat::assemble_fragments) and precursor features 4-16, 28 (precursor_features.py:13-102).  Every sum
    // over fragments keeps the reference's order (k ascending) but all sums advance together: lane k provides
    // its term of every sum, then lane j adds up sum j (skipped terms are added as +0, which leaves a sum
    // unchanged) and finishes the features that hang on it: one float64 division and one log for the whole
    // candidate, each lane dividing its own operands.
    const bool ipos = present && obs_int > 0.0f;
    const bool hpos = present && m2 > 0.0;
    const bool isb = present && rec_type(lrec) == 98, isy = present && rec_type(lrec) == 121;
    const unsigned b_isb = (unsigned)((__ballot(isb) >> gsh) & 0xFFFFull);
    const unsigned b_isy = (unsigned)((__ballot(isy) >> gsh) & 0xFFFFull);
    const int n_int = __popc((unsigned)((__ballot(ipos) >> gsh) & 0xFFFFull));
    const int n_hei = __popc((unsigned)((__ballot(hpos) >> gsh) & 0xFFFFull));
    const int n_hrows = __popc((unsigned)((__ballot(present && hrow) >> gsh) & 0xFFFFull));
    const int nb = __popc(b_isb), ny = __popc(b_isy);
    const int lpos = rec_position(lrec);
    int min_y = isy ? lpos : 255, max_b = isb ? lpos : 0;
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) {
        min_y = min(min_y, __shfl_xor(min_y, m, GS));
        max_b = max(max_b, __shfl_xor(max_b, m, GS));
    }
    const bool ov = (isy && lpos < max_b) || (isb && lpos > min_y);
    const int n_ov = __popc((unsigned)((__ballot(ov) >> gsh) & 0xFFFFull));
    const int n3 = min(K, 3);
    if (present) {
        double *t = Q.u.at.t64[kk];
        t[0] = area;
        t[1] = m2;
        t[2] = (double)g_fin_l;
        t[3] = merr_l;
        t[4] = ov ? area : 0.0;
        t[5] = ov ? merr_l : 0.0;
        // cosine_similarity_a1 (features_utils.py:40-47) of the observation sums
        float tn = 0.0f, fn = 0.0f, dot = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) tn += tsum[o] * tsum[o];
        tn = sqrtf(tn);
#pragma unroll
        for (int o = 0; o < NO; ++o) fn += rowsum_l[o] * rowsum_l[o];
        fn = sqrtf(fn);
#pragma unroll
        for (int o = 0; o < NO; ++o) dot += rowsum_l[o] * tsum[o];
        const float pr = fn * tn;
        const float score = (float)((double)dot / ((double)pr + 0.0001));
        float *u = Q.u.at.t32[kk];
        u[0] = ipos ? g_fin_l : 0.0f;
        u[1] = hpos ? g_fin_l : 0.0f;
        u[2] = ipos ? score : 0.0f;
        u[3] = isb ? obs_int : 0.0f;
        u[4] = isy ? obs_int : 0.0f;
    }
    adh_wave_sync();
    {
        // lane j < 6: float64 sum j; 6 <= j < 11: float32 sum j - 6; lane 11: mean_top3 mass error, by rank
        double s64 = 0.0;
        float s32 = 0.0f;
        {
            double v64[KMAX];
            float v32[KMAX];
            const int c64 = min(sub, 5), c32 = min(max(sub - 6, 0), 5);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                v64[k] = Q.u.at.t64[k][c64];
                v32[k] = Q.u.at.t32[k][c32];
            }
            if (sub == 11) {
#pragma unroll
                for (int i = 0; i < 3; ++i) v64[i] = Q.merr[i < n3 ? Q.ord[i] : 0];
            }
            const int n64 = sub == 11 ? n3 : K;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                s64 += (k < n64) ? v64[k] : 0.0;
                s32 += (k < K) ? v32[k] : 0.0f;
            }
        }
        // ---- the quotients: lane -> (numerator, denominator)
        //   0-2 sums of area / height / intensity over K (np.corrcoef means), 3 mass error / K (42),
        //   4, 5 overlap area / mass error over n_ov (44, 45), 6 n_int / K (20), 7 n_hei / K (21),
        //   8 cosine sum / n_int (24), 9 n_present / K0 (28, candidate.py:362), 11 top-3 mass error / n3 (41),
        //   12 + i the mass error of isotope i (precursor_features.py:40-50)
        double num = s64, den = (double)K;
        if (sub == 4 || sub == 5) den = (double)n_ov;
        if (sub == 6) num = (double)n_int;
        if (sub == 7) num = (double)n_hei;
        if (sub == 8) num = (double)s32, den = (double)n_int;
        if (sub == 11) den = (double)n3;
        double omzp_i = 0.0;
        if (sub >= ISO0) {
            const int i = sub - ISO0;
            omzp_i = KP.omzp[i];
            num = omzp_i - (double)KP.iso_mz[i];
            den = (double)KP.iso_mz[i];
        }
        if (sub == 9) num = (double)n_present, den = (double)K0;
        const double quo = num / den;
        if (sub < 3) Q.red64[sub] = quo;
        if (sub == 1) Q.red64[3] = s64;  // (the sum of the heights decides whether feature 19 is taken)
        if (sub >= ISO0) {
            const double me = quo * 1e6;
            Q.red64[4 + sub - ISO0] = (omzp_i > 0) ? me * (double)KP.iso_int[sub - ISO0] : 0.0;
        }
        if (ADH_FUSED_SCALAR && alive) {
            float *ft = Q.feat;
            if (sub == 3) ft[42] = (float)quo;
            if ((sub == 4 || sub == 5) && nb > 0 && ny > 0) {
                if (sub == 4) ft[43] = (float)n_ov;
                ft[40 + sub] = n_ov > 0 ? (float)quo : (sub == 4 ? 0.0f : 15.0f);
            }
            if (sub == 6) ft[22] = s32, ft[20] = (float)quo;
            if (sub == 7) ft[23] = s32, ft[21] = (float)quo;
            if (sub == 8 && n_int > 0) ft[24] = (float)quo;
            if (sub == 9 || sub == 10) {
                const float lg = (float)log((double)s32 + 1.0);
                ft[16 + sub] = ((sub == 9 ? nb : ny) > 0) ? lg : 0.0f;
            }
            if (sub == 11) ft[41] = (float)quo;
            if (sub == 9) ft[28] = (float)quo, ft[17] = (float)NO;
        }
    }
    adh_wave_sync();
    {
        // np.corrcoef terms (feat::corrcoef01): area vs intensity, height vs intensity
        const double mx_a = Q.red64[0], mx_h = Q.red64[1], my = Q.red64[2];
        if (present) {
            const double a = area - mx_a, h = m2 - mx_h, b = (double)g_fin_l - my;
            double *t = Q.u.at.t64[kk];
            t[0] = a * a;
            t[1] = b * b;
            t[2] = a * b;
            t[3] = h * h;
            t[4] = h * b;
        }
    }
    adh_wave_sync();
    {
        double v64[KMAX], s64 = 0.0;
        const int c64 = min(sub, 4);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) v64[k] = Q.u.at.t64[k][c64];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) s64 += (k < K) ? v64[k] : 0.0;
        if (sub < 5) Q.red64[8 + sub] = s64;
    }
    adh_wave_sync();
    if (ADH_FUSED_SCALAR && alive && sub < 2) {
        // lane 0: feature 18 (areas), lane 1: feature 19 (heights)
        const double fact = fmax((double)K - 1.0, 0.0);
        const double inv = 1.0 / fact;
        const double cxx = Q.red64[sub ? 11 : 8] * inv, cyy = Q.red64[9] * inv;
        const double cxy = Q.red64[sub ? 12 : 10] * inv;
        const double s0 = sqrt(cxx), s1 = sqrt(cyy);
        double cc = cxy / s1 / s0;
        if (fabs(cc) > 1.0) cc = (cc > 0) ? 1.0 : -1.0;
        const bool on = sub ? (Q.red64[3] > 0.0) : (n_hrows > 0);
        if (on) Q.feat[18 + sub] = (float)cc;
    }
    if (ADH_FUSED_SCALAR && alive && sub == 2) Q.feat[27] = Q.feat[25] - Q.feat[26];
    if (ADH_FUSED_SCALAR && alive && sub == 3)
        precursor_features<NO>(Q.feat, I, KP.iso_int, KP.iso_mz, KP.spi, KP.hp, &Q.red64[4], oi);
    if (stop_phase == 6) return;

    // ================= profile features (profile_features.py:18-206), experimental_xic =======
    {
        // normalize_profiles (scoring_utils.py:71-117): centre +- 1 are registers RC-1, RC, RC+1
        float sm = 0.0f;
        sm += P[RC - 1];
        sm += P[RC];
        sm += P[RC + 1];
        const double cn = (double)sm / 3.0;
        const bool cpos = cn > 0;
        const Recip rcn(cpos ? cn : 1.0);
        // median over fragments per cycle (scoring_utils.py:120-152): 16x16 transposes via LDS
#pragma unroll
        for (int half = 0; half < (FM + 15) / 16; ++half) {
            adh_wave_sync();  // previous users of the union are done
            if (present) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    if (half * 16 + t >= FM) break;
                    const float x = P[half * 16 + t];
                    Q.u.nrmT[t][kk] = cpos ? (float)rcn.div((double)x) : 0.0f;
                }
            }
            adh_wave_sync();
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = (j < K) ? Q.u.nrmT[sub][j] : INFINITY;
            static_assert(KMAX <= 12, "the median network sorts twelve inputs");
            fast::sort_first12(v);  // (KMAX = 12: v[12..15] are +inf and stay where they are)
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
                const float s2 = lo_v + hi_v;
                m = (float)((double)s2 / 2.0);
            }
            const int r = half * 16 + sub;
            const int f = r + shift;
            if (r < FM) Q.med[r] = (alive && f >= 0 && f < F) ? m : 0.0f;
        }
    }
    adh_wave_sync();
    if (stop_phase == 61) return;
    float corr_l = 0.0f;
    {
        // correlation_coefficient (scoring_utils.py:14-68)
        float sx = 0.0f;
        FU_FOR_R {
            sx += Q.med[r];
            FU_FENCE(r);
        }
        const float mx = (float)((double)sx / (double)F);
        float sxx = 0.0f, sy = 0.0f;
        FU_FOR_R {
            float xm = Q.med[r] - mx;
            xm = FU_OK(r) ? xm : 0.0f;
            sxx += xm * xm;
            FU_FENCE(r);
        }
        const double var_x = (double)sxx / (double)F;
        FU_FOR_R sy += P[r];
        const float my = (float)((double)sy / (double)F);
        float sxy = 0.0f, syy = 0.0f;
        FU_FOR_R {
            float xm = Q.med[r] - mx;
            const float ym = P[r] - my;
            xm = FU_OK(r) ? xm : 0.0f;  // (one zero factor is enough)
            sxy += xm * ym;
            FU_FENCE(r);
        }
        FU_FOR_R {
            float ym = P[r] - my;
            ym = FU_OK(r) ? ym : 0.0f;
            syy += ym * ym;
        }
        const double cov = (double)sxy / (double)F;
        const double var_y = (double)syy / (double)F;
        const double var_xy = var_x * var_y;
        corr_l = (var_xy == 0) ? 0.0f : (float)(cov / sqrt(var_xy));
    }
    if (NO == 1) profile_stats<FM>(P, Q.tfp, F, shift, rt_width, ftc_l[0], fw_l[0], fpeak_l[0]);
#pragma unroll
    for (int o = 0; o < NO; ++o) Q.fpeak[sub][o] = present ? fpeak_l[o] : INT_MAX;  // (an absent lane ranks behind every apex)
    if (stop_phase == 62) return;
    if (present) Q.corr[kk] = corr_l;
    adh_wave_sync();
    // ---- features 31-38, 40 (profile_features.py:70-113,141-146,196-204; the scalar form is
    // feat::assemble_part2), sums organised as above
    {
        const int r_lo = (K - 1) / 2, r_hi = K / 2;
        if (present) {
            // median apex per observation (profile_features.py:196-198): rank of this fragment's apex
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const int va = fpeak_l[o];
                int rk = 0;
#pragma unroll
                for (int b = 0; b < KMAX; ++b) {  // (fragment lanes only; absent ones hold INT_MAX)
                    const int vb = Q.fpeak[b][o];
                    rk += (int)((vb < va) | ((vb == va) & (b < sub)));
                }
                if (rk == r_lo) Q.medlo[o] = va;
                if (rk == r_hi) Q.medhi[o] = va;
            }
            const float cr = Q.corr[Q.ord[kk]];  // correlation of the fragment with intensity rank kk
            // b / y: mask in original order applied to the sorted index array (profile_features.py:94-113)
            const bool b3 = isb && __popc(b_isb & ((1u << sub) - 1u)) < 3;
            const bool y3 = isy && __popc(b_isy & ((1u << sub) - 1u)) < 3;
            float rr = 0.0f, ml = 0.0f;
#pragma unroll
            for (int o = 0; o < NO; ++o) rr += ftc_l[o] * oi[o];
#pragma unroll
            for (int o = 0; o < NO; ++o) ml += fw_l[o] * oi[o];
            float *u = Q.u.at.t32[kk];
            u[0] = corr_l;
            u[1] = rr * g_int_l;
            u[2] = ml * g_int_l;
            u[3] = b3 ? cr : 0.0f;
            u[4] = y3 ? cr : 0.0f;
            u[5] = (kk < n3) ? cr : 0.0f;
        }
    }
    adh_wave_sync();
    {
        // lane j < 6 adds up sum j and finishes its feature (one division for all of them); lane 6: feature 40
        float v32[KMAX], s32 = 0.0f;
        const int c32 = min(sub, 5);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) v32[k] = Q.u.at.t32[k][c32];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) s32 += (k < K) ? v32[k] : 0.0f;
        const int dn = sub == 0 ? K : (sub == 5 ? n3 : (sub == 3 ? min(nb, 3) : min(ny, 3)));
        const double quo = (double)s32 / (double)dn;
        if (ADH_FUSED_SCALAR && alive) {
            float *ft = Q.feat;
            if (sub == 0) ft[31] = (float)quo;
            if (sub == 5) ft[32] = (float)quo;
            if (sub == 1) ft[33] = s32;
            if (sub == 2) ft[38] = s32;
            if (sub == 3 && nb > 0) ft[34] = (float)quo, ft[35] = (float)nb;
            if (sub == 4 && ny > 0) ft[36] = (float)quo, ft[37] = (float)ny;
            if (sub == 6) {
                double acc = 0.0;
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    const double med = (K & 1) ? (double)Q.medhi[o] : (double)(Q.medlo[o] + Q.medhi[o]) / 2.0;
                    const float medpk = (float)med;
                    acc += ((double)medpk - floor((double)F / 2.0)) * (double)oi[o];
                }
                ft[40] = (float)acc;
            }
        }
    }
    adh_wave_sync();

    // ---- output row (candidate.py:403-481)
    if (alive) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = sub + 16 * j;
            if (idx < ADH_NUM_FEATURES) out.features[(int64_t)row * ADH_NUM_FEATURES + idx] = Q.feat[idx];
        }
        if (cfg.collect_fragments && present && kk < top_k) {
            const int64_t o = (int64_t)row * top_k + kk;
            if (out.fragment_precursor_idx) {  // (NULL: the columns that repeat ids / the library are rebuilt later)
                out.fragment_precursor_idx[o] = rec.precursor_idx;
                out.fragment_rank[o] = rec.rank;
                out.fragment_mz_library[o] = __uint_as_float(lrec.a.x);
                out.fragment_mz[o] = lrec_mz;
                out.fragment_position[o] = (uint8_t)lpos;
                out.fragment_number[o] = (uint8_t)(lrec.a.w >> 24);
                out.fragment_type[o] = (uint8_t)(lrec.a.w & 0xFFu);
                out.fragment_charge[o] = (uint8_t)((lrec.a.w >> 16) & 0xFFu);
                out.fragment_loss_type[o] = (uint8_t)((lrec.a.w >> 8) & 0xFFu);
            }
            out.fragment_mz_observed[o] = (float)m1;
            out.fragment_height[o] = (float)m2;
            out.fragment_intensity[o] = (float)area;
            out.fragment_mass_error[o] = (float)merr_l;
            out.fragment_correlation[o] = corr_l;
            if (out.fragment_lib_slot) out.fragment_lib_slot[o] = (uint16_t)(1u + (lrec.b >> 16));
        }
        if (sub == 0) out.valid[row] = 1;
    }
}

// The classes of one batch in ONE launch: a launch drains the GPU at its end (a wavefront lives ~70 us) and
// costs ~10 us to start, and a chunk of the host -> host pipeline has up to fourteen classes of them.  Wavefronts are
// ordered by class (the plan is), so a compute unit runs one class at a time except at the seams.
#define ADH_FUSED_MAX_CLASSES 7
struct FusedClasses {
    int32_t n;                                       // classes in this launch
    int32_t first_block[ADH_FUSED_MAX_CLASSES + 1];  // first wavefront of class i; [n] = all
    int32_t first_cand[ADH_FUSED_MAX_CLASSES];       // first candidate of the class in the plan
    int32_t n_cand[ADH_FUSED_MAX_CLASSES];
    int32_t kind[ADH_FUSED_MAX_CLASSES];             // (FM - 8) / 4 + 7 * (observations - 1)
};
template <int TW>
constexpr size_t adh_fused_lds_bytes(int fm_max, int no) {
    return (fm_max > 28 ? (no > 1 ? sizeof(fused::GroupLds<32, 2, TW>) : sizeof(fused::GroupLds<32, 1, TW>))
                        : (no > 1 ? sizeof(fused::GroupLds<28, 2, TW>) : sizeof(fused::GroupLds<28, 1, TW>))) *
           (ADH_WAVE / 16);
}

// One launch per observation count: with the bodies of both in one kernel every wavefront pays the larger LDS
// block and spill area of the two-observation code (measured: 3 ms of the 13.5 ms of the one-observation
// candidates of the bench).  And one more each for the longest rows - FM = 32 needs more LDS, and with one
// observation the bodies for 28 and 32 cycles are the only ones that do not fit the registers of three
// wavefronts per SIMD (120 / 252 bytes of scratch per lane, 2.5 GB of spill stores per 3 M candidates):
// they run at two, without a spill, in a launch of their own.
// The arguments travel as ONE struct and the kernel reads them through the kernel-argument segment pointer where it
// needs them: as formal parameters all ~150 dwords are loaded in the prologue, do not fit the scalar registers beside
// the rest, get parked in the lanes of three vector registers and come back one v_readlane (a vector-ALU slot in a
// kernel that is bound by them) per dword and use - ~300 per wavefront.  Scalar loads from the (cached, read-only)
// segment at the point of use cost the vector ALUs nothing.
struct FusedArgs {
    DevRun run;
    const LibRec *lib;
    const CandRec *plan;
    FusedClasses fc;
    const float *iso_table;
    int32_t n_iso_cols;
    adh_scoring_config_t cfg;
    const double *wtp_table;
    DevOut out;
    int32_t stop_phase;
};
template <int FM_MIN, int FM_MAX, int NO, int TW>
__global__ __launch_bounds__(ADH_WAVE, (NO == 1 && FM_MAX <= ADH_FUSED_FM3) ? ADH_FUSED_WAVES : ADH_FUSED_WAVES2) void adh_fused_kernel(
    FusedArgs formal_args_not_read) {
    const FusedArgs &A = *(const FusedArgs *)__builtin_amdgcn_kernarg_segment_ptr();  // (the only argument: offset 0)
    const DevRun &run = A.run;
    const LibRec *__restrict__ lib = A.lib;
    const CandRec *__restrict__ plan = A.plan;
    const FusedClasses &fc = A.fc;
    const float *__restrict__ iso_table = A.iso_table;
    const int32_t n_iso_cols = A.n_iso_cols;
    const adh_scoring_config_t &cfg = A.cfg;
    const double *__restrict__ wtp_table = A.wtp_table;
    const DevOut &out = A.out;
    const int32_t stop_phase = A.stop_phase;
    __shared__ __align__(16) unsigned char smem[adh_fused_lds_bytes<TW>(FM_MAX, NO)];
    const int32_t b = (int32_t)blockIdx.x;
    int c = 0;
    while (c + 1 < fc.n && b >= fc.first_block[c + 1]) ++c;
    const CandRec *recs = plan + fc.first_cand[c];
    const int32_t n = fc.n_cand[c], blk = b - fc.first_block[c];
    const int kind = fc.kind[c] % 7;
#define ADH_FUSED_CASE(KIND, FM)                                                                                       \
    if constexpr (FM >= FM_MIN && FM <= FM_MAX) {                                                                      \
        if (kind == KIND) {                                                                                            \
            fused_body<FM, NO, TW>(run, lib, recs, n, blk, iso_table, n_iso_cols, cfg, wtp_table, out, stop_phase, smem); \
            return;                                                                                                    \
        }                                                                                                              \
    }
    ADH_FUSED_CASE(0, 8)
    ADH_FUSED_CASE(1, 12)
    ADH_FUSED_CASE(2, 16)
    ADH_FUSED_CASE(3, 20)
    ADH_FUSED_CASE(4, 24)
    ADH_FUSED_CASE(5, 28)
    ADH_FUSED_CASE(6, 32)
#undef ADH_FUSED_CASE
}
