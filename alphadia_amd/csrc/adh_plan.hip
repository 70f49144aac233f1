// adh_plan.hip - the processing plan of a candidate batch, built ON THE DEVICE.
//
// Replaces, together with the host function assemble_candidates (alphadia_amd/scoring.py),
//   assemble_score_group_container   alphadia/search/scoring/scoring.py:273-353
//   ScoreGroupContainer.build_from_df alphadia/search/scoring/containers/score_group.py:145-229
//   the quadrupole overlap test of get_dense  alphadia/search/jitclasses/alpharaw_jit.py:19-50
//                                             alphadia/search/jitclasses/bruker_jit.py:315-350
// The reference creates one jitclass object per candidate on the host.  Here the candidate
// columns are uploaded as they are (struct of arrays, the dtypes of the reference) and one thread
// per candidate
//   * checks every bound the scoring kernels rely on (an offending row sets an error code; the
//     host refuses the batch before any scoring kernel runs),
//   * finds the cycle rows whose isolation window overlaps the precursor ("observations"),
//   * picks the kernel class (register kernels by cycle count / observation count, or the
//     generic LDS kernel) and the size of the candidate's scratch block,
//   * emits a sort key (class, first cycle).
// A stable radix sort of (key, row) pairs, an exclusive scan of the scratch sizes in sorted order
// and one scatter produce the CandRec table in processing order; class boundaries, LDS capacities
// and the scratch size come back to the host in one 128-byte record.  Nothing of this depends on
// the order of candidates in the input, and the output rows are the input rows.
#pragma once
#include "adh_device.h"

// kernel classes, in processing order:
//   0 ..  6  fused gather + feature kernel (adh_fused.hip), one observation, FM = 8, 12, ..., 32 registers
//   7 .. 13  the same for two observations
//  14 .. 16  register kernels for two observations (FM = 16, 24, 32) behind the gather kernel: shapes the
//            fused kernel does not take (more than 12 fragments kept, library slices beyond 64)
//  17 .. 23  register kernels for one observation (FM = 8 ... 32) behind the gather kernel, likewise
//  24 .. 26  the WIDE register kernels (one candidate per wavefront, 64 fragment lanes; FM = 16, 24, 32) for two
//            observations: candidates of the register shape that keep 33 ... 64 fragments (transfer-library
//            requantification, top_k_fragments = 9999)
//  27 .. 29  the same for one observation
//  30 .. 35  the same pair with two candidates per wavefront, 32 lanes each: 17 ... 32 fragments kept
//  36        the generic LDS kernel behind the gather kernel
// ion-mobility plans use classes 0 (one observation), 1 (two), ADH_CLASS_IM_SMALL (one observation, a tile
// within the limits below: a feature-kernel instantiation with 12.8 KB of LDS instead of 16.3) and the generic one
#define ADH_CLASS_IM_SMALL 2
#define ADH_IM_SMALL_K 12
#define ADH_IM_SMALL_S 32
#define ADH_IM_SMALL_F 24
#define ADH_IM_SMALL_SF 640
#define ADH_CLASS_FUSED0 0
#define ADH_CLASS_FUSED2 7
#define ADH_CLASS_FAST2 14
#define ADH_CLASS_FAST1 17
#define ADH_CLASS_WIDE2 24
#define ADH_CLASS_WIDE1 27
#define ADH_CLASS_MID2 30
#define ADH_CLASS_MID1 33
#define ADH_CLASS_GENERIC 36
#define ADH_N_CLASSES 37
#define ADH_PLAN_WIDE_KMAX 64

// mirrors of the register-kernel limits (adh_features_fast.hip)
#define ADH_PLAN_FMAX 32
#define ADH_PLAN_FAST_OMAX 2

// candidate columns in HBM (adh_candidates_t, device copies)
struct DevCands {
    const uint32_t *precursor_idx, *frag_start, *frag_stop;
    const uint8_t *rank, *flags, *charge;  // flags may be NULL
    const int64_t *scan_start, *scan_stop, *scan_center, *frame_start, *frame_stop, *frame_center;
    const float *precursor_mz;
};

// error codes of the validation (first row that shows the highest code is reported)
enum {
    ADH_PLAN_OK = 0,
    ADH_PLAN_ERR_FRAG_SLICE = 1,
    ADH_PLAN_ERR_FRAME_LIMITS = 2,
    ADH_PLAN_ERR_CYCLE_BOUNDARY = 3,
    ADH_PLAN_ERR_SCAN_LIMITS = 4,
    ADH_PLAN_ERR_ALPHARAW_SCANS = 5,
    ADH_PLAN_ERR_CHARGE = 6,
    ADH_PLAN_ERR_TOO_MANY_OBS = 7,
    ADH_PLAN_ERR_TOO_MANY_MS1 = 8,
};

struct PlanMeta {
    int32_t err, pad;
    int32_t all_k, all_o, all_f, all_n_lib, all_s, all_op;  // maxima over the batch
    int32_t gen_k, gen_o, gen_f, gen_n_lib;                 // maxima over the generic class
    uint32_t class_first[ADH_N_CLASSES + 1];                // first processing position of every class
    uint64_t scratch_bytes;
};

struct PlanArgs {
    int64_t row0, n;        // first candidate row of the batch, rows in it
    int64_t n_frames;       // spectra (AlphaRaw) / frames (ion mobility) of the run
    int64_t n_lib;          // fragments of the staged library
    int32_t L;              // cycle length
    int32_t rows;           // cycle rows (AlphaRaw: L * cycle_scans)
    int32_t scan_max;       // ion mobility: scans per frame
    int32_t zeroth;         // ion mobility: 1 when frame 0 is the empty alphatims frame
    int32_t I;              // isotopes used
    uint32_t top_k;
    int32_t fast_cfg, quant_all;  // fast_cfg: bit 0 the register kernels, bit 1 their wide form
    int32_t fused_cfg;      // the fused kernel may be used (one MS1 row per cycle, <= 3 isotopes): bit 0 for one, bit 1 for two observations
    int32_t n_cyc_bins;     // first-cycle bins per class in the sort key
};

namespace plan {

constexpr double ISOTOPE_DELTA = 1.0033548350700006;  // candidate.py:160

// maximum over the wavefront, then ONE atomic per wavefront and slot (a quarter of a million
// same-address atomics per launch cost 0.3 ms)
__device__ __forceinline__ void raise_max(int32_t *slot, int32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    if ((threadIdx.x & 63) == 0 && v > *reinterpret_cast<volatile int32_t *>(slot)) atomicMax(slot, v);
}

// isotope m/z range -> quadrupole query range exactly as the kernels compute it
// (candidate.py:151-163,203-205)
__device__ __forceinline__ void quad_range(float precursor_mz, uint8_t charge, int I, float &q_lo, float &q_hi) {
    float mn = 0.f, mx = 0.f;
    for (int k = 0; k < I; ++k) {
        float m = (float)((double)k * ISOTOPE_DELTA / (double)charge) + precursor_mz;
        if (k == 0 || m < mn) mn = m;
        if (k == 0 || m > mx) mx = m;
    }
    q_lo = (float)((double)mn - 0.5);
    q_hi = (float)((double)mx + 0.5);
}

}  // namespace plan

__global__ __launch_bounds__(256) void adh_plan_rec_kernel(DevCands c, const double *__restrict__ cyc, PlanArgs p,
                                                           CandRec *__restrict__ recs, uint32_t *__restrict__ keys,
                                                           uint32_t *__restrict__ idx, uint64_t *__restrict__ bytes,
                                                           uint32_t *__restrict__ hist, PlanMeta *__restrict__ meta) {
    const int64_t j0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = j0 < p.n;          // dead lanes of the last block ride along (wave-wide reductions below)
    const int64_t j = live ? j0 : p.n - 1;
    const int64_t i = p.row0 + j;
    CandRec r;
    memset(&r, 0, sizeof(r));
    r.precursor_idx = c.precursor_idx[i];
    r.frag_start = c.frag_start[i];
    r.frag_stop = c.frag_stop[i];
    const int64_t fs = c.frame_start[i], fe = c.frame_stop[i], fc = c.frame_center[i];
    const int64_t ss = c.scan_start[i], se = c.scan_stop[i], sc = c.scan_center[i];
    r.frame_start = (int32_t)fs;
    r.frame_stop = (int32_t)fe;
    r.frame_center = (int32_t)fc;
    r.scan_start = (int32_t)ss;
    r.scan_stop = (int32_t)se;
    r.scan_center = (int32_t)sc;
    r.precursor_mz = c.precursor_mz[i];
    r.charge = c.charge[i];
    r.rank = c.rank[i];
    r.flags = c.flags ? c.flags[i] : (uint8_t)0;
    r.row = (uint32_t)i;
    int cls = ADH_CLASS_GENERIC;
    uint32_t bin = 0;
    uint64_t nbytes = 0;
    int32_t mx_k = 0, mx_o = 0, mx_f = 0, mx_l = 0, gx_k = 0, gx_o = 0, gx_f = 0, gx_l = 0;
    if (!(r.flags & ADH_FLAG_SKIP)) {
        int err = 0;
        if (r.frag_stop < r.frag_start || (int64_t)r.frag_stop > p.n_lib) err = ADH_PLAN_ERR_FRAG_SLICE;
        else if (fs < 0 || fe < fs || fe > p.n_frames || fc < 0 || fc >= p.n_frames) err = ADH_PLAN_ERR_FRAME_LIMITS;
        else if (fs % p.L != 0) err = ADH_PLAN_ERR_CYCLE_BOUNDARY;
        else if (se - ss != 1 || ss != 0 || sc != 0) err = ADH_PLAN_ERR_ALPHARAW_SCANS;
        else if (r.charge == 0) err = ADH_PLAN_ERR_CHARGE;
        int O = 0;
        if (!err) {
            float q_lo, q_hi;
            plan::quad_range(r.precursor_mz, r.charge, p.I, q_lo, q_hi);
            for (int row = 0; row < p.rows; ++row) {
                if ((double)q_lo <= cyc[2 * row + 1] && (double)q_hi >= cyc[2 * row]) {
                    if (O >= ADH_MAX_OBS) {
                        err = ADH_PLAN_ERR_TOO_MANY_OBS;
                        break;
                    }
                    r.obs[O++] = (uint16_t)row;
                }
            }
        }
        if (err) {
            atomicMax(&meta->err, err);
            r.flags |= ADH_FLAG_SKIP;  // never reaches a scoring kernel: the host refuses the batch
        } else {
            r.n_obs = (uint8_t)O;
            const int64_t nl = (int64_t)r.frag_stop - (int64_t)r.frag_start;
            r.k_cap = (uint32_t)min((int64_t)p.top_k, nl);
            const int F = r.frame_stop / p.L - r.frame_start / p.L;
            // shape handled by the register-resident kernels (adh_features_fast.hip); several
            // observations only with quant_all
            const bool shape_any_k = (p.fast_cfg & 1) && O >= 1 && O <= ADH_PLAN_FAST_OMAX && F >= 3 && F <= ADH_PLAN_FMAX && p.I <= 4;
            const bool shape = shape_any_k && r.k_cap <= 16;
            const bool fast = shape && (O == 1 || p.quant_all);
            const bool wide = shape_any_k && (p.fast_cfg & 2) && r.k_cap > 16 && r.k_cap <= ADH_PLAN_WIDE_KMAX && (O == 1 || p.quant_all);
            // gather and features in one kernel (adh_fused.hip): lanes 12..15 of a 16-lane group carry the isotopes;
            // it quantifies the best of two observations itself when quant_all is off (round 4)
            const bool fused = shape && ((p.fused_cfg >> (O - 1)) & 1) && r.k_cap <= 12 && nl <= 64;
            cls = fused ? (O == 1 ? ADH_CLASS_FUSED0 : ADH_CLASS_FUSED2) + max(F - 5, 0) / 4
                        : (wide ? (r.k_cap <= 32 ? (O == 1 ? ADH_CLASS_MID1 : ADH_CLASS_MID2) : (O == 1 ? ADH_CLASS_WIDE1 : ADH_CLASS_WIDE2)) +
                                      (F <= 16 ? 0 : (F <= 24 ? 1 : 2))
                           : (!fast ? ADH_CLASS_GENERIC
                                    : (O == 1 ? ADH_CLASS_FAST1 + max(F - 5, 0) / 4
                                              : ADH_CLASS_FAST2 + (F <= 16 ? 0 : (F <= 24 ? 1 : 2)))));
            bin = (uint32_t)(r.frame_start / p.L);
            nbytes = fused ? 0 : adh_scratch_bytes(r.k_cap, O, max(F, 0), p.I);  // nothing leaves the CU there
            if (live) {
                mx_k = (int32_t)r.k_cap, mx_o = O, mx_f = F, mx_l = (int32_t)nl;
                if (cls == ADH_CLASS_GENERIC) gx_k = mx_k, gx_o = mx_o, gx_f = mx_f, gx_l = mx_l;
            }
        }
    }
    plan::raise_max(&meta->all_k, mx_k);
    plan::raise_max(&meta->all_o, mx_o);
    plan::raise_max(&meta->all_f, mx_f);
    plan::raise_max(&meta->all_n_lib, mx_l);
    plan::raise_max(&meta->gen_k, gx_k);
    plan::raise_max(&meta->gen_o, gx_o);
    plan::raise_max(&meta->gen_f, gx_f);
    plan::raise_max(&meta->gen_n_lib, gx_l);
    if (!live) return;
    const uint32_t key = (uint32_t)cls * (uint32_t)p.n_cyc_bins + min(bin, (uint32_t)p.n_cyc_bins - 1u);
    recs[j] = r;
    keys[j] = key;
    idx[j] = (uint32_t)j;
    bytes[j] = nbytes;
    if (hist) atomicAdd(&hist[key], 1u);  // counting sort by (class, first cycle)
}

// Ion-mobility plan: observation lists = sorted unique dia_precursor_cycle values of the cycle rows
// inside the scan range that overlap the fragment / the (-1, -1) precursor quadrupole range.
__global__ __launch_bounds__(256) void adh_plan_rec_im_kernel(DevCands c, const double *__restrict__ cyc,
                                                              const int32_t *__restrict__ dpc, PlanArgs p,
                                                              CandRecIM *__restrict__ recs, uint32_t *__restrict__ keys,
                                                              uint32_t *__restrict__ idx, uint64_t *__restrict__ bytes,
                                                              uint32_t *__restrict__ hist, PlanMeta *__restrict__ meta) {
    const int64_t j0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = j0 < p.n;
    const int64_t j = live ? j0 : p.n - 1;
    const int64_t i = p.row0 + j;
    CandRecIM r;
    memset(&r, 0, sizeof(r));
    r.precursor_idx = c.precursor_idx[i];
    r.frag_start = c.frag_start[i];
    r.frag_stop = c.frag_stop[i];
    const int64_t fs = c.frame_start[i], fe = c.frame_stop[i], fc = c.frame_center[i];
    const int64_t ss = c.scan_start[i], se = c.scan_stop[i], sc = c.scan_center[i];
    r.frame_start = (int32_t)fs;
    r.frame_stop = (int32_t)fe;
    r.frame_center = (int32_t)fc;
    r.scan_start = (int32_t)ss;
    r.scan_stop = (int32_t)se;
    r.scan_center = (int32_t)sc;
    r.precursor_mz = c.precursor_mz[i];
    r.charge = c.charge[i];
    r.rank = c.rank[i];
    r.flags = c.flags ? c.flags[i] : (uint8_t)0;
    r.row = (uint32_t)i;
    uint32_t bin = 0;
    uint64_t nbytes = 0;
    int32_t mx_k = 0, mx_o = 0, mx_f = 0, mx_s = 0, mx_p = 0, mx_l = 0;
    bool small_tile = false;
    if (!(r.flags & ADH_FLAG_SKIP)) {
        int err = 0;
        const int64_t z = p.zeroth;
        if (r.frag_stop < r.frag_start || (int64_t)r.frag_stop > p.n_lib) err = ADH_PLAN_ERR_FRAG_SLICE;
        else if (fs < z || fe < fs || fe > p.n_frames || fc < 0 || fc >= p.n_frames) err = ADH_PLAN_ERR_FRAME_LIMITS;
        else if ((fs - z) % p.L != 0) err = ADH_PLAN_ERR_CYCLE_BOUNDARY;
        else if (ss < 0 || se < ss || se > p.scan_max || sc < 0 || sc >= p.scan_max) err = ADH_PLAN_ERR_SCAN_LIMITS;
        else if (r.charge == 0) err = ADH_PLAN_ERR_CHARGE;
        if (!err) {
            float qf_lo, qf_hi;
            plan::quad_range(r.precursor_mz, r.charge, p.I, qf_lo, qf_hi);
            const double q_lo = (double)qf_lo, q_hi = (double)qf_hi;
            uint32_t seen_f[32], seen_p[32];  // bit v: cycle row value v was hit (L <= 1024, checked by the host)
            for (int w = 0; w < 32; ++w) seen_f[w] = seen_p[w] = 0u;
            for (int fr = 0; fr < p.L; ++fr)
                for (int s = (int)ss; s < (int)se; ++s) {
                    const int64_t rowi = (int64_t)fr * p.scan_max + s;
                    const double wl = cyc[2 * rowi], wh = cyc[2 * rowi + 1];
                    const int v = dpc[rowi];
                    if (q_lo <= wh && q_hi >= wl) {
                        seen_f[v >> 5] |= 1u << (v & 31);
                        if (fr < 32) r.frames_f |= 1u << fr;
                    }
                    if (-1.0 <= wh && -1.0 >= wl) {
                        seen_p[v >> 5] |= 1u << (v & 31);
                        if (fr < 32) r.frames_p |= 1u << fr;
                    }
                }
            int nf = 0, np = 0;
            for (int v = 0; v < p.L && !err; ++v) {
                if (seen_f[v >> 5] >> (v & 31) & 1u) {
                    if (nf >= ADH_MAX_OBS) err = ADH_PLAN_ERR_TOO_MANY_OBS;
                    else r.obs[nf++] = (uint16_t)v;
                }
                if (seen_p[v >> 5] >> (v & 31) & 1u) {
                    if (np >= ADH_MAX_MS1_OBS) err = ADH_PLAN_ERR_TOO_MANY_MS1;
                    else r.ms1_obs[np++] = (uint16_t)v;
                }
            }
            r.n_obs = (uint8_t)nf;
            r.n_ms1 = (uint8_t)np;
        }
        if (err) {
            atomicMax(&meta->err, err);
            r.flags |= ADH_FLAG_SKIP;
            r.n_obs = r.n_ms1 = 0;
        } else {
            const int64_t nl = (int64_t)r.frag_stop - (int64_t)r.frag_start;
            r.k_cap = (uint32_t)min((int64_t)p.top_k, nl);
            const int z32 = p.zeroth;
            const int F = max((r.frame_stop - z32) / p.L - (r.frame_start - z32) / p.L, 0);
            const int S = max(r.scan_stop - r.scan_start, 0);
            bin = (uint32_t)((r.frame_start - z32) / p.L);
            nbytes = adh_im_scratch_bytes(r.k_cap, r.n_obs, S, F, p.I, r.n_ms1);
            if (live) mx_k = (int32_t)r.k_cap, mx_o = r.n_obs, mx_f = F, mx_s = S, mx_p = r.n_ms1, mx_l = (int32_t)nl;
            small_tile = r.k_cap <= ADH_IM_SMALL_K && S <= ADH_IM_SMALL_S && F <= ADH_IM_SMALL_F && S * F <= ADH_IM_SMALL_SF;
        }
    }
    plan::raise_max(&meta->all_k, mx_k);
    plan::raise_max(&meta->all_o, mx_o);
    plan::raise_max(&meta->all_f, mx_f);
    plan::raise_max(&meta->all_s, mx_s);
    plan::raise_max(&meta->all_op, mx_p);
    plan::raise_max(&meta->all_n_lib, mx_l);
    if (!live) return;
    // feature launches by observation count (class 0: one, class 1: two, generic: more) and, with one observation,
    // by tile size (class ADH_CLASS_IM_SMALL): the LDS of the ion-mobility feature kernel scales with both, its
    // resident blocks with the LDS and its time with the resident blocks; most precursors sit in one isolation window
    const int cls = (r.flags & ADH_FLAG_SKIP) ? ADH_CLASS_GENERIC
                    : (r.n_obs <= 1 ? (small_tile ? ADH_CLASS_IM_SMALL : 0) : (r.n_obs == 2 ? 1 : ADH_CLASS_GENERIC));
    const uint32_t key = (uint32_t)cls * (uint32_t)p.n_cyc_bins + min(bin, (uint32_t)p.n_cyc_bins - 1u);
    recs[j] = r;
    keys[j] = key;
    idx[j] = (uint32_t)j;
    bytes[j] = nbytes;
    if (hist) atomicAdd(&hist[key], 1u);
}

// counting sort, scatter step: position = running cursor of the key (order inside one (class, cycle)
// bin is arbitrary: results do not depend on the processing order, only the locality does)
__global__ void adh_plan_scatter_kernel(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ bytes, int64_t n,
                                        uint32_t *__restrict__ cursor, uint32_t *__restrict__ sorted_keys,
                                        uint32_t *__restrict__ idx_out, uint64_t *__restrict__ sorted_bytes) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t key = keys[j];
    const uint32_t pos = atomicAdd(&cursor[key], 1u);
    sorted_keys[pos] = key;
    idx_out[pos] = (uint32_t)j;
    sorted_bytes[pos] = bytes[j];
}


__global__ void adh_plan_take_bytes_kernel(const uint64_t *__restrict__ bytes, const uint32_t *__restrict__ idx,
                                           int64_t n, uint64_t *__restrict__ sorted_bytes) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) sorted_bytes[j] = bytes[idx[j]];
}

// records into processing order with their scratch offsets; class boundaries; scratch size
template <typename Rec>
__global__ void adh_plan_order_kernel(const Rec *__restrict__ recs, const uint32_t *__restrict__ sorted_keys,
                                      const uint32_t *__restrict__ idx, const uint64_t *__restrict__ offs,
                                      const uint64_t *__restrict__ sorted_bytes, int64_t n, uint32_t n_cyc_bins,
                                      Rec *__restrict__ ordered, PlanMeta *__restrict__ meta) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    Rec r = recs[idx[j]];
    r.scratch_off = offs[j];
    ordered[j] = r;
    const int cj = (int)(sorted_keys[j] / n_cyc_bins);
    const int cp = j > 0 ? (int)(sorted_keys[j - 1] / n_cyc_bins) : -1;
    for (int c = cp + 1; c <= cj; ++c) meta->class_first[c] = (uint32_t)j;
    if (j == n - 1) {
        for (int c = cj + 1; c <= ADH_N_CLASSES; ++c) meta->class_first[c] = (uint32_t)n;
        meta->scratch_bytes = offs[j] + sorted_bytes[j];
    }
}
