// adh_select_im.hip - candidate selection on ion-mobility (timsTOF) runs.
//
// Same step as adh_select.hip with a real scan axis (_select_candidates_pjit / _build_candidates,
// alphadia/search/selection/selection.py:78-526): tiles are (scans x cycles), the smoothing is
// two-dimensional and peaks are picked in both directions (find_peaks_2d,
// selection/utils.py:80-115).  Two kernels per batch of precursors:
//
//   adh_select_gather_im_kernel   one wavefront per precursor, little LDS, many resident waves:
//       isotope / fragment windows, then one lane per window walks the TOF bins of the window
//       and adds the detector events of the tile's cycles that fall into the scan range and whose
//       quadrupole window overlaps the precursor - TimsTOFTransposeJIT.get_dense_intensity
//       (alphadia/search/jitclasses/bruker_jit.py:506-645) - into a float32 tile in HBM scratch.
//       A cell is touched by exactly one lane in (TOF index, push) order, so its running float32
//       sum is the reference's.
//   adh_select_score_im_kernel    one 256-thread workgroup per precursor: every tile is smoothed
//       with the separable form of the Gaussian kernel (circular, float64 accumulation, one
//       rounding per pass; the reference uses a float32 FFT), log(smooth + 1) is summed over
//       fragments and isotopes, then peaks, joins and symmetric limits as in the reference.
#include "adh_device.h"

namespace selim {

constexpr int MAX_W = 64;    // m/z windows (fragments + isotopes) per precursor
constexpr int MAX_CAND = 16;
constexpr int SCORE_THREADS = 256;

// per-precursor record prepared by the host (frame / scan limits as the reference computes them)
struct __attribute__((aligned(16))) PrecRec {
    uint32_t precursor_idx, frag_start, frag_stop;
    float mz;
    int32_t cycle_start, n_cycles;   // first cycle, F
    int32_t scan_start, n_scans;     // first scan, S
    uint64_t scratch_off;            // bytes: header (32) + [W][S][F] float
    uint8_t charge, ok, pad[6];
};
static_assert(sizeof(PrecRec) == 48, "PrecRec must be 48 bytes");

}  // namespace selim

__global__ __launch_bounds__(ADH_WAVE) void adh_select_gather_im_kernel(
    DevTims run, const LibRec *__restrict__ lib, const selim::PrecRec *__restrict__ recs, int32_t n_prec,
    adh_selection_config_t cfg, int32_t n_iso, unsigned char *__restrict__ scratch) {
    using namespace selim;
    __shared__ float s_mz[MAX_W];       // window centres: fragments ascending, then isotopes
    __shared__ int s_tlo[MAX_W], s_thi[MAX_W];
    __shared__ float s_raw[MAX_W];
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    if (i >= n_prec) return;
    const PrecRec r = recs[i];
    uint32_t *header = reinterpret_cast<uint32_t *>(scratch + r.scratch_off);
    if (!r.ok) {
        if (lane == 0) header[0] = 0;
        return;
    }
    // fragments: slice, cardinality filter, sort by m/z (selection.py:124-139)
    const int n_lib = min((int)(r.frag_stop - r.frag_start), MAX_W);
    if (lane < n_lib) {
        const LibRec f = lib[r.frag_start + lane];
        s_raw[lane] = (cfg.exclude_shared_ions && f.cardinality > 1) ? -1.0f : f.mz;
    }
    __syncthreads();
    int K = 0;
    for (int j = 0; j < n_lib; ++j) K += s_raw[j] >= 0.0f;
    if (K <= 3 || K + n_iso > MAX_W) {  // selection.py:141 (more than MAX_W windows: rejected by the host)
        if (lane == 0) header[0] = 0;
        return;
    }
    if (lane < n_lib && s_raw[lane] >= 0.0f) {
        const float ma = s_raw[lane];
        int slot = 0;
        for (int b = 0; b < n_lib; ++b) {
            const float mb = s_raw[b];
            if (mb < 0.0f) continue;
            slot += (mb < ma) || (mb == ma && b < lane);
        }
        s_mz[slot] = ma;
    }
    if (lane < n_iso)  // assemble_isotope_mz (selection/utils.py:24-46)
        s_mz[K + lane] = (float)((double)r.mz + (double)lane * 1.0033548350700006 / (double)r.charge);
    __syncthreads();
    const int W = K + n_iso;
    // TOF index limits of every window: searchsorted(mz_values, mass_range(...), "left")
    if (lane < W) {
        const float m = s_mz[lane];
        const float tol = (float)(lane < K ? cfg.fragment_mz_tolerance : cfg.precursor_mz_tolerance);
        float t = tol * m;
        float q = t / 1000000.0f;
        const double lo = (double)(m - q), hi = (double)(m + q);
        int64_t a = 0, b = run.n_tof;
        while (a < b) {
            const int64_t mid = (a + b) >> 1;
            if (run.mz[mid] < lo) a = mid + 1; else b = mid;
        }
        s_tlo[lane] = (int)a;
        b = run.n_tof;
        while (a < b) {
            const int64_t mid = (a + b) >> 1;
            if (run.mz[mid] < hi) a = mid + 1; else b = mid;
        }
        s_thi[lane] = (int)a;
    }
    const int S = r.n_scans, F = r.n_cycles, L = run.cycle_len, SM = run.scan_max, z = run.zeroth;
    float *tiles = reinterpret_cast<float *>(scratch + r.scratch_off + 32);
    const int n_cells = W * S * F;
    for (int c = lane; c < n_cells; c += ADH_WAVE) tiles[c] = 0.0f;
    if (lane == 0) {
        header[0] = (uint32_t)K;
        header[1] = (uint32_t)W;
    }
    __syncthreads();
    const double q_lo = (double)s_mz[K], q_hi = (double)s_mz[K + n_iso - 1];
    // ---- one lane per window: per TOF bin ONE binary search for the first push of the tile's first
    // cycle, then the bin's events of all F cycles in storage order (they are contiguous: pushes
    // ascend with the frame).  A lane per (window, cycle) would repeat the search F times - 7e9
    // dependent HBM probes for 200 000 precursors on the full-size run.  A cell is still touched by
    // exactly one lane in (TOF index, push) order, so its running float32 sum is the reference's.
    const uint32_t push_lo = (uint32_t)(r.cycle_start * L + z) * (uint32_t)SM;
    const uint32_t push_hi = (uint32_t)((r.cycle_start + F) * L + z) * (uint32_t)SM;
    for (int w = lane; w < W; w += ADH_WAVE) {
        const bool prec = w >= K;
        const double ql = prec ? -1.0 : q_lo, qh = prec ? -1.0 : q_hi;
        float *cells = tiles + (size_t)w * S * F;
        for (int tof = s_tlo[w]; tof < s_thi[w]; ++tof) {
            const int64_t b = run.tof_indptr[tof + 1];
            int64_t lo = run.tof_indptr[tof], hi = b;
            while (lo < hi) {
                const int64_t m = (lo + hi) >> 1;
                if (run.push[m] < push_lo) lo = m + 1; else hi = m;
            }
            for (int64_t e = lo; e < b; ++e) {
                const uint32_t p = run.push[e];
                if (p >= push_hi) break;
                const int frame = (int)(p / (uint32_t)SM), scan = (int)(p % (uint32_t)SM);
                if (scan < r.scan_start || scan >= r.scan_start + S) continue;
                const int fr = frame - z;
                const int cyc = fr / L;
                const int crow = (fr - cyc * L) * SM + scan;
                if (!(ql <= run.cycle[2 * crow + 1] && qh >= run.cycle[2 * crow])) continue;
                float *c = cells + (size_t)(scan - r.scan_start) * F + (cyc - r.cycle_start);
                *c = *c + (float)run.inten[e];  // bruker_jit.py:575-580: float32 running sum
            }
        }
    }
}

__global__ __launch_bounds__(selim::SCORE_THREADS) void adh_select_score_im_kernel(
    DevTims run, const selim::PrecRec *__restrict__ recs, int32_t n_prec, int64_t first_prec,
    adh_selection_config_t cfg, const double *__restrict__ ku_g, const double *__restrict__ kv_g, int32_t k0,
    int32_t k1, int32_t cap_cells, int32_t cap_tp, int32_t cap_mp, int32_t cap_s, int32_t cap_f,
    unsigned char *__restrict__ scratch, DevCandTable out) {
    using namespace selim;
    extern __shared__ __align__(16) unsigned char smem[];
    // tile padded along the cycles / pass-1 result padded along the scans (circular copies in the
    // pads, so the convolution loops carry no wrap-around logic), then the two log-sum tiles
    float *tile = reinterpret_cast<float *>(smem);
    float *tmp = tile + cap_tp;
    float *ls = tmp + cap_mp;  // log-sum tile: fragments first (parked in HBM when done), then isotopes
    double *ku = reinterpret_cast<double *>(ls + cap_cells);
    double *kv = ku + k0;
    double *mob = kv + k1, *cyc = mob + cap_s;  // scan / cycle profiles of symetric_limits_2d
    unsigned char *flag = reinterpret_cast<unsigned char *>(cyc + cap_f);
    __shared__ double red_v[SCORE_THREADS];
    __shared__ int red_i[SCORE_THREADS];
    __shared__ int pk_idx[MAX_CAND];
    __shared__ double pk_val[MAX_CAND];
    __shared__ double s_norm[2];
    const int tid = threadIdx.x;
    const int i = blockIdx.x;
    if (i >= n_prec) return;
    const PrecRec r = recs[i];
    const uint32_t *header = reinterpret_cast<const uint32_t *>(scratch + r.scratch_off);
    const int K = (int)header[0], W = (int)header[1];
    if (!r.ok || K == 0) return;
    const float *tiles = reinterpret_cast<const float *>(scratch + r.scratch_off + 32);
    float *park = reinterpret_cast<float *>(scratch + r.scratch_off + 32);  // tile 0, free once smoothed
    const int S = r.n_scans, F = r.n_cycles, SF = S * F;
    for (int c = tid; c < k0; c += SCORE_THREADS) ku[c] = ku_g[c];
    for (int c = tid; c < k1; c += SCORE_THREADS) kv[c] = kv_g[c];
    for (int c = tid; c < SF; c += SCORE_THREADS) {
        ls[c] = 0.0f;
    }
    __syncthreads();
    // circular convolution, separable: out(s, f) = sum_a ku[a] * (sum_b kv[b] * x[(s + k0/2 - a) mod S][(f + k1/2 - b) mod F]),
    // float64 fused multiply-adds in tap order (as in the oracle), one rounding to float32 per pass
    const int h0 = k0 / 2, h1 = k1 / 2;
    const int padl = k1 - h1 - 1, FP = F + k1;   // padded row: column j holds cycle (j - padl) mod F
    const int padt = k0 - h0 - 1;                // padded pass-1 tile: row j holds scan (j - padt) mod S
    const bool fast = k0 == 30 && k1 == 30;      // the default kernel: taps in registers, loops unrolled
    for (int w = 0; w < W; ++w) {
        for (int c = tid; c < S * FP; c += SCORE_THREADS) {
            const int sc = c / FP, j = c - sc * FP;
            int src = j - padl;  // one wrap suffices: F >= k1 (_is_valid)
            src += (src < 0) ? F : 0;
            src -= (src >= F) ? F : 0;
            tile[c] = tiles[(size_t)w * SF + sc * F + src];
        }
        __syncthreads();
        // pass 1: along the cycles, kernel centred at column k1 / 2 (default kernel: taps in registers,
        // loop unrolled; four outputs per thread sharing one converted window measured no faster)
        {
            double kr[30];
            if (fast) {
#pragma unroll
                for (int b = 0; b < 30; ++b) kr[b] = kv[b];
            }
            for (int c = tid; c < SF; c += SCORE_THREADS) {
                const int sc = c / F, f = c - sc * F;
                const float *rp = tile + sc * FP + padl + f + h1;  // rp[-b] = x[sc][(f + h1 - b) mod F]
                double acc = 0.0;
                if (fast) {
#pragma unroll
                    for (int b = 0; b < 30; ++b) acc = fma(kr[b], (double)rp[-b], acc);
                } else {
                    for (int b = 0; b < k1; ++b) acc = fma(kv[b], (double)rp[-b], acc);
                }
                const float v = (float)acc;
                tmp[(padt + sc) * F + f] = v;
                if (sc < h0) tmp[(padt + S + sc) * F + f] = v;            // copy below the last scan
                if (sc >= S - padt) tmp[(sc - (S - padt)) * F + f] = v;  // copy above the first (S >= k0, _is_valid)
            }
        }
        __syncthreads();
        // pass 2: along the scans, kernel centred at row k0 / 2; log(smooth + 1) summed per group
        {
            float *lsum = ls;
            double kr[30];
            if (fast) {
#pragma unroll
                for (int a = 0; a < 30; ++a) kr[a] = ku[a];
            }
            for (int c = tid; c < SF; c += SCORE_THREADS) {
                const int sc = c / F, f = c - sc * F;
                const float *cp = tmp + (padt + sc + h0) * F + f;  // cp[-a * F] = pass1[(sc + h0 - a) mod S][f]
                double acc = 0.0;
                if (fast) {
#pragma unroll
                    for (int a = 0; a < 30; ++a) acc = fma(kr[a], (double)cp[-a * F], acc);
                } else {
                    for (int a = 0; a < k0; ++a) acc = fma(ku[a], (double)cp[-a * F], acc);
                }
                const float sm = (float)acc;
                lsum[c] += (float)log((double)(sm + 1.0f));  // _build_features (selection.py:206-226)
            }
            if (w == K - 1) {
                // fragment sum complete: park it in the (consumed) first tile of the scratch block, the
                // LDS array starts over for the isotopes (every thread owns the same cells in both loops)
                for (int c = tid; c < SF; c += SCORE_THREADS) {
                    park[c] = ls[c];
                    ls[c] = 0.0f;
                }
            }
        }
        __syncthreads();
    }
    // feature = fragment sum + isotope sum (float32), in place
    for (int c = tid; c < SF; c += SCORE_THREADS) ls[c] = park[c] + ls[c];
    __syncthreads();
    // ---- score (selection.py:396-421): kept as the float32 feature + the affine map
    double mean = cfg.feature_mean, sd = cfg.feature_std, weight = cfg.feature_weight;
    if (!cfg.use_weighted_score) {
        if (tid == 0) {  // amean1 / astd1 (selection/utils.py:118-133), sequential
            double m = 0;
            for (int c = 0; c < SF; ++c) m += (double)ls[c];
            m /= (double)SF;
            double v = 0;
            for (int c = 0; c < SF; ++c) {
                const double d = (double)ls[c] - m;
                v += d * d;
            }
            s_norm[0] = m;
            s_norm[1] = sqrt(v / (double)SF);
        }
        __syncthreads();
        mean = s_norm[0];
        sd = s_norm[1];
        weight = 1.0;
    }
    double *score = reinterpret_cast<double *>(tile);  // tile + tmp hold SF doubles
    for (int c = tid; c < SF; c += SCORE_THREADS) {
        const float ft = ls[c];
        score[c] = weight * ((double)ft - mean) / (sd + 1e-6);
    }
    __syncthreads();
    auto A = [&](int s, int f) { return score[s * F + f]; };
    // ---- find_peaks_2d / find_peaks_1d (selection/utils.py:49-115): flag, then top_n rounds of a
    // block-wide arg-max with key (score, linear index): equal scores in reversed index order
    for (int c = tid; c < SF; c += SCORE_THREADS) {
        const int s = c / F, p = c - s * F;
        bool pk = false;
        if (S <= 2) {
            pk = s == 0 && p >= 2 && p < F - 2 && A(0, p - 2) < A(0, p - 1) && A(0, p - 1) < A(0, p) &&
                 A(0, p) > A(0, p + 1) && A(0, p + 1) > A(0, p + 2);
        } else if (s >= 2 && s < S - 2 && p >= 2 && p < F - 2) {
            pk = A(s - 2, p) < A(s - 1, p) && A(s - 1, p) < A(s, p) && A(s, p) > A(s + 1, p) && A(s + 1, p) > A(s + 2, p);
            pk = pk && A(s, p - 2) < A(s, p - 1) && A(s, p - 1) < A(s, p) && A(s, p) > A(s, p + 1) && A(s, p + 1) > A(s, p + 2);
        }
        flag[c] = pk ? 1 : 0;
    }
    __syncthreads();
    const int top_n = (int)min((int64_t)MAX_CAND, cfg.candidate_count);
    int n_pk = 0;
    for (int round = 0; round < top_n; ++round) {
        double best = -INFINITY;
        int best_i = -1;
        for (int c = tid; c < SF; c += SCORE_THREADS)
            if (flag[c]) {
                const double v = score[c];
                if (best_i < 0 || v > best || (v == best && c > best_i)) {
                    best = v;
                    best_i = c;
                }
            }
        red_v[tid] = best;
        red_i[tid] = best_i;
        __syncthreads();
        for (int off = SCORE_THREADS / 2; off > 0; off >>= 1) {
            if (tid < off) {
                const int j = red_i[tid + off];
                if (j >= 0) {
                    const double v = red_v[tid + off];
                    const int ci = red_i[tid];
                    if (ci < 0 || v > red_v[tid] || (v == red_v[tid] && j > ci)) {
                        red_v[tid] = v;
                        red_i[tid] = j;
                    }
                }
            }
            __syncthreads();
        }
        const int win = red_i[0];
        if (win < 0) break;  // uniform
        if (tid == 0) {
            pk_idx[n_pk] = win;
            pk_val[n_pk] = red_v[0];
            flag[win] = 0;
        }
        ++n_pk;
        __syncthreads();
    }
    if (tid != 0) return;
    // ---- joins and limits by one thread (short loops), in the reference's order
    int p_scan[MAX_CAND], p_cycle[MAX_CAND], p_sl[MAX_CAND][2], p_cl[MAX_CAND][2];
    double p_score[MAX_CAND];
    for (int a = 0; a < n_pk; ++a) {
        p_scan[a] = pk_idx[a] / F;
        p_cycle[a] = pk_idx[a] - p_scan[a] * F;
        p_score[a] = pk_val[a];
    }
    {  // _join_close_peaks (selection.py:229-278), tolerances 3 / 3
        bool mask[MAX_CAND];
        for (int a = 0; a < n_pk; ++a) mask[a] = true;
        for (int a = 0; a < n_pk; ++a) {
            if (!mask[a]) continue;
            for (int b = a + 1; b < n_pk; ++b) {
                if (!mask[b]) continue;
                if (abs(p_scan[a] - p_scan[b]) <= 3 && abs(p_cycle[a] - p_cycle[b]) <= 3) {
                    if (p_score[a] > p_score[b]) mask[b] = false; else mask[a] = false;
                }
            }
        }
        int m = 0;
        for (int a = 0; a < n_pk; ++a)
            if (mask[a]) {
                p_scan[m] = p_scan[a];
                p_cycle[m] = p_cycle[a];
                p_score[m] = p_score[a];
                ++m;
            }
        n_pk = m;
    }
    // symetric_limits_2d (selection/utils.py:283-312)
    for (int a = 0; a < n_pk; ++a) {
        const int mob_lower = (int)max((int64_t)0, (int64_t)p_scan[a] - cfg.min_size_mobility);
        const int mob_upper = (int)min((int64_t)S, (int64_t)p_scan[a] + cfg.min_size_mobility);
        const int cyc_lower = (int)max((int64_t)0, (int64_t)p_cycle[a] - cfg.min_size_rt);
        const int cyc_upper = (int)min((int64_t)F, (int64_t)p_cycle[a] + cfg.min_size_rt);
        for (int s = 0; s < S; ++s) {
            double v = 0.0;
            for (int f = cyc_lower; f < cyc_upper; ++f) v += A(s, f);
            mob[s] = v;
        }
        for (int f = 0; f < F; ++f) cyc[f] = 0.0;
        for (int s = mob_lower; s < mob_upper; ++s)
            for (int f = 0; f < F; ++f) cyc[f] += A(s, f);
        sel::symetric_limits_1d(mob, S, p_scan[a], cfg.f_mobility, cfg.center_fraction, cfg.min_size_mobility,
                                cfg.max_size_mobility, p_sl[a]);
        sel::symetric_limits_1d(cyc, F, p_cycle[a], cfg.f_rt, cfg.center_fraction, cfg.min_size_rt, cfg.max_size_rt,
                                p_cl[a]);
    }
    if (cfg.join_close_candidates) {  // _join_overlapping_candidates (selection.py:281-345)
        bool mask[MAX_CAND];
        for (int a = 0; a < n_pk; ++a) mask[a] = true;
        for (int a = 0; a < n_pk; ++a) {
            if (!mask[a]) continue;
            for (int b = a + 1; b < n_pk; ++b) {
                if (!mask[b]) continue;
                const double cyc_ov = (double)(min(p_cl[a][1], p_cl[b][1]) - max(p_cl[a][0], p_cl[b][0])) /
                                      (double)(p_cl[a][1] - p_cl[a][0]);
                const double scan_ov = (double)(min(p_sl[a][1], p_sl[b][1]) - max(p_sl[a][0], p_sl[b][0])) /
                                       (double)(p_sl[a][1] - p_sl[a][0]);
                if (scan_ov < 0 || cyc_ov < 0) continue;
                if (cyc_ov > cfg.join_close_candidates_cycle_threshold &&
                    scan_ov > cfg.join_close_candidates_scan_threshold) {
                    p_sl[a][0] = min(p_sl[a][0], p_sl[b][0]);
                    p_sl[a][1] = max(p_sl[a][1], p_sl[b][1]);
                    p_cl[a][0] = min(p_cl[a][0], p_cl[b][0]);
                    p_cl[a][1] = max(p_cl[a][1], p_cl[b][1]);
                    mask[b] = false;
                }
            }
        }
        int m = 0;
        for (int a = 0; a < n_pk; ++a)
            if (mask[a]) {
                p_scan[m] = p_scan[a];
                p_cycle[m] = p_cycle[a];
                p_score[m] = p_score[a];
                p_sl[m][0] = p_sl[a][0];
                p_sl[m][1] = p_sl[a][1];
                p_cl[m][0] = p_cl[a][0];
                p_cl[m][1] = p_cl[a][1];
                ++m;
            }
        n_pk = m;
    }
    // ---- absolute coordinates (selection.py:480-526)
    const int L = run.cycle_len, z = run.zeroth;
    const int64_t scan_max = run.scan_max, frame_max = run.n_frames - 1, frame0 = (int64_t)r.cycle_start * L + z;
    auto wrap0 = [](int64_t v, int64_t limit) { return v < 0 ? (int64_t)0 : min(v, limit); };
    for (int q = 0; q < n_pk; ++q) {
        const int64_t row = (first_prec + i) * cfg.candidate_count + q;
        out.precursor_idx[row] = r.precursor_idx;
        out.rank[row] = (uint8_t)q;
        out.score[row] = (float)p_score[q];
        out.scan_center[row] = (uint32_t)wrap0((int64_t)p_scan[q] + r.scan_start, scan_max);
        out.scan_start[row] = (uint32_t)wrap0((int64_t)p_sl[q][0] + r.scan_start, scan_max);
        out.scan_stop[row] = (uint32_t)wrap0((int64_t)p_sl[q][1] + r.scan_start, scan_max);
        out.frame_center[row] = (uint32_t)wrap0((int64_t)p_cycle[q] * L + frame0, frame_max);
        out.frame_start[row] = (uint32_t)wrap0((int64_t)p_cl[q][0] * L + frame0, frame_max);
        out.frame_stop[row] = (uint32_t)wrap0((int64_t)p_cl[q][1] * L + frame0, frame_max);
    }
}
