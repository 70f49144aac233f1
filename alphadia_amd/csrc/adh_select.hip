// adh_select.hip - candidate selection on AlphaRaw runs (the step before scoring).
//
// One 64-lane wavefront per precursor.  Replaces, for every precursor,
//   _select_candidates_pjit / _build_candidates   alphadia/search/selection/selection.py:78-526
//   AlphaRawJIT.get_dense_intensity               alphadia/search/jitclasses/alpharaw_jit.py:339-425
//   get_frame_indices                             alphadia/search/jitclasses/utils.py:24-88
//   fft.convolve_fourier                          alphadia/search/selection/fft.py:119-212
//   find_peaks_1d / symetric_limits_2d            alphadia/search/selection/utils.py:49-77,218-312
//
// The XICs of all (cardinality-filtered) fragments and of the isotopes over rt +- tolerance
// come out of the same time-major transposed copy of the run that the scoring gather uses
// (adh_gather.hip): one lane per (m/z window, cycle row, cycle block) task.  The reference
// smooths every XIC with a 2-D FFT convolution; both scan slots of an AlphaRaw tile are equal,
// so the circular convolution collapses to one dimension and is evaluated directly (float64
// accumulation, rounded to float32 once - the reference's float32 FFT carries ~1e-3 of absolute
// noise).  Peak picking, joining and the symmetric limits are short scalar loops
// executed by lane 0 in the reference's order.
#include "adh_device.h"

namespace sel {

using gather::Window;

constexpr int WB = 8;         // m/z windows gathered per batch
constexpr int MAX_ROWS = 16;  // cycle rows overlapping one quadrupole range
constexpr int MAX_CAND = 16;

// intensity-only variant of gather::gather_task: cells[f] += intensity, f = cycle - c0
__device__ __forceinline__ void gather_sum_task(const DevRun &run, const Window &w, int row, int blk, int c0,
                                                int F, float *cells) {
    if (w.b_hi < w.b_lo) return;
    const int bs = run.block_shift;
    const int cyc_base = blk << bs;
    const int grp_base = (blk >> ADH_SUB_SHIFT) << (bs + ADH_SUB_SHIFT);
    const int f_lo = max(c0, cyc_base) - grp_base;
    const int f_hi = min(c0 + F, cyc_base + (1 << bs)) - grp_base;
    const uint32_t *t = adh_tab_row(run, row, blk) + (blk & (ADH_SUB - 1));
    const uint2 *ent = adh_group_entries(run, blk);
    for (int b = w.b_lo; b <= w.b_hi; ++b) {  // bin after bin: a cell keeps ascending m/z
        uint32_t idx = t[b * ADH_SUB];
        const uint32_t end = t[b * ADH_SUB + 1];
        while (idx < end) {
            uint2 e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) e[u] = ent[min(idx + (uint32_t)u, end - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = idx + (uint32_t)u;
                if (i >= end) break;
                const int cyc = (int)(e[u].x >> ADH_BIN_SHIFT);
                if (cyc < f_lo || cyc >= f_hi) continue;
                const float mz = __uint_as_float(((uint32_t)(run.bin0 + b) << ADH_BIN_SHIFT) |
                                                 (e[u].x & ((1u << ADH_BIN_SHIFT) - 1u)));
                if (!(mz >= w.lo && mz > w.excl) || !(mz <= w.hi)) continue;
                // alpharaw_jit.py:405-420: float32 running sum in ascending m/z
                float *c = cells + (cyc + grp_base - c0);
                *c = *c + __uint_as_float(e[u].y);
            }
            idx += 4;
        }
    }
}

// one output of the smoothing: sum over kernel rows a (outer) and columns b (inner) of
// kernel[a][b] * row[(f + k1/2 - b) mod F], float64 fused multiply-adds, in this order (the oracle's)
template <int K1>
__device__ __forceinline__ double conv_fixed(const float *rp, const double *kd, int k0) {
    double win[K1];  // the K1 row values this output touches, converted once
#pragma unroll
    for (int b = 0; b < K1; ++b) win[b] = (double)rp[-b];
    double acc = 0.0;
    for (int a = 0; a < k0; ++a) {
#pragma unroll
        for (int b = 0; b < K1; ++b) acc = fma(kd[a * K1 + b], win[b], acc);
    }
    return acc;
}

__device__ __forceinline__ double conv_any(const float *rp, const double *kd, int k0, int k1) {
    double acc = 0.0;
    for (int a = 0; a < k0; ++a)
        for (int b = 0; b < k1; ++b) acc = fma(kd[a * k1 + b], (double)rp[-b], acc);
    return acc;
}

// selection/utils.py:218-280
__device__ inline void symetric_limits_1d(const double *a, int n, int center, double f, double center_fraction,
                                          int64_t min_size, int64_t max_size, int out[2]) {
    if (n == 0 || center < 0 || center >= n) {
        out[0] = out[1] = center;
        return;
    }
    const double center_intensity = a[center];
    double trailing = center_intensity;
    int64_t limit = min_size;
    for (int64_t s = min_size + 1; s < max_size; ++s) {
        const int64_t il = max((int64_t)center - s, (int64_t)0), ir = min((int64_t)center + s, (int64_t)n - 1);
        const double intensity = (a[il] + a[ir]) / 2;
        if (intensity < f * trailing) {
            if (intensity > center_intensity * center_fraction) {
                limit = s;
                trailing = intensity;
            } else {
                break;
            }
        } else {
            break;
        }
    }
    out[0] = (int)max((int64_t)center - limit, (int64_t)0);
    out[1] = (int)min((int64_t)center + limit + 1, (int64_t)n);
}

struct SelCaps {
    int32_t n_lib;   // longest fragment slice
    int32_t f;       // largest cycle count of a tile
    int32_t n_iso;   // isotopes used
    int32_t k_rows, k_cols;
};

__host__ __device__ inline size_t lds_bytes(const SelCaps &c) {
    size_t b = 0;
    b += (size_t)c.f * 8;                          // score (double)
    b += (size_t)(c.n_lib + c.n_iso) * sizeof(Window);
    b = (b + 7) / 8 * 8;
    b += (size_t)((c.n_lib + 1) & ~1) * 4;         // raw fragment m/z (even count: keeps 8-byte alignment)
    b += (size_t)WB * c.f * 4;                     // tile of one batch (>= one float64 row)
    b += (size_t)c.f * 4 * 2;                      // lf, lp
    b += (size_t)WB * (c.f + 2 * c.k_cols) * 4;    // rows of one batch with wrap-around margins
    b = (b + 7) / 8 * 8;
    b += (size_t)c.k_rows * c.k_cols * 8;          // kernel (float64)
    return (b + 15) / 16 * 16;
}

}  // namespace sel

struct DevPrecursors {
    const uint32_t *precursor_idx, *frag_start, *frag_stop;
    const uint8_t *charge;
    const float *rt, *mz, *iso;
    const int32_t *cycle_start, *cycle_count;  // first cycle / number of cycles of every tile
    int32_t n_iso_cols;
};

struct DevCandTable {
    uint32_t *precursor_idx;
    uint8_t *rank;
    float *score;
    uint32_t *scan_center, *scan_start, *scan_stop, *frame_center, *frame_start, *frame_stop;
};

__global__ __launch_bounds__(ADH_WAVE) void adh_select_kernel(DevRun run, const LibRec *__restrict__ lib,
                                                             DevPrecursors pc, int64_t n_prec,
                                                             adh_selection_config_t cfg,
                                                             const float *__restrict__ kernel_g, sel::SelCaps caps,
                                                             DevCandTable out) {
    using namespace sel;
    extern __shared__ __align__(16) unsigned char smem[];
    double *score = reinterpret_cast<double *>(smem);
    Window *win = reinterpret_cast<Window *>(score + caps.f);
    unsigned char *p8 = smem + ((size_t)caps.f * 8 + (size_t)(caps.n_lib + caps.n_iso) * sizeof(Window) + 7) / 8 * 8;
    float *raw_mz = reinterpret_cast<float *>(p8);
    float *tile = raw_mz + ((caps.n_lib + 1) & ~1);
    float *lf = tile + (size_t)WB * caps.f;
    float *lp = lf + caps.f;
    float *rowbuf = lp + caps.f;
    const int row_stride = caps.f + 2 * caps.k_cols;
    double *kern = reinterpret_cast<double *>(
        smem + ((size_t)(reinterpret_cast<unsigned char *>(rowbuf + (size_t)WB * row_stride) - smem) + 7) / 8 * 8);
    __shared__ int s_rows[MAX_ROWS];
    __shared__ int s_misc[8];

    const int lane = threadIdx.x;
    const int64_t i = blockIdx.x;
    if (i >= n_prec) return;
    const int L = run.cycle_len;
    const int k0 = caps.k_rows, k1 = caps.k_cols;
    for (int c = lane; c < k0 * k1; c += ADH_WAVE) kern[c] = (double)kernel_g[c];

    // ---- isotopes (assemble_isotope_mz, selection/utils.py:24-46): float32 array += float64 offsets
    const int n_iso = caps.n_iso;
    const int W0 = caps.n_lib;  // isotope windows live behind the fragment windows
    if (lane < n_iso) {
        const double off = (double)lane * 1.0033548350700006 / (double)pc.charge[i];
        const float m = (float)((double)pc.mz[i] + off);
        float t = (float)cfg.precursor_mz_tolerance * m;  // mass_range, jitclasses/utils.py:15-20
        float q = t / 1000000.0f;
        win[W0 + lane].lo = m - q;
        win[W0 + lane].hi = m + q;
        if (lane == 0) s_misc[0] = __float_as_int(m);
        if (lane == n_iso - 1) s_misc[1] = __float_as_int(m);
    }
    // ---- fragments: slice, cardinality filter, sort by m/z (selection.py:124-139); no top-k
    const uint32_t fs = pc.frag_start[i], fe = pc.frag_stop[i];
    const int n_lib = (int)(fe - fs);
    __syncthreads();
    for (int j = lane; j < n_lib; j += ADH_WAVE) {
        const LibRec r = lib[fs + j];
        raw_mz[j] = (cfg.exclude_shared_ions && r.cardinality > 1) ? -1.0f : r.mz;  // m/z > 0 always
    }
    __syncthreads();
    int K = 0;
    for (int j = 0; j < n_lib; ++j) K += raw_mz[j] >= 0.0f;
    if (K <= 3) return;  // selection.py:141
    for (int a = lane; a < n_lib; a += ADH_WAVE) {
        const float ma = raw_mz[a];
        if (ma < 0.0f) continue;
        int slot = 0;
        for (int b = 0; b < n_lib; ++b) {
            const float mb = raw_mz[b];
            if (mb < 0.0f) continue;
            slot += (mb < ma) || (mb == ma && b < a);
        }
        float t = (float)cfg.fragment_mz_tolerance * ma;
        float q = t / 1000000.0f;
        win[slot].lo = ma - q;
        win[slot].hi = ma + q;
    }
    __syncthreads();
    if (lane == 0) {
        float e = -INFINITY;
        for (int k = 0; k < K; ++k) {
            win[k].excl = e;
            e = fmaxf(e, win[k].hi);
        }
        e = -INFINITY;
        for (int k = 0; k < n_iso; ++k) {
            win[W0 + k].excl = e;
            e = fmaxf(e, win[W0 + k].hi);
        }
        // frame limits (get_frame_indices, jitclasses/utils.py:24-88): searched on the host, which
        // needs them anyway to size the tiles (adh_select_candidates)
        s_misc[2] = pc.cycle_start[i];
        s_misc[3] = pc.cycle_count[i];
    }
    __syncthreads();
    for (int w = lane; w < K + n_iso; w += ADH_WAVE) gather::bins_of(run, win[w < K ? w : W0 + (w - K)]);
    const int cs = s_misc[2], F = s_misc[3];
    // _is_valid (selection.py:40-75) with two scan slots
    if (F <= 0 || F > caps.f || n_iso == 0 || 2 < k0 || F < k1) return;
    const double q_lo = (double)__int_as_float(s_misc[0]), q_hi = (double)__int_as_float(s_misc[1]);
    const int bs = run.block_shift;
    const int blk0 = cs >> bs, n_blk = ((cs + F - 1) >> bs) - blk0 + 1;
    for (int f = lane; f < F; f += ADH_WAVE) {
        lf[f] = 0.0f;
        lp[f] = 0.0f;
    }

    // ---- two groups of windows: fragments (rows overlapping the isotope range) and isotopes (MS1 rows)
    for (int group = 0; group < 2; ++group) {
        // cycle rows of this group, ascending (_calculate_valid_scans, alpharaw_jit.py:19-50)
        __syncthreads();
        if (lane == 0) {
            int n = 0;
            for (int row = 0; row < L; ++row) {
                const double lo = run.cycle[2 * row], hi = run.cycle[2 * row + 1];
                const bool ok = group == 0 ? (q_lo <= hi && q_hi >= lo) : (-1.0 <= hi && -1.0 >= lo);
                if (ok && n < MAX_ROWS) s_rows[n++] = row;
            }
            s_misc[4] = n;
        }
        __syncthreads();
        const int n_rows = s_misc[4];
        const int Wn = group == 0 ? K : n_iso;
        float *lsum = group == 0 ? lf : lp;
        for (int w0 = 0; w0 < Wn; w0 += WB) {
            const int wb = min(WB, Wn - w0);
            for (int c = lane; c < wb * F; c += ADH_WAVE) tile[c] = 0.0f;
            __syncthreads();
            // one lane per (window, cycle block); the cycle rows are visited in ascending order so
            // that a cell keeps the reference's running float32 sum (alpharaw_jit.py:392-420)
            const int n_tasks = wb * n_blk;
            for (int t = lane; t < n_tasks; t += ADH_WAVE) {
                const int bi = t % n_blk, w = t / n_blk;
                const Window &ww = win[group == 0 ? w0 + w : W0 + w0 + w];
                for (int r = 0; r < n_rows; ++r)
                    gather_sum_task(run, ww, s_rows[r], blk0 + bi, cs, F, tile + w * F);
            }
            __syncthreads();
            // the rows with k1 wrap-around cells on either side: rowbuf[w][k1 + j] = row_w[j mod F]
            for (int c = lane; c < wb * (F + 2 * k1); c += ADH_WAVE) {
                const int w = c / (F + 2 * k1), j = c - w * (F + 2 * k1);
                int src = (j - k1) % F;
                if (src < 0) src += F;
                rowbuf[w * row_stride + j] = tile[w * F + src];
            }
            __syncthreads();
            // circular convolution, kernel centred at column k1 / 2 (selection/fft.py:163-212), for
            // all (window, cycle) outputs of the batch; log(smooth + 1) replaces the tile value
            for (int o = lane; o < wb * F; o += ADH_WAVE) {
                const int w = o / F, f = o - w * F;
                const float *rp = rowbuf + w * row_stride + k1 + f + k1 / 2;  // rp[-b] = row[(f + k1/2 - b) mod F]
                const double acc = (k1 == 30) ? conv_fixed<30>(rp, kern, k0) : conv_any(rp, kern, k0, k1);
                const float sm = (float)acc;
                const float x1 = sm + 1.0f;  // _build_features (selection.py:206-226)
                tile[o] = (float)log((double)x1);
            }
            __syncthreads();
            for (int f = lane; f < F; f += ADH_WAVE) {
                float a = lsum[f];
                for (int w = 0; w < wb; ++w) a += tile[w * F + f];  // np.sum over the windows, in order
                lsum[f] = a;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- score (selection.py:396-421), single feature
    {
        double mean = cfg.feature_mean, sd = cfg.feature_std, weight = cfg.feature_weight;
        if (!cfg.use_weighted_score) {
            // amean1 / astd1 over the (2, F) feature matrix; lane 0, sequential
            if (lane == 0) {
                double m = 0;
                for (int f = 0; f < F; ++f) m += 2.0 * (double)(lf[f] + lp[f]);
                m /= (2.0 * F);
                double v = 0;
                for (int f = 0; f < F; ++f) {
                    const double d = (double)(lf[f] + lp[f]) - m;
                    v += 2.0 * d * d;
                }
                score[0] = m;
                score[1] = sqrt(v / (2.0 * F));
            }
            __syncthreads();
            mean = score[0];
            sd = score[1];
            weight = 1.0;
            __syncthreads();
        }
        for (int f = lane; f < F; f += ADH_WAVE) {
            const float ft = lf[f] + lp[f];
            score[f] = weight * ((double)ft - mean) / (sd + 1e-6);
        }
    }
    __syncthreads();
    if (lane != 0) return;

    // ---- peaks, joins, limits: short scalar loops in the reference's order
    int p_cycle[MAX_CAND];
    double p_score[MAX_CAND];
    int p_sl[MAX_CAND][2], p_cl[MAX_CAND][2];
    int n_pk = 0;
    const int top_n = (int)min((int64_t)MAX_CAND, cfg.candidate_count);
    // find_peaks_1d (selection/utils.py:49-77): keep the top_n by score, descending; equal scores in
    // reversed index order (argsort()[::-1])
    for (int p = 2; p < F - 2; ++p) {
        if (!(score[p - 2] < score[p - 1] && score[p - 1] < score[p] && score[p] > score[p + 1] &&
              score[p + 1] > score[p + 2]))
            continue;
        const double s = score[p];
        int pos = n_pk;
        while (pos > 0 && (p_score[pos - 1] < s || (p_score[pos - 1] == s && p_cycle[pos - 1] < p))) --pos;
        if (pos >= top_n) continue;
        const int last = min(n_pk, top_n - 1);
        for (int j = last; j > pos; --j) {
            p_score[j] = p_score[j - 1];
            p_cycle[j] = p_cycle[j - 1];
        }
        p_score[pos] = s;
        p_cycle[pos] = p;
        if (n_pk < top_n) ++n_pk;
    }
    // _join_close_peaks (selection.py:229-278), tolerances 3 / 3; all scans are 0 here
    {
        bool mask[MAX_CAND];
        for (int a = 0; a < n_pk; ++a) mask[a] = true;
        for (int a = 0; a < n_pk; ++a) {
            if (!mask[a]) continue;
            for (int b = a + 1; b < n_pk; ++b) {
                if (!mask[b]) continue;
                if (abs(p_cycle[a] - p_cycle[b]) <= 3) {
                    if (p_score[a] > p_score[b]) mask[b] = false; else mask[a] = false;
                }
            }
        }
        int m = 0;
        for (int a = 0; a < n_pk; ++a)
            if (mask[a]) {
                p_cycle[m] = p_cycle[a];
                p_score[m] = p_score[a];
                ++m;
            }
        n_pk = m;
    }
    // symetric_limits_2d (selection/utils.py:283-312) on the (2, F) matrix with equal rows;
    // the cycle profile goes to the (now free) fragment log-sum row as float64 pairs
    double *cyc = reinterpret_cast<double *>(tile);
    for (int a = 0; a < n_pk; ++a) {
        const int cen = p_cycle[a];
        const int mob_lower = 0, mob_upper = (int)min((int64_t)2, (int64_t)0 + cfg.min_size_mobility);
        const int cyc_lower = (int)max((int64_t)0, (int64_t)cen - cfg.min_size_rt);
        const int cyc_upper = (int)min((int64_t)F, (int64_t)cen + cfg.min_size_rt);
        double mob[2] = {0.0, 0.0};
        for (int s = 0; s < 2; ++s)
            for (int f = cyc_lower; f < cyc_upper; ++f) mob[s] += score[f];
        for (int f = 0; f < F; ++f) {
            double v = 0.0;
            for (int s = mob_lower; s < mob_upper; ++s) v += score[f];
            cyc[f] = v;
        }
        symetric_limits_1d(mob, 2, 0, cfg.f_mobility, cfg.center_fraction, cfg.min_size_mobility,
                           cfg.max_size_mobility, p_sl[a]);
        symetric_limits_1d(cyc, F, cen, cfg.f_rt, cfg.center_fraction, cfg.min_size_rt, cfg.max_size_rt, p_cl[a]);
    }
    // _join_overlapping_candidates (selection.py:281-345)
    if (cfg.join_close_candidates) {
        bool mask[MAX_CAND];
        for (int a = 0; a < n_pk; ++a) mask[a] = true;
        for (int a = 0; a < n_pk; ++a) {
            if (!mask[a]) continue;
            for (int b = a + 1; b < n_pk; ++b) {
                if (!mask[b]) continue;
                const double cyc_len = (double)(p_cl[a][1] - p_cl[a][0]);
                const double cyc_ov = (double)(min(p_cl[a][1], p_cl[b][1]) - max(p_cl[a][0], p_cl[b][0])) / cyc_len;
                const double scan_len = (double)(p_sl[a][1] - p_sl[a][0]);
                const double scan_ov = (double)(min(p_sl[a][1], p_sl[b][1]) - max(p_sl[a][0], p_sl[b][0])) / scan_len;
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
    // ---- absolute coordinates (selection.py:480-526): scan_max_index = 1, frame_max_index = n_spectra - 1
    const int64_t scan_max = 1, frame_max = run.n_spectra - 1, frame0 = (int64_t)cs * L;
    for (int r = 0; r < n_pk; ++r) {
        const int64_t row = i * cfg.candidate_count + r;
        auto wrap0 = [](int64_t v, int64_t limit) { return v < 0 ? (int64_t)0 : min(v, limit); };
        out.precursor_idx[row] = pc.precursor_idx[i];
        out.rank[row] = (uint8_t)r;
        out.score[row] = (float)p_score[r];
        out.scan_center[row] = (uint32_t)wrap0(0, scan_max);
        out.scan_start[row] = (uint32_t)wrap0(p_sl[r][0], scan_max);
        out.scan_stop[row] = (uint32_t)wrap0(p_sl[r][1], scan_max);
        out.frame_center[row] = (uint32_t)wrap0((int64_t)p_cycle[r] * L + frame0, frame_max);
        out.frame_start[row] = (uint32_t)wrap0((int64_t)p_cl[r][0] * L + frame0, frame_max);
        out.frame_stop[row] = (uint32_t)wrap0((int64_t)p_cl[r][1] * L + frame0, frame_max);
    }
}
