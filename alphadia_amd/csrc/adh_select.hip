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
// noise).  Peaks are flagged by all lanes; joining and the symmetric limits are short scalar
// loops executed by lane 0 in the reference's order.
//
// Round 4: fragments and isotopes share a batch of 16 windows (one gather phase per precursor instead of three),
// the cycle rows of both groups are found by all lanes (the one-lane loop over the cycle table with its dependent
// loads was most of the kernel's time), the smoothing has its taps in scalar registers (kernel argument) and a
// lane computes five neighbouring outputs from 34 row values in registers (no margin copy, no modulo), the log
// is the table-driven one of the ion-mobility selection.  Every sum keeps its order: bit-identical candidates.
#include "adh_device.h"
#include "adh_log_f32.h"

namespace sel {

using gather::Window;

constexpr int WB = 16;        // m/z windows (fragments, then isotopes) per batch
constexpr int MAX_ROWS = 16;  // cycle rows overlapping one quadrupole range
constexpr int MAX_CAND = 16;
constexpr int CONV_R = 10;    // neighbouring outputs of one lane in the smoothing (80 cycles x 15 windows: two full rounds)
// The smoothing kernel as a kernel argument (scalar operands): its columns [col0, col0 + NB) - the others are zero
// in every row, and a zero tap leaves a running fma sum as it is (the reference's Gaussian over 30 cycles has 15
// non-zero columns) - as float64, NB padded with zeros to 16 columns, two rows.
constexpr int TAP_COLS = 16;
struct Taps {
    double v[2 * TAP_COLS];
    int32_t col0, cols;  // cols: TAP_COLS, or 0 = the kernel does not fit (more non-zero columns, another row count: taps from LDS)
};

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

// R neighbouring outputs f0 ... f0 + R - 1 of the smoothing of one row (circular, kernel centred at column k1 / 2,
// selection/fft.py:163-212): out[f] = sum over kernel rows a (outer) and columns b (inner) of
// kernel[a][b] * row[(f + k1/2 - b) mod F], float64 fused multiply-adds in this order (the oracle's), over the
// NB columns from col0 on (Taps).  The NB + R - 1 row values the outputs touch are read and converted once.
template <int K0, int NB, int R>
__device__ __forceinline__ void conv_chunk(const float *row, int F, int f0, int k1, const Taps &taps, double (&acc)[R]) {
    double win[NB + R - 1];  // win[j] = row[(f0 + k1/2 - col0 - (NB - 1) + j) mod F]
    int idx = (f0 + k1 / 2 - taps.col0 - (NB - 1)) % F;  // (|first term| < F + 64: one division, then steps)
    idx += idx < 0 ? F : 0;
#pragma unroll
    for (int j = 0; j < NB + R - 1; ++j) {
        win[j] = (double)row[idx];
        ++idx;
        idx -= idx >= F ? F : 0;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
#pragma unroll
    for (int a = 0; a < K0; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = fma(taps.v[a * NB + b], win[r + NB - 1 - b], acc[r]);
}

// one output, any kernel shape (taps in LDS)
__device__ __forceinline__ double conv_any(const float *row, int F, int f, const double *kd, int k0, int k1) {
    double acc = 0.0;
    int start = (f + k1 / 2) % F;
    for (int a = 0; a < k0; ++a) {
        int idx = start;
        for (int b = 0; b < k1; ++b) {
            acc = fma(kd[a * k1 + b], (double)row[idx], acc);
            idx = idx == 0 ? F - 1 : idx - 1;
        }
    }
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
    int32_t stop;    // developer ablation (ADH_DEBUG_SELECT_STOP): 1 windows, 2 + cycle rows, 3 + gather, 4 + smoothing, 5 + score
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
    b += (size_t)WB * c.f * 4;                     // log(smooth + 1) of one batch
    b = (b + 7) / 8 * 8;
    b += (size_t)c.k_rows * c.k_cols * 8;          // kernel (float64; shapes that do not travel as an argument)
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

// (arguments as ONE struct, read through the kernel-argument segment where they are needed - adh_fused.hip, FusedArgs:
// as formal parameters they are loaded in the prologue, spilled to vector-register lanes and read back by v_readlane)
struct SelectArgs {
    DevRun run;
    const LibRec *lib;
    DevPrecursors pc;
    int64_t n_prec;
    adh_selection_config_t cfg;
    const float *kernel_g;
    sel::Taps taps;
    sel::SelCaps caps;
    DevCandTable out;
};
__global__ __launch_bounds__(ADH_WAVE) void adh_select_kernel(SelectArgs formal_args_not_read) {
    const SelectArgs &KA = *(const SelectArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const DevRun &run = KA.run;
    const LibRec *__restrict__ lib = KA.lib;
    const DevPrecursors &pc = KA.pc;
    const int64_t n_prec = KA.n_prec;
    const adh_selection_config_t &cfg = KA.cfg;
    const float *__restrict__ kernel_g = KA.kernel_g;
    const sel::Taps &taps = KA.taps;
    const sel::SelCaps &caps = KA.caps;
    const DevCandTable &out = KA.out;
    using namespace sel;
    extern __shared__ __align__(16) unsigned char smem[];
    double *score = reinterpret_cast<double *>(smem);
    Window *win = reinterpret_cast<Window *>(score + caps.f);
    unsigned char *p8 = smem + ((size_t)caps.f * 8 + (size_t)(caps.n_lib + caps.n_iso) * sizeof(Window) + 7) / 8 * 8;
    float *raw_mz = reinterpret_cast<float *>(p8);
    float *tile = raw_mz + ((caps.n_lib + 1) & ~1);
    float *lf = tile + (size_t)WB * caps.f;
    float *lp = lf + caps.f;
    float *logt = lp + caps.f;
    double *kern = reinterpret_cast<double *>(
        smem + ((size_t)(reinterpret_cast<unsigned char *>(logt + (size_t)WB * caps.f) - smem) + 7) / 8 * 8);
    __shared__ int s_rows[2][MAX_ROWS];
    __shared__ int s_misc[8];

    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int64_t i = blockIdx.x;
    if (i >= n_prec) return;
    const int L = run.cycle_len;
    const int k0 = caps.k_rows, k1 = caps.k_cols;
    const bool fixed_taps = k0 == 2 && taps.cols != 0;  // (taps in scalar registers)
    if (!fixed_taps)
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
    int K = 0;
    for (int j0 = 0; j0 < n_lib; j0 += ADH_WAVE) {
        const int j = j0 + lane;
        bool keep = false;
        if (j < n_lib) {
            const LibRec r = lib[fs + j];
            keep = !(cfg.exclude_shared_ions && r.cardinality > 1);
            raw_mz[j] = keep ? r.mz : -1.0f;  // m/z > 0 always
        }
        K += __popcll(__ballot(keep));
    }
    __syncthreads();
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
    if (lane < 2) {  // the exclusive lower bounds of a group's windows (lane 0: fragments, lane 1: isotopes)
        const int base = lane == 0 ? 0 : W0, cnt = lane == 0 ? K : n_iso;
        float e = -INFINITY;
        for (int k = 0; k < cnt; ++k) {
            win[base + k].excl = e;
            e = fmaxf(e, win[base + k].hi);
        }
    }
    __syncthreads();
    for (int w = lane; w < K + n_iso; w += ADH_WAVE) gather::bins_of(run, win[w < K ? w : W0 + (w - K)]);
    // frame limits (get_frame_indices, jitclasses/utils.py:24-88): searched on the host, which needs them anyway
    // to size the tiles (adh_select_candidates)
    if (caps.stop == 1) return;
    const int cs = pc.cycle_start[i], F = pc.cycle_count[i];
    // _is_valid (selection.py:40-75) with two scan slots
    if (F <= 0 || F > caps.f || n_iso == 0 || 2 < k0 || F < k1) return;
    const double q_lo = (double)__int_as_float(s_misc[0]), q_hi = (double)__int_as_float(s_misc[1]);
    const int bs = run.block_shift;
    const int blk0 = cs >> bs, n_blk = ((cs + F - 1) >> bs) - blk0 + 1;
    for (int f = lane; f < F; f += ADH_WAVE) {
        lf[f] = 0.0f;
        lp[f] = 0.0f;
    }
    // ---- cycle rows of the two groups of windows, ascending (_calculate_valid_scans, alpharaw_jit.py:19-50):
    // fragments (rows overlapping the isotope range) and isotopes (MS1 rows); a lane per row
    int n_rows[2] = {0, 0};
    for (int row0 = 0; row0 < L; row0 += ADH_WAVE) {
        const int row = row0 + lane;
        double lo = 0.0, hi = -2.0;
        if (row < L) {
            lo = run.cycle[2 * row];
            hi = run.cycle[2 * row + 1];
        }
#pragma unroll
        for (int group = 0; group < 2; ++group) {
            const bool ok = row < L && (group == 0 ? (q_lo <= hi && q_hi >= lo) : (-1.0 <= hi && -1.0 >= lo));
            const unsigned long long mask = __ballot(ok);
            const int pos = n_rows[group] + __popcll(mask & lt);
            if (ok && pos < MAX_ROWS) s_rows[group][pos] = row;
            n_rows[group] += __popcll(mask);
        }
    }
    n_rows[0] = min(n_rows[0], MAX_ROWS);
    n_rows[1] = min(n_rows[1], MAX_ROWS);

    if (caps.stop == 2) return;

    // ---- batches of windows: fragments, then isotopes
    const int Wt = K + n_iso;
    const int n_chunk = (F + CONV_R - 1) / CONV_R;
    for (int w0 = 0; w0 < Wt; w0 += WB) {
        const int wb = min(WB, Wt - w0);
        __syncthreads();
        for (int c = lane; c < wb * F; c += ADH_WAVE) tile[c] = 0.0f;
        __syncthreads();
        // one lane per (window, cycle block); the cycle rows are visited in ascending order so
        // that a cell keeps the reference's running float32 sum (alpharaw_jit.py:392-420)
        const int n_tasks = wb * n_blk;
        for (int t = lane; t < n_tasks; t += ADH_WAVE) {
            const int w = t / n_blk, bi = t - w * n_blk;
            const int gw = w0 + w, group = gw >= K;
            const Window &ww = win[group ? W0 + (gw - K) : gw];
            for (int r = 0; r < n_rows[group]; ++r)
                gather_sum_task(run, ww, s_rows[group][r], blk0 + bi, cs, F, tile + w * F);
        }
        __syncthreads();
        if (caps.stop == 3) continue;
        // circular convolution for all (window, cycle) outputs of the batch, then log(smooth + 1)
        // (_build_features, selection.py:206-226)
        auto put = [&](int w, int f, double acc) {
            const float sm = (float)acc;
            const float x1 = sm + 1.0f;
            float lg = 0.0f;  // (log(1) = 0: smoothed values below 6e-8 round away in sm + 1)
            if (x1 != 1.0f) lg = (x1 >= 1.0f && x1 < INFINITY) ? (float)adh_log_f32(x1) : adh_log_f32_rare(x1);
            logt[w * F + f] = lg;
        };
        if (fixed_taps) {
            for (int c = lane; c < wb * n_chunk; c += ADH_WAVE) {
                const int w = c / n_chunk, f0 = (c - w * n_chunk) * CONV_R;
                double acc[CONV_R];
                conv_chunk<2, TAP_COLS, CONV_R>(tile + w * F, F, f0, k1, taps, acc);
#pragma unroll
                for (int r = 0; r < CONV_R; ++r)
                    if (f0 + r < F) put(w, f0 + r, acc[r]);
            }
        } else {
            for (int o = lane; o < wb * F; o += ADH_WAVE) {
                const int w = o / F, f = o - w * F;
                put(w, f, conv_any(tile + w * F, F, f, kern, k0, k1));
            }
        }
        __syncthreads();
        // np.sum over the windows of a group, in order
        const int n_frag = max(0, min(K, w0 + wb) - w0);  // fragment windows of this batch come first
        for (int f = lane; f < F; f += ADH_WAVE) {
            float a = lf[f];
            for (int w = 0; w < n_frag; ++w) a += logt[w * F + f];
            lf[f] = a;
            a = lp[f];
            for (int w = n_frag; w < wb; ++w) a += logt[w * F + f];
            lp[f] = a;
        }
    }
    __syncthreads();
    if (caps.stop == 3 || caps.stop == 4) return;
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
    // the cycle profile of symetric_limits_2d (selection/utils.py:283-312) on the (2, F) matrix with equal rows
    // does not depend on the peak: score[f] added once per scan slot in reach; it goes to the (now free) tile
    double *cyc = reinterpret_cast<double *>(tile);
    const int mob_lower = 0, mob_upper = (int)min((int64_t)2, (int64_t)0 + cfg.min_size_mobility);
    for (int f = lane; f < F; f += ADH_WAVE) {
        double v = 0.0;
        for (int sl = mob_lower; sl < mob_upper; ++sl) v += score[f];
        cyc[f] = v;
    }

    if (caps.stop == 5) return;
    // ---- peaks, joins, limits: the peaks are flagged by all lanes, the rest are short scalar loops in the
    // reference's order
    // (the candidate lists live in LDS: indexed at run time, as registers they cost 110 VGPRs and a select chain
    // per access)
    __shared__ int p_cycle[MAX_CAND];
    __shared__ double p_score[MAX_CAND];
    __shared__ int p_sl[MAX_CAND][2], p_cl[MAX_CAND][2];
    int n_pk = 0;
    const int top_n = (int)min((int64_t)MAX_CAND, cfg.candidate_count);
    // find_peaks_1d (selection/utils.py:49-77): keep the top_n by score, descending; equal scores in
    // reversed index order (argsort()[::-1])
    for (int pb = 0; pb < F; pb += ADH_WAVE) {
        const int p = pb + lane;
        bool pk = false;
        if (p >= 2 && p < F - 2)
            pk = score[p - 2] < score[p - 1] && score[p - 1] < score[p] && score[p] > score[p + 1] &&
                 score[p + 1] > score[p + 2];
        const unsigned long long flags = __ballot(pk);
        unsigned long long mask = lane == 0 ? flags : 0ull;  // (lane 0 walks the flags, in ascending order)
        while (mask) {
            const int pp = pb + __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const double sc = score[pp];
            int pos = n_pk;
            while (pos > 0 && (p_score[pos - 1] < sc || (p_score[pos - 1] == sc && p_cycle[pos - 1] < pp))) --pos;
            if (pos >= top_n) continue;
            const int last = min(n_pk, top_n - 1);
            for (int j = last; j > pos; --j) {
                p_score[j] = p_score[j - 1];
                p_cycle[j] = p_cycle[j - 1];
            }
            p_score[pos] = sc;
            p_cycle[pos] = pp;
            if (n_pk < top_n) ++n_pk;
        }
    }
    __syncthreads();  // (cyc is complete)
    if (lane != 0) return;
    // _join_close_peaks (selection.py:229-278), tolerances 3 / 3; all scans are 0 here
    {
        uint32_t mask = 0xFFFFFFFFu;  // (bit a: candidate a is kept)
        for (int a = 0; a < n_pk; ++a) {
            if (!(mask >> a & 1u)) continue;
            for (int b = a + 1; b < n_pk; ++b) {
                if (!(mask >> b & 1u)) continue;
                if (abs(p_cycle[a] - p_cycle[b]) <= 3) {
                    if (p_score[a] > p_score[b]) mask &= ~(1u << b); else mask &= ~(1u << a);
                }
            }
        }
        int m = 0;
        for (int a = 0; a < n_pk; ++a)
            if (mask >> a & 1u) {
                p_cycle[m] = p_cycle[a];
                p_score[m] = p_score[a];
                ++m;
            }
        n_pk = m;
    }
    // symetric_limits_2d (selection/utils.py:283-312) on the (2, F) matrix with equal rows (cyc: above)
    for (int a = 0; a < n_pk; ++a) {
        const int cen = p_cycle[a];
        const int cyc_lower = (int)max((int64_t)0, (int64_t)cen - cfg.min_size_rt);
        const int cyc_upper = (int)min((int64_t)F, (int64_t)cen + cfg.min_size_rt);
        double mob[2] = {0.0, 0.0};
        for (int s = 0; s < 2; ++s)
            for (int f = cyc_lower; f < cyc_upper; ++f) mob[s] += score[f];
        symetric_limits_1d(mob, 2, 0, cfg.f_mobility, cfg.center_fraction, cfg.min_size_mobility,
                           cfg.max_size_mobility, p_sl[a]);
        symetric_limits_1d(cyc, F, cen, cfg.f_rt, cfg.center_fraction, cfg.min_size_rt, cfg.max_size_rt, p_cl[a]);
    }
    // _join_overlapping_candidates (selection.py:281-345)
    if (cfg.join_close_candidates) {
        uint32_t mask = 0xFFFFFFFFu;
        for (int a = 0; a < n_pk; ++a) {
            if (!(mask >> a & 1u)) continue;
            for (int b = a + 1; b < n_pk; ++b) {
                if (!(mask >> b & 1u)) continue;
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
                    mask &= ~(1u << b);
                }
            }
        }
        int m = 0;
        for (int a = 0; a < n_pk; ++a)
            if (mask >> a & 1u) {
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

// Tile limits of every precursor (get_frame_indices_tolerance -> get_frame_indices, alpharaw_jit.py:172-203,
// jitclasses/utils.py:24-88: the frames of rt +- tolerance by two searches over the retention times, as cycles,
// at least kernel_size of them, padded to a multiple of 16 and moved inside the run) and the checks of the
// precursor columns.  red[0] = longest fragment slice, red[1] = largest cycle count, red[2] = error bits
// (1: fragment slice outside the library, 2: charge 0).
__global__ void adh_select_limits_kernel(const float *__restrict__ rt, int64_t n_spectra, int L, DevPrecursors pc, int64_t n,
                                         int64_t n_lib, double rt_tolerance, int64_t kernel_size,
                                         int32_t *__restrict__ cyc_start, int32_t *__restrict__ cyc_count,
                                         int32_t *__restrict__ red) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int32_t slice = 0, cnt = 0, err = 0;
    if (i < n) {
        const uint32_t fs = pc.frag_start[i], fe = pc.frag_stop[i];
        if (fe < fs || (int64_t)fe > n_lib) err |= 1;
        else slice = (int32_t)(fe - fs);
        if (pc.charge[i] == 0) err |= 2;
        const float lo = (float)((double)pc.rt[i] - rt_tolerance), hi = (float)((double)pc.rt[i] + rt_tolerance);
        auto lower_bound = [&](float x) {
            int64_t a = 0, b = n_spectra;
            while (a < b) {
                const int64_t m = (a + b) >> 1;
                if (rt[m] < x) a = m + 1; else b = m;
            }
            return a;
        };
        const int64_t cmax = n_spectra / L;
        const int64_t c_lo = lower_bound(lo) / L, c_hi = lower_bound(hi) / L;
        int64_t len = max(c_hi - c_lo, kernel_size);
        len = (len + 15) / 16 * 16;
        int64_t cs = c_lo, ce = c_lo + len;
        if (ce > cmax) {
            ce = cmax;
            cs = cmax - len;
            if (cs < 0) cs = (cmax % 2 == 0) ? 0 : 1;
        }
        cyc_start[i] = (int32_t)cs;
        cyc_count[i] = cnt = (int32_t)(ce - cs);
    }
    for (int off = 32; off > 0; off >>= 1) {
        slice = max(slice, __shfl_xor(slice, off));
        cnt = max(cnt, __shfl_xor(cnt, off));
        err |= __shfl_xor(err, off);
    }
    if ((threadIdx.x & (ADH_WAVE - 1)) == 0) {
        atomicMax(&red[0], slice);
        atomicMax(&red[1], cnt);
        if (err) atomicOr(&red[2], err);
    }
}
