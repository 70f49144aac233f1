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
#include "adh_log_f32.h"

#ifndef ADH_SEL_TAP_FIT
#define ADH_SEL_TAP_FIT 1  // batches of the smoothing kernel are cut so that their tap lists fit (0: by the row storage only)
#endif
#ifndef ADH_SEL_LOG_LDS
#define ADH_SEL_LOG_LDS 0  // 1: the log table of the smoothing kernel in LDS - measured 4 % slower: its 2 KB come out of the tap lists
#endif

namespace selim {

constexpr int MAX_W = 64;    // m/z windows (fragments + isotopes) per precursor
constexpr int MAX_CAND = 16;
constexpr int SCORE_THREADS = 512;
constexpr int SMOOTH_THREADS = 1024;  // threads of the smoothing kernel (16 wavefronts)
constexpr int SEL_BATCH = 4;  // windows per batch of the smoothing kernel (<= SMOOTH_THREADS / 64)
constexpr int PAR_PEAKS = 4;  // peaks whose symmetric limits the score kernel works out side by side
constexpr int SEL_HEADER = 32 + 4 * (MAX_W + 2);  // bytes in front of the tiles of a precursor (see below)
constexpr uint32_t SEL_DENSE = 0u, SEL_COMPACT = 1u;
struct SelEntry {
    uint32_t cell;  // (window * S + scan) * F + cycle
    float x;        // summed intensity
};

// per-precursor record prepared by the host (frame / scan limits as the reference computes them)
struct __attribute__((aligned(16))) PrecRec {
    uint32_t precursor_idx, frag_start, frag_stop;
    float mz;
    int32_t cycle_start, n_cycles;   // first cycle, F
    int32_t scan_start, n_scans;     // first scan, S
    uint64_t scratch_off;            // bytes: SEL_HEADER + the tiles (W * S * F * 4) + one parked tile (S * F * 4)
    uint8_t charge, ok, pad[6];
};
static_assert(sizeof(PrecRec) == 48, "PrecRec must be 48 bytes");

}  // namespace selim

// ------------------------------------------------------------------------------------------------
// Scratch block of one precursor: header (SEL_HEADER bytes: [0] K, [1] W, [2] mode, [3] entry count,
// then the first entry of every window, W + 1 values), the tiles - in sparse form (SEL_COMPACT: the
// non-zero cells of all windows as (cell, intensity) entries sorted by cell,
// cell = (window * S + scan) * F + cycle) or, when there are too many for that, as float[W][S][F]
// (SEL_DENSE) - and one float[S][F] tile where the score kernel parks its sums.  The tiles of a
// precursor are ~0.1 % full (176 scans x 48 cycles x 15 windows, a handful of events per window).
__global__ __launch_bounds__(ADH_WAVE) void adh_select_gather_im_kernel(
    DevTims run, const LibRec *__restrict__ lib, const selim::PrecRec *__restrict__ recs, int32_t n_prec,
    adh_selection_config_t cfg, int32_t n_iso, unsigned char *__restrict__ scratch, int32_t debug_dense) {
    using namespace selim;
    __shared__ float s_mz[MAX_W];       // window centres: fragments ascending, then isotopes
    __shared__ int s_tlo[MAX_W], s_thi[MAX_W];
    __shared__ float s_raw[MAX_W];
    __shared__ int w_p0[MAX_W + 1];                   // first (window, TOF bin) pair of every window
    __shared__ int64_t w_base[MAX_W];                 // first event of every window's first TOF bin
    __shared__ uint32_t p_lo[ADH_IM_PAIR_CAP];        // first event of the pair's range, counted from w_base
    __shared__ uint32_t p_off[ADH_IM_PAIR_CAP + 2];   // events before the pair
    __shared__ uint8_t p_win[ADH_IM_PAIR_CAP];
    __shared__ uint32_t s_key[ADH_IM_SORT_CAP];       // cell << 9 | position in the list
    __shared__ uint16_t s_int[ADH_IM_SORT_CAP];
    __shared__ uint8_t s_pair[ADH_IM_SORT_CAP];
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
    const int K = __popcll(__ballot(lane < n_lib && s_raw[lane] >= 0.0f));  // (n_lib <= MAX_W <= 64; this lane's own write)
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
    if (W <= ADH_WAVE / 2) {
        // two lanes per window, one per end (as adh_gather_im_kernel: the searches are dependent look-ups of a wavefront
        // with nothing else to do)
        const int w = lane & (ADH_WAVE / 2 - 1);
        const bool upper = lane >= ADH_WAVE / 2, act = w < W;
        int bound = 0;
        if (act) {
            const float m = s_mz[w];
            const float tol = (float)(w < K ? cfg.fragment_mz_tolerance : cfg.precursor_mz_tolerance);
            float t = tol * m;
            float q = t / 1000000.0f;
            bound = index_im::tof_lower_bound(run, (double)(upper ? m + q : m - q));
        }
        const int other = __shfl(bound, lane ^ (ADH_WAVE / 2));  // the lower end, seen from the upper lane
        if (act && !upper) s_tlo[w] = bound;
        if (act && upper) s_thi[w] = bound > other ? bound : other;
    } else if (lane < W) {
        const float m = s_mz[lane];
        const float tol = (float)(lane < K ? cfg.fragment_mz_tolerance : cfg.precursor_mz_tolerance);
        float t = tol * m;
        float q = t / 1000000.0f;
        const int a = index_im::tof_lower_bound(run, (double)(m - q)), b = index_im::tof_lower_bound(run, (double)(m + q));
        s_tlo[lane] = a;
        s_thi[lane] = b > a ? b : a;
    }
    const int S = r.n_scans, F = r.n_cycles, L = run.cycle_len, SM = run.scan_max, z = run.zeroth;
    const int c0 = r.cycle_start;
    unsigned char *body = scratch + r.scratch_off + SEL_HEADER;
    __syncthreads();
    const double q_lo = (double)s_mz[K], q_hi = (double)s_mz[K + n_iso - 1];
    const uint64_t ph64 = (uint64_t)((int64_t)(c0 + F) * L + z) * (uint64_t)SM;
    const uint32_t push_lo = (uint32_t)(c0 * L + z) * (uint32_t)SM;
    const uint32_t push_hi = ph64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ph64;

    // ---- sparse form (the scheme of adh_gather_im_kernel): (window, TOF bin) pairs -> one range of events
    // each (the bin's events of the tile's cycles are contiguous) -> one stream of raw events; those in the
    // scan range queue up, look up their quadrupole row, and the survivors are sorted by (cell, stream
    // position) and summed per cell in that order: TOF ascending, then push - the reference's order
    // (bruker_jit.py:575-580: float32 running sum).
    const int P = index_im::pair_setup(
        run, W, s_tlo, s_thi, [](int w) { return w; }, c0, F, push_lo, push_hi, w_p0, w_base, p_lo, p_off, p_win, lane);
    const int64_t n_cells = (int64_t)W * S * F;
    bool over = P > ADH_IM_PAIR_CAP || run.n_events >= 0xFFFFFFFFll || n_cells >= (1 << 23) ||
                debug_dense != 0;  // (developer switch ADH_DEBUG_SELECT_IM_DENSE: dense tiles for every precursor)
    SelEntry *out_list = reinterpret_cast<SelEntry *>(body);
    const uint32_t out_cap = (uint32_t)(n_cells * 4 / (int64_t)sizeof(SelEntry));
    uint32_t out_n = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const double inv_sm = 1.0 / (double)SM, inv_l = 1.0 / (double)L;
    int m = 0;  // events in the list (wave-uniform)
    auto flush = [&]() {
        __syncthreads();
        if (m > 1) {
            if (m <= ADH_WAVE) gather_im::sort_keys<1>(s_key, m, lane);
            else if (m <= 2 * ADH_WAVE) gather_im::sort_keys<2>(s_key, m, lane);
            else if (m <= 4 * ADH_WAVE) gather_im::sort_keys<4>(s_key, m, lane);
            else gather_im::sort_keys<8>(s_key, m, lane);
        }
        __syncthreads();
        for (int e0 = 0; e0 < m; e0 += ADH_WAVE) {
            const int e = e0 + lane;
            bool owner = false;
            SelEntry en;
            en.cell = 0u, en.x = 0.0f;
            if (e < m) {
                const uint32_t cell = s_key[e] >> 9;
                owner = e == 0 || (s_key[e - 1] >> 9) != cell;
                if (owner) {
                    float x = 0.0f;
                    for (int q = e; q < m && (s_key[q] >> 9) == cell; ++q) x = x + (float)s_int[s_key[q] & 511u];
                    en.cell = cell, en.x = x;
                }
            }
            const unsigned long long mask = __ballot(owner);
            if (owner) {
                const uint32_t at = out_n + (uint32_t)__popcll(mask & lt);
                if (at < out_cap) out_list[at] = en;
            }
            out_n += (uint32_t)__popcll(mask);
        }
        __syncthreads();
        if (out_n > out_cap) over = true;
        m = 0;
    };
    int w0 = 0;
    while (w0 < W && !over) {
        int w1 = w0 + 1;
        if (m > 0 && (uint32_t)m + (p_off[w_p0[w1]] - p_off[w_p0[w0]]) > ADH_IM_SORT_CAP) {
            flush();
            if (over) break;
        }
        while (w1 < W && (uint32_t)m + (p_off[w_p0[w1 + 1]] - p_off[w_p0[w0]]) <= ADH_IM_SORT_CAP) ++w1;
        const int pa0 = w_p0[w0], pb0 = w_p0[w1];
        const uint32_t r0 = p_off[pa0], r1 = p_off[pb0];
        if (r1 - r0 > 0xFFFFu) {  // (raw numbers are queued as 16-bit offsets)
            over = true;
            break;
        }
        const int nq = index_im::queue_scan_range(run, pa0, pb0, r0, r1, r.scan_start, r.scan_start + S, m, w_base, p_win, p_lo, p_off,
                                                  s_key, s_int, s_pair, lane);
        if (m + nq > ADH_IM_SORT_CAP) {  // (only a single window can be this full)
            over = true;
            break;
        }
        __syncthreads();
        const int q_base = m;
        for (int q0 = 0; q0 < nq; q0 += ADH_WAVE) {
            const int qi = q0 + lane;
            bool ok = false;
            uint32_t cell = 0u;
            uint16_t ni = 0;
            if (qi < nq) {
                const uint32_t pvq = s_key[q_base + qi];
                const int pa = (int)s_pair[q_base + qi];
                const uint32_t e = r0 + (uint32_t)s_int[q_base + qi];
                const int w = (int)p_win[pa];
                const int64_t idx = w_base[w] + (int64_t)p_lo[pa] + (int64_t)(e - p_off[pa]);
                const bool prec = w >= K;
                uint32_t fq = (uint32_t)((double)pvq * inv_sm);
                if (pvq - fq * (uint32_t)SM >= (uint32_t)SM) ++fq;
                const int frame = (int)fq, scan = (int)(pvq - fq * (uint32_t)SM);
                uint32_t cq = (uint32_t)((double)(frame - z) * inv_l);
                if ((uint32_t)(frame - z) - cq * (uint32_t)L >= (uint32_t)L) ++cq;
                const int crow = (frame - z - (int)cq * L) * SM + scan;
                const double cy0 = run.cycle[2 * crow], cy1 = run.cycle[2 * crow + 1];
                ni = run.inten[idx];
                const double ql = prec ? -1.0 : q_lo, qh = prec ? -1.0 : q_hi;
                ok = ql <= cy1 && qh >= cy0;
                cell = (uint32_t)((w * S + (scan - r.scan_start)) * F + ((int)cq - c0));
            }
            __syncthreads();  // the queue slots of this step are in registers: the list may grow over them
            const unsigned long long mask = __ballot(ok);
            if (ok) {
                const int pos = m + __popcll(mask & lt);
                s_key[pos] = (cell << 9) | (uint32_t)pos;
                s_int[pos] = ni;
            }
            m += __popcll(mask);
        }
        __syncthreads();
        w0 = w1;
    }
    if (!over && m > 0) flush();
    if (!over) {
        // first entry of every window (the list is sorted by cell)
        __syncthreads();
        for (int w = lane; w <= W; w += ADH_WAVE) {
            const uint32_t target = (uint32_t)((int64_t)w * S * F);
            uint32_t a = 0, b = out_n;
            while (a < b) {
                const uint32_t mid = (a + b) >> 1;
                if (out_list[mid].cell < target) a = mid + 1; else b = mid;
            }
            header[4 + w] = a;
        }
        if (lane == 0) {
            header[0] = (uint32_t)K;
            header[1] = (uint32_t)W;
            header[2] = SEL_COMPACT;
            header[3] = out_n;
        }
        return;
    }

    // ---- dense form: zero the tiles, one lane per window walks the TOF bins of the window: the bin's events
    // of all F cycles are contiguous (pushes ascend with the frame).  A cell is touched by exactly one lane
    // in (TOF index, push) order, so its running float32 sum is the reference's.
    __syncthreads();
    float *tiles = reinterpret_cast<float *>(body);
    for (int64_t c = lane; c < n_cells; c += ADH_WAVE) tiles[c] = 0.0f;
    if (lane == 0) {
        header[0] = (uint32_t)K;
        header[1] = (uint32_t)W;
        header[2] = SEL_DENSE;
        header[3] = 0u;
    }
    __syncthreads();
    for (int w = lane; w < W; w += ADH_WAVE) {
        const bool prec = w >= K;
        const double ql = prec ? -1.0 : q_lo, qh = prec ? -1.0 : q_hi;
        float *cells = tiles + (size_t)w * S * F;
        for (int tof = s_tlo[w]; tof < s_thi[w]; ++tof) {
            const int64_t b = run.tof_indptr[tof + 1];
            int64_t lo = run.tof_indptr[tof], hi = b;
            while (lo < hi) {
                const int64_t mm = (lo + hi) >> 1;
                if (run.push[mm] < push_lo) lo = mm + 1; else hi = mm;
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

// LDS of the smoothing kernel for tiles of at most cap_cells = S * F cells and cap_s scans
size_t adh_select_smooth_im_lds_bytes(int cap_cells, int cap_s, int k0, int k1, int tap_budget) {
    size_t b = (size_t)cap_cells * 8;                    // rows + log-sum tile
    b += (size_t)(k0 + k1) * 8;                          // kernel factors
#if ADH_SEL_LOG_LDS
    b += 256 * 8;                                        // the log table
#endif
    b += (size_t)cap_s * 4 * 2 * selim::SEL_BATCH + 16;  // row tables and tap counts of a batch of windows
    b += (size_t)tap_budget * 2 + 8;                     // tap lists of pass 2 (uint16 entries)
    return (b + 15) / 16 * 16;
}
// Entries of the tap lists a launch can afford: the lists take LDS, and the kernel wants two blocks per CU
// (80 KB each) more than it wants the lists; with one block per CU anyway they may fill what is left.
// 0: no lists (kernels of more than 64 taps or tiles of more than 1023 scans do not fit the 16-bit entries).
int adh_select_tap_budget(int cap_cells, int cap_s, int k0, int k1) {
    if (k0 > 64 || cap_s > 1023) return 0;
    const size_t base = adh_select_smooth_im_lds_bytes(cap_cells, cap_s, k0, k1, 0);
    const size_t two_blocks = 80 * 1024, one_block = 150 * 1024;
    const size_t room = base + 2048 <= two_blocks ? two_blocks - base : (base + 2048 <= one_block ? one_block - base : 0);
    const size_t want = (size_t)selim::SEL_BATCH * cap_s * (size_t)std::min(cap_s, k0) * 2;  // every list at its longest
    return (int)(std::min(room > 64 ? room - 64 : 0, want) / 2);
}
// LDS of the score kernel
size_t adh_select_score_im_lds_bytes(int cap_cells, int cap_s, int cap_f) {
    size_t b = (size_t)cap_cells * 8;                    // the float64 scores
    b += (size_t)selim::PAR_PEAKS * (cap_s + cap_f) * 8; // scan / cycle profiles of PAR_PEAKS peaks at a time
    b += (size_t)((cap_cells + 31) / 32) * 4;            // peak flags, one bit per cell
    return (b + 15) / 16 * 16;
}

// Kernel 2 of 3: the smoothed log-sum feature tile of every precursor, written to the parked tile of its
// scratch block.  Separate from the peak search (kernel 3) because it is bound by LDS latency and needs
// wavefronts: 1024 threads at <= 64 VGPRs, two blocks = 32 wavefronts per CU (the peak search carries
// per-candidate arrays in registers and would cap the whole at half of that).
// (DBG: the instantiation that honours the developer switches ADH_DEBUG_SELECT_IM_ABL; the product runs the one in which
// they are compile-time zeros - their tests sat in the innermost loops and their flags in scalar registers the kernel
// has to spill)
template <bool DBG>
__global__ __launch_bounds__(selim::SMOOTH_THREADS, 8) void adh_select_smooth_im_kernel(
    const selim::PrecRec *__restrict__ recs, int32_t n_prec, const double *__restrict__ ku_g,
    const double *__restrict__ kv_g, int32_t k0, int32_t k1, int32_t cap_cells, int32_t cap_s,
    unsigned char *__restrict__ scratch, int32_t debug_abl_arg, int32_t tap_budget) {
    using namespace selim;
    const int32_t debug_abl = DBG ? debug_abl_arg : 0;
    constexpr int SCORE_THREADS = SMOOTH_THREADS;  // (this kernel's block size, under the name the loops use)
    extern __shared__ __align__(16) unsigned char smem[];
    // The smoothing is the separable circular convolution
    //   out(s, f) = sum_a ku[a] * (sum_b kv[b] * x[(s + k0/2 - a) mod S][(f + k1/2 - b) mod F])
    // with float64 fused multiply-adds in tap order and one rounding to float32 per pass (as in the oracle).
    // A tile holds a handful of events, and a zero input leaves a running fma sum as it is, so only the
    // scans WITH events are kept ("rows": compacted, in scan order): pass 1 runs over those rows, and
    // pass 2 adds, for every output cell, the rows within the kernel's reach in tap order.  What is left
    // of the dense algorithm is the log per non-zero output cell.
    float *rows = reinterpret_cast<float *>(smem);      // [n_rows][F] events, then pass-1 result (in place)
    float *ls = rows + cap_cells;                       // log-sum tile: fragments first (parked in HBM when done), then isotopes
    double *ku = reinterpret_cast<double *>(ls + cap_cells);
    double *kv = ku + k0;
#if ADH_SEL_LOG_LDS
    double *ltab = kv + k1;  // adh_log_tab, 2 KB: the look-up of a log is an LDS read instead of a global load
    double *after_k = ltab + 256;
#else
    const double *ltab = &adh_log_tab[0][0];
    double *after_k = kv + k1;
#endif
    // per window g of a batch, at [g * S ...]:
    int16_t *row_slot = reinterpret_cast<int16_t *>(after_k);  // [S] slot of a scan, -1: no event
    int16_t *row_list = row_slot + SEL_BATCH * cap_s;   // [n_rows] scans with events, ascending
    int16_t *row_top = row_list + SEL_BATCH * cap_s;    // [S] last row <= (s + k0/2) mod S (circular), as index into row_list
    // Tap lists of pass 2, per (window of the batch, scan): the rows within the kernel's reach in tap order, as
    // (row slot << 6 | tap).  The walk that finds them depends on the scan only; done by every thread of a row
    // of cells it was 40 % of this kernel (index arithmetic, wrap-arounds and the break: 16 instructions per
    // tap).  A window's lists are min(rows of the window, k0) entries per scan; a batch whose lists exceed the
    // launch's budget (dense tiles) walks as before.
    int16_t *tap_cnt = row_top + SEL_BATCH * cap_s;  // [S] per window
    uint16_t *tap_tab = reinterpret_cast<uint16_t *>(tap_cnt + SEL_BATCH * cap_s);
    __shared__ int batch_rows[SEL_BATCH];  // scans with events of every window of the batch
    const int tid = threadIdx.x;
    const int i = blockIdx.x;
    if (i >= n_prec) return;
    const PrecRec r = recs[i];
    const uint32_t *header = reinterpret_cast<const uint32_t *>(scratch + r.scratch_off);
    const int K = (int)header[0], W = (int)header[1];
    if (!r.ok || K == 0) return;
    const bool compact = header[2] == SEL_COMPACT;
    const unsigned char *body = scratch + r.scratch_off + SEL_HEADER;
    const SelEntry *entries = reinterpret_cast<const SelEntry *>(body);
    const float *tiles = reinterpret_cast<const float *>(body);
    const int S = r.n_scans, F = r.n_cycles, SF = S * F;
    float *park = reinterpret_cast<float *>(scratch + r.scratch_off + SEL_HEADER) + (size_t)W * SF;  // the parked tile
    for (int c = tid; c < k0; c += SCORE_THREADS) ku[c] = ku_g[c];
    for (int c = tid; c < k1; c += SCORE_THREADS) kv[c] = kv_g[c];
#if ADH_SEL_LOG_LDS
    if (tid < 256) ltab[tid] = adh_log_tab[tid >> 1][tid & 1];
#endif
    for (int c = tid; c < SF; c += SCORE_THREADS) ls[c] = 0.0f;
    __syncthreads();
    const int h0 = k0 / 2, h1 = k1 / 2;
    // The factors of a Gaussian over 30 taps are exact zeros outside ~15 of them, and a zero tap leaves a running fma
    // sum as it is: pass 1 runs over the columns [b_lo, b_hi] only, the tap lists of pass 2 hold the non-zero taps -
    // and a scan whose list is empty (no event within reach of a non-zero tap) has nothing to smooth.
    int b_lo = 0, b_hi = k1 - 1;
    while (b_lo < b_hi && kv[b_lo] == 0.0) ++b_lo;
    while (b_hi > b_lo && kv[b_hi] == 0.0) --b_hi;
    const double inv_f = 1.0 / (double)F;
    const int rows_per_step = max(SCORE_THREADS / F, 1);  // pass 1: whole rows per step (F <= 256: host check)
    // Windows are taken SEL_BATCH at a time: the block-wide phases (row discovery, row fill, pass 1) and
    // their barriers are paid once per batch; pass 2 then runs window after window without a barrier (a
    // thread owns its cells, the rows are read-only), which keeps the float32 log sums in window order.
    const double inv_sf = 1.0 / (double)SF;
    const int wv = tid / ADH_WAVE, ln = tid % ADH_WAVE;
    for (int w0 = 0; w0 < W;) {
        const int gmax = min(SEL_BATCH, (w0 < K ? K : W) - w0);  // (never across the fragment / isotope boundary)
        // ---- which scans have events, per window of the batch
        for (int c = tid; c < gmax * S; c += SCORE_THREADS) row_slot[c] = -1;
        __syncthreads();
        const int e_lo = compact ? (int)header[4 + w0] : 0, e_hi = compact ? (int)header[4 + w0 + gmax] : 0;
        if (compact) {
            for (int e = e_lo + tid; e < e_hi; e += SCORE_THREADS) {
                const int rem = (int)entries[e].cell - w0 * SF;
                int g = (int)((double)rem * inv_sf);
                if (rem - g * SF >= SF) ++g;
                const int r2 = rem - g * SF;
                int sc = (int)((double)r2 * inv_f);
                if (r2 - sc * F >= F) ++sc;
                row_slot[g * S + sc] = 0;
            }
        } else {
            for (int c = tid; c < gmax * SF; c += SCORE_THREADS)
                if (tiles[(size_t)w0 * SF + c] != 0.0f) {
                    int g = (int)((double)c * inv_sf);
                    if (c - g * SF >= SF) ++g;
                    const int r2 = c - g * SF;
                    int sc = (int)((double)r2 * inv_f);
                    if (r2 - sc * F >= F) ++sc;
                    row_slot[g * S + sc] = 0;
                }
        }
        __syncthreads();
        // compact them in scan order: wavefront g takes window g (ballots, no block barrier)
        if (wv < gmax) {
            int cnt = 0;
            for (int base = 0; base < S; base += ADH_WAVE) {
                const int sc = base + ln;
                const bool has = sc < S && row_slot[wv * S + sc] == 0;
                const unsigned long long mask = __ballot(has);
                if (has) {
                    const int slot = cnt + __popcll(mask & ((1ull << ln) - 1ull));
                    row_slot[wv * S + sc] = (int16_t)slot;
                    row_list[wv * S + slot] = (int16_t)sc;
                }
                cnt += __popcll(mask);
            }
            if (ln == 0) batch_rows[wv] = cnt;
        }
        __syncthreads();
        // the windows of the batch whose rows fit the row storage together (S rows; one window always fits)
        // ... and whose tap lists fit what LDS is left (S lists of min(rows, k0) entries per window): a window that would
        // push the batch into the per-cell walk of pass 2 waits for the next batch - batches are cheap, the walk is not
        int gc = 0, tot = 0, base_r[SEL_BATCH], taps = 0;
#pragma unroll
        for (int g = 0; g < SEL_BATCH; ++g) {
            base_r[g] = tot;
            const int tl = g < gmax ? min(batch_rows[g], k0) : 0;
            const bool lists_fit = !ADH_SEL_TAP_FIT || g == 0 || S * (taps + tl) <= tap_budget;
            if (g < gmax && gc == g && tot + batch_rows[g] <= S && lists_fit) {
                tot += batch_rows[g];
                taps += tl;
                gc = g + 1;
            }
        }
        if (tot > 0) {
            // Pass 1 from the entries themselves (sparse tiles, the normal case): a row holds one or two events, and a
            // zero input leaves a running fma sum as it is, so output cell f of a row needs the row's EVENTS in tap order
            // (descending column, circularly, from column f + k1/2 - b_lo), not the 15 columns of the kernel's support
            // read back from a zero-filled row.  The entries are sorted by cell = (window, scan, cycle): a row is a
            // run of entries; the first entry and the first cell of every row are noted behind the rows (the tail of
            // the row storage), and the support must not be longer than a row (no column met twice).  No zero fill, no
            // scatter of the events, one barrier instead of two per step.
            const int e_end = compact ? (int)header[4 + w0 + gc] : 0;
            // (at most four events per row on average - denser batches take the dense pass -, and the entries themselves
            // are staged behind the row starts: 10 tot words in all)
            const bool sparse1 = compact && e_end - e_lo <= 4 * tot && tot * (F + 10) <= SF && b_hi - b_lo + 1 <= F &&
                                 debug_abl != 8;  // (8: developer switch, always the dense pass)
            int *row_e0 = reinterpret_cast<int *>(rows) + (SF - 10 * tot);  // first entry of every row, counted from e_lo
            int *row_c0 = row_e0 + tot;                                     // first cell of every row
            uint32_t *ent_w = reinterpret_cast<uint32_t *>(row_c0 + tot);   // the batch's entries: (cell, intensity bits)
            if (sparse1) {
                for (int e = e_lo + tid; e < e_end; e += SCORE_THREADS) {
                    const SelEntry en = entries[e];
                    ent_w[2 * (e - e_lo)] = en.cell;
                    ent_w[2 * (e - e_lo) + 1] = __float_as_uint(en.x);
                    const int cell = (int)en.cell;
                    const int rem = cell - w0 * SF;
                    int g = (int)((double)rem * inv_sf);
                    if (rem - g * SF >= SF) ++g;
                    const int r2 = rem - g * SF;
                    int sc = (int)((double)r2 * inv_f);
                    if (r2 - sc * F >= F) ++sc;
                    const int c0 = cell - (r2 - sc * F);  // first cell of the row
                    if (e == e_lo || (int)entries[e - 1].cell < c0) {
                        int br = 0;
#pragma unroll
                        for (int q = 0; q < SEL_BATCH; ++q) br = q == g ? base_r[q] : br;
                        const int slot = br + (int)row_slot[g * S + sc];
                        row_e0[slot] = e - e_lo;
                        row_c0[slot] = c0;
                    }
                }
            } else {
            // the rows of the events, dense along the cycles
            for (int c = tid; c < tot * F; c += SCORE_THREADS) rows[c] = 0.0f;
            __syncthreads();
            if (compact) {
                for (int e = e_lo + tid; e < e_end; e += SCORE_THREADS) {
                    const SelEntry en = entries[e];
                    const int rem = (int)en.cell - w0 * SF;
                    int g = (int)((double)rem * inv_sf);
                    if (rem - g * SF >= SF) ++g;
                    const int r2 = rem - g * SF;
                    int sc = (int)((double)r2 * inv_f);
                    if (r2 - sc * F >= F) ++sc;
                    int br = 0;
#pragma unroll
                    for (int q = 0; q < SEL_BATCH; ++q) br = q == g ? base_r[q] : br;
                    rows[(br + (int)row_slot[g * S + sc]) * F + (r2 - sc * F)] = en.x;
                }
            } else {
#pragma unroll
                for (int g = 0; g < SEL_BATCH; ++g)
                    if (g < gc)
                        for (int c = tid; c < batch_rows[g] * F; c += SCORE_THREADS) {
                            int slot = (int)((double)c * inv_f);
                            if (c - slot * F >= F) ++slot;
                            rows[(base_r[g] + slot) * F + (c - slot * F)] =
                                tiles[(size_t)(w0 + g) * SF + (int)row_list[g * S + slot] * F + (c - slot * F)];
                        }
            }
            }
            // for every scan: the last row at or below (s + h0) mod S, circularly
            for (int c = tid; c < gc * S; c += SCORE_THREADS) {
                const int g = c / S, sc = c - g * S;
                const int nr = batch_rows[g];
                int top = sc + h0;
                top -= top >= S ? S : 0;
                int lo = 0, hi = nr;  // number of rows <= top
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((int)row_list[g * S + mid] <= top) lo = mid + 1; else hi = mid;
                }
                row_top[c] = (int16_t)(lo == 0 ? nr - 1 : lo - 1);
            }
            // tap lists, if those of the batch fit: window g gets S lists of tap_len[g] entries at tap_base[g]
            int tap_len[SEL_BATCH], tap_base[SEL_BATCH], tap_total = 0;
#pragma unroll
            for (int g = 0; g < SEL_BATCH; ++g) {
                tap_len[g] = g < gc ? min(batch_rows[g], k0) : 0;
                tap_base[g] = tap_total;
                tap_total += S * tap_len[g];
            }
            const bool use_taps = tap_total <= tap_budget && debug_abl != 5;  // (5: developer switch, always walk)
            if (use_taps) {
                for (int c = tid; c < gc * S; c += SCORE_THREADS) {
                    const int g = c / S, sc = c - g * S;
                    const int nr = batch_rows[g];
                    int tb = 0, tl = 0;
#pragma unroll
                    for (int q = 0; q < SEL_BATCH; ++q) {
                        tb = q == g ? tap_base[q] : tb;
                        tl = q == g ? tap_len[q] : tl;
                    }
                    uint16_t *tp = tap_tab + tb + sc * tl;
                    // (row_top[c] is this thread's own write of the loop above)
                    int idx = (int)row_top[c], cnt = 0;
                    for (int t = 0; t < nr; ++t) {
                        int a = sc + h0 - (int)row_list[g * S + idx];  // tap of this row: (sc + h0 - row) mod S, ascending along the walk
                        a += a < 0 ? S : 0;
                        a -= a >= S ? S : 0;
                        if (a >= k0) break;
                        if (ku[a] != 0.0) tp[cnt++] = (uint16_t)(idx << 6 | a);
                        idx = idx == 0 ? nr - 1 : idx - 1;
                    }
                    tap_cnt[c] = (int16_t)cnt;
                }
            }
            __syncthreads();
            // ---- pass 1, in place: along the cycles, kernel centred at column k1 / 2.  A step takes whole rows:
            // every thread reads the taps of its cell, then all write.
            if (sparse1) {
                const int lr = tid / F, f = tid - lr * F;
                int cs = f + h1 - b_lo;  // column of the first tap of the support
                cs += cs < 0 ? F : 0;
                cs -= cs >= F ? F : 0;
                for (int slot = lr; slot < (lr < rows_per_step ? tot : 0); slot += rows_per_step) {
                    const int e0 = row_e0[slot], e1 = slot + 1 < tot ? row_e0[slot + 1] : e_end - e_lo, c0 = row_c0[slot];
                    const int n = e1 - e0;
                    const uint32_t *ew = ent_w + 2 * e0;
                    // the last event at or below column cs (none: the row's last event, one turn earlier)
                    int i = n - 1;
                    for (int j = n - 1; j >= 0; --j)
                        if ((int)ew[2 * j] - c0 <= cs) {
                            i = j;
                            break;
                        }
                    double acc = 0.0;
                    for (int t = 0; t < n; ++t) {
                        int bb = cs - ((int)ew[2 * i] - c0);  // tap of this event, counted from b_lo: ascending along the walk
                        bb += bb < 0 ? F : 0;
                        bb += b_lo;
                        if (bb > b_hi) break;
                        acc = fma(kv[bb], (double)__uint_as_float(ew[2 * i + 1]), acc);
                        i = i == 0 ? n - 1 : i - 1;
                    }
                    rows[__mul24(slot, F) + f] = (float)acc;
                }
                __syncthreads();
            } else
            for (int base = 0; base < tot; base += rows_per_step) {
                const int lr = tid / F, f = tid - lr * F;
                const int slot = base + lr;
                const bool act = lr < rows_per_step && slot < tot;
                float v = 0.0f;
                if (act) {
                    const float *rp = rows + slot * F;
                    double acc = 0.0;
                    int col = f + h1 - b_lo;  // column of tap b: (f + h1 - b) mod F
                    col += col < 0 ? F : 0;
                    col -= col >= F ? F : 0;
                    for (int bb = b_lo; bb <= b_hi; ++bb) {
                        acc = fma(kv[bb], (double)rp[col], acc);
                        col = col == 0 ? F - 1 : col - 1;
                    }
                    v = (float)acc;
                }
                __syncthreads();
                if (act) rows[slot * F + f] = v;
                __syncthreads();
            }
            // ---- pass 2: along the scans, kernel centred at row k0 / 2; log(smooth + 1) summed per group.
            // (a thread keeps its column and strides over the scans: no index arithmetic per cell)
            const int p2_rows = SCORE_THREADS / F, p2_lr = tid / F, p2_f = tid - p2_lr * F;
#pragma unroll
            for (int g = 0; g < SEL_BATCH; ++g) {
                if (g >= gc) continue;
                const int n_rows = batch_rows[g];
                if (n_rows == 0) continue;  // an empty tile smooths to zeros: log(0 + 1) = 0 changes nothing
                const int16_t *rl = row_list + g * S, *rt = row_top + g * S;
                const float *rw = rows + base_r[g] * F;
                for (int sc = p2_lr; sc < (debug_abl == 2 || p2_lr >= p2_rows ? 0 : S); sc += p2_rows) {  // (2: developer ablation, no pass 2)
                    const int f = p2_f, c = __mul24(sc, F) + f;  // (24-bit multiply-adds in the cell and tap loops: one instruction each)
                    int idx = (int)rt[sc];
                    double acc = 0.0;
                    if (use_taps) {
                        const uint16_t *tp = tap_tab + tap_base[g] + __mul24(sc, tap_len[g]);
                        const int cnt = debug_abl == 3 ? 0 : (int)tap_cnt[g * S + sc];
                        if (cnt == 0) continue;  // (nothing within reach: smooth = 0, log(0 + 1) = 0)
                        for (int t = 0; t < cnt; ++t) {
                            const int u = (int)tp[t];
                            acc = fma(ku[u & 63], (double)rw[__mul24(u >> 6, F) + f], acc);
                        }
                    } else
                    for (int t = 0; t < (debug_abl == 3 ? 0 : n_rows); ++t) {  // (3: developer ablation, no row walk)
                        int a = sc + h0 - (int)rl[idx];  // tap of this row: (sc + h0 - row) mod S, ascending along the walk
                        a += a < 0 ? S : 0;
                        a -= a >= S ? S : 0;
                        if (a >= k0) break;
                        if (debug_abl == 4) acc += 1.0;  // (4: developer ablation, the walk without the taps)
                        else acc = fma(ku[a], (double)rw[__mul24(idx, F) + f], acc);
                        idx = idx == 0 ? n_rows - 1 : idx - 1;
                    }
                    const float sm = (float)acc;
                    // _build_features (selection.py:206-226).  log(1) = 0: the tails of the kernel - a smoothed value below
                    // 6e-8 - round away in the float32 sum sm + 1, and the rows and columns that hold nothing else
                    // (most of the reach of a 30 x 30 kernel) never enter the log
                    const float x = sm + 1.0f;
                    if (x != 1.0f) {
                        if (debug_abl == 1) ls[c] += x;  // (developer ablation: no log)
                        else ls[c] += (x >= 1.0f && x < INFINITY) ? (float)adh_log_f32(x, ltab) : adh_log_f32_rare(x);
                    }
                }
            }
        }
        if (w0 + gc == K) {
            // fragment sum complete: park it in the scratch block, the LDS array starts over for the isotopes
            __syncthreads();  // (the cells are owned by other threads here than in pass 2)
            for (int c = tid; c < SF; c += SCORE_THREADS) {
                park[c] = ls[c];
                ls[c] = 0.0f;
            }
        }
        __syncthreads();
        w0 += gc;
    }
    // feature = fragment sum + isotope sum (float32), left in the parked tile for the score kernel
    for (int c = tid; c < SF; c += SCORE_THREADS) park[c] = park[c] + ls[c];
}

// Kernel 3 of 3: scores, peaks, joins and limits of every precursor from its feature tile.
__global__ __launch_bounds__(selim::SCORE_THREADS, 4) void adh_select_score_im_kernel(
    DevTims run, const selim::PrecRec *__restrict__ recs, int32_t n_prec, int64_t first_prec,
    adh_selection_config_t cfg, int32_t cap_cells, int32_t cap_s, int32_t cap_f,
    unsigned char *__restrict__ scratch, DevCandTable out) {
    using namespace selim;
    extern __shared__ __align__(16) unsigned char smem[];
    // scan / cycle profiles of symetric_limits_2d, PAR_PEAKS peaks at a time
    double *mob = reinterpret_cast<double *>(smem) + cap_cells, *cyc = mob + (size_t)PAR_PEAKS * cap_s;
    uint32_t *flag = reinterpret_cast<uint32_t *>(cyc + (size_t)PAR_PEAKS * cap_f);   // peak flags, one bit per cell
    __shared__ double red_v[SCORE_THREADS / ADH_WAVE];  // one entry per wavefront (arg-max rounds)
    __shared__ int red_i[SCORE_THREADS / ADH_WAVE];
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
    const int S = r.n_scans, F = r.n_cycles, SF = S * F;
    const float *park = reinterpret_cast<const float *>(scratch + r.scratch_off + SEL_HEADER) + (size_t)W * SF;  // the feature tile
    // ---- score (selection.py:396-421): kept as the float32 feature + the affine map
    double mean = cfg.feature_mean, sd = cfg.feature_std, weight = cfg.feature_weight;
    if (!cfg.use_weighted_score) {
        // amean1 / astd1 (selection/utils.py:118-133): sequential float64 sums over the tile, by one thread.
        // The tile is staged in LDS by all threads first and read eight cells at a time (reads in flight
        // together, adds in order): from global memory this loop was two thirds of the whole selection.
        float *ftile = reinterpret_cast<float *>(smem);
        for (int c = tid; c < SF; c += SCORE_THREADS) ftile[c] = park[c];
        __syncthreads();
        if (tid == 0) {
            double m = 0;
            int c = 0;
            for (; c + 8 <= SF; c += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = ftile[c + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) m += (double)t[u];
            }
            for (; c < SF; ++c) m += (double)ftile[c];
            m /= (double)SF;
            double v = 0;
            c = 0;
            for (; c + 8 <= SF; c += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = ftile[c + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double d = (double)t[u] - m;
                    v += d * d;
                }
            }
            for (; c < SF; ++c) {
                const double d = (double)ftile[c] - m;
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
    double *score = reinterpret_cast<double *>(smem);  // the two float tiles hold SF doubles
    for (int c = tid; c < SF; c += SCORE_THREADS) {
        const float ft = park[c];
        score[c] = weight * ((double)ft - mean) / (sd + 1e-6);
    }
    for (int c = tid; c < (SF + 31) / 32; c += SCORE_THREADS) flag[c] = 0u;
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
        if (pk) atomicOr(&flag[c >> 5], 1u << (c & 31));
    }
    __syncthreads();
    const int top_n = (int)min((int64_t)MAX_CAND, cfg.candidate_count);
    int n_pk = 0;
    for (int round = 0; round < top_n; ++round) {
        double best = -INFINITY;
        int best_i = -1;
        for (int c = tid; c < SF; c += SCORE_THREADS)
            if ((flag[c >> 5] >> (c & 31)) & 1u) {
                const double v = score[c];
                if (best_i < 0 || v > best || (v == best && c > best_i)) {
                    best = v;
                    best_i = c;
                }
            }
        // the wavefront's best by shuffles, the block's best from one entry per wavefront: two barriers per round
        // (a tree over 512 LDS slots took ten)
        auto better = [](double v, int j, double bv, int bj) { return j >= 0 && (bj < 0 || v > bv || (v == bv && j > bj)); };
#pragma unroll
        for (int off = ADH_WAVE / 2; off > 0; off >>= 1) {
            const double v = __shfl_xor(best, off);
            const int j = __shfl_xor(best_i, off);
            if (better(v, j, best, best_i)) {
                best = v;
                best_i = j;
            }
        }
        if ((tid & (ADH_WAVE - 1)) == 0) {
            red_v[tid / ADH_WAVE] = best;
            red_i[tid / ADH_WAVE] = best_i;
        }
        __syncthreads();
        double bv = red_v[0];
        int bi = red_i[0];
#pragma unroll
        for (int w = 1; w < SCORE_THREADS / ADH_WAVE; ++w)
            if (better(red_v[w], red_i[w], bv, bi)) {
                bv = red_v[w];
                bi = red_i[w];
            }
        __syncthreads();  // (red_* are rewritten in the next round)
        const int win = bi;
        if (win < 0) break;  // uniform
        if (tid == 0) {
            pk_idx[n_pk] = win;
            pk_val[n_pk] = bv;
            flag[win >> 5] &= ~(1u << (win & 31));
        }
        ++n_pk;
        __syncthreads();
    }
    // ---- joins and limits, in the reference's order.  The joins are short loops for one thread; the scan /
    // cycle profiles of symetric_limits_2d are sums over S x ~10 and ~10 x F cells per peak and were, on one
    // thread, two thirds of the whole selection: one thread per scan / per cycle now (the sum of a thread
    // keeps its order).
    __shared__ int sh_scan[MAX_CAND], sh_cycle[MAX_CAND], sh_npk;
    int p_scan[MAX_CAND], p_cycle[MAX_CAND], p_sl[MAX_CAND][2], p_cl[MAX_CAND][2];
    double p_score[MAX_CAND];
    if (tid == 0) {
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
        for (int a = 0; a < n_pk; ++a) {
            sh_scan[a] = p_scan[a];
            sh_cycle[a] = p_cycle[a];
        }
        sh_npk = n_pk;
    }
    __syncthreads();
    n_pk = sh_npk;
    // symetric_limits_2d (selection/utils.py:283-312): the profiles of PAR_PEAKS peaks by all threads, then one lane
    // per peak walks its two profiles (the walks are chains of dependent LDS reads: side by side they cost one)
    __shared__ int sh_sl[MAX_CAND][2], sh_cl[MAX_CAND][2];
    for (int a0 = 0; a0 < n_pk; a0 += PAR_PEAKS) {
        const int na = min(PAR_PEAKS, n_pk - a0);
        for (int t = tid; t < na * S; t += SCORE_THREADS) {
            const int q = t / S, sc = t - q * S;
            const int pa_cycle = sh_cycle[a0 + q];
            const int cyc_lower = (int)max((int64_t)0, (int64_t)pa_cycle - cfg.min_size_rt);
            const int cyc_upper = (int)min((int64_t)F, (int64_t)pa_cycle + cfg.min_size_rt);
            double v = 0.0;
            for (int f = cyc_lower; f < cyc_upper; ++f) v += A(sc, f);
            mob[q * cap_s + sc] = v;
        }
        for (int t = tid; t < na * F; t += SCORE_THREADS) {
            const int q = t / F, f = t - q * F;
            const int pa_scan = sh_scan[a0 + q];
            const int mob_lower = (int)max((int64_t)0, (int64_t)pa_scan - cfg.min_size_mobility);
            const int mob_upper = (int)min((int64_t)S, (int64_t)pa_scan + cfg.min_size_mobility);
            double v = 0.0;
            for (int sc = mob_lower; sc < mob_upper; ++sc) v += A(sc, f);
            cyc[q * cap_f + f] = v;
        }
        __syncthreads();
        if (tid < na) {
            int sl[2], cl[2];
            sel::symetric_limits_1d(mob + tid * cap_s, S, sh_scan[a0 + tid], cfg.f_mobility, cfg.center_fraction,
                                    cfg.min_size_mobility, cfg.max_size_mobility, sl);
            sel::symetric_limits_1d(cyc + tid * cap_f, F, sh_cycle[a0 + tid], cfg.f_rt, cfg.center_fraction, cfg.min_size_rt,
                                    cfg.max_size_rt, cl);
            sh_sl[a0 + tid][0] = sl[0];
            sh_sl[a0 + tid][1] = sl[1];
            sh_cl[a0 + tid][0] = cl[0];
            sh_cl[a0 + tid][1] = cl[1];
        }
        __syncthreads();
    }
    if (tid == 0)
        for (int a = 0; a < n_pk; ++a) {
            p_sl[a][0] = sh_sl[a][0];
            p_sl[a][1] = sh_sl[a][1];
            p_cl[a][0] = sh_cl[a][0];
            p_cl[a][1] = sh_cl[a][1];
        }
    if (tid != 0) return;
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

// ---- the per-precursor plan of the ion-mobility selection on the device (round 4; 16 host threads took 16 ms per
// 200 000 precursors for it): frame limits (get_frame_indices, jitclasses/utils.py:24-88, with the zeroth frame),
// scan limits (_get_scan_indices, bruker_jit.py:204-245, with its ceil of a negative quotient), the validity rules
// of _is_valid (selection.py:40-75) and the "empty push query" exit (bruker_jit.py:516-519): one thread per
// precursor.  red: [0] error bits (1 slice outside the library, 2 charge 0, 4 more than MAX_W windows), [1] largest
// S * F, [2] largest S, [3] largest F of the precursors that go on; need[i] = bytes of the precursor's scratch
// block (a multiple of 256), biggest = the largest of them.
struct SelPlanIn {
    const uint32_t *precursor_idx, *frag_start, *frag_stop;
    const uint8_t *charge;
    const float *rt, *mobility, *mz;
};
__global__ void adh_select_plan_im_kernel(DevTims T, SelPlanIn pc, int64_t n, int64_t n_lib, int n_iso, double rt_tolerance,
                                          double mobility_tolerance, int64_t kernel_size, int k0, int k1,
                                          selim::PrecRec *__restrict__ recs, unsigned long long *__restrict__ need,
                                          int32_t *__restrict__ red, unsigned long long *__restrict__ biggest) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int L = T.cycle_len, SM = T.scan_max, z = T.zeroth;
    selim::PrecRec r;
    memset(&r, 0, sizeof(r));
    int err = 0;
    const uint32_t fs = pc.frag_start[i], fe = pc.frag_stop[i];
    if (fe < fs || (int64_t)fe > n_lib) err = 1;
    else if (pc.charge[i] == 0) err = 2;
    else if ((int64_t)(fe - fs) + n_iso > selim::MAX_W) err = 4;
    if (err) {
        atomicOr(&red[0], err);
        recs[i] = r;
        need[i] = 0ull;
        return;
    }
    r.precursor_idx = pc.precursor_idx[i];
    r.frag_start = fs;
    r.frag_stop = fe;
    r.mz = pc.mz[i];
    r.charge = pc.charge[i];
    const float lo = (float)((double)pc.rt[i] - rt_tolerance), hi = (float)((double)pc.rt[i] + rt_tolerance);
    auto rt_lower = [&](double x) {
        int64_t a = 0, b = T.n_frames;
        while (a < b) {
            const int64_t m = (a + b) >> 1;
            if (T.rt[m] < x) a = m + 1; else b = m;
        }
        return a;
    };
    auto rev_upper = [&](float v) {  // searchsorted(mobility_values[::-1], v, "right")
        int64_t a = 0, b = SM;
        while (a < b) {
            const int64_t m = (a + b) >> 1;
            if (T.mobility[SM - 1 - m] <= (double)v) a = m + 1; else b = m;
        }
        return a;
    };
    const int64_t cmax = (T.n_frames - 1) / L;  // precursor_cycle_max_index (bruker_jit.py:131)
    const int64_t c_lo = (rt_lower((double)lo) + z) / L, c_hi = (rt_lower((double)hi) + z) / L;
    int64_t len = max(c_hi - c_lo, kernel_size);
    len = 16 * (int64_t)ceil((double)len / 16.0);
    int64_t cs = c_lo, ce = c_lo + len;
    if (ce > cmax) {
        ce = cmax;
        cs = cmax - len;
        if (cs < 0) cs = (cmax % 2 == 0) ? 0 : 1;
    }
    const float m_hi = (float)((double)pc.mobility[i] + mobility_tolerance);
    const float m_lo = (float)((double)pc.mobility[i] - mobility_tolerance);
    const int64_t s_first = SM - rev_upper(m_hi), s_second = SM - rev_upper(m_lo);
    const int64_t opt_len = 16 * (int64_t)ceil((double)(s_first - s_second) / 16.0);
    int64_t ss = s_first, se = s_first - opt_len;
    if (se < 0) {
        se = 0;
        ss = min(opt_len, (int64_t)SM);
    }
    const int64_t S = max(se - ss, (int64_t)0), F = ce - cs;
    r.cycle_start = (int32_t)cs;
    r.n_cycles = (int32_t)max(F, (int64_t)0);
    r.scan_start = (int32_t)ss;
    r.n_scans = (int32_t)S;
    bool ok = F > 0 && S > 0 && n_iso > 0 && S % 2 == 0 && S >= k0 && F >= k1 && ss >= 0 && ss + S <= SM;
    if (ok) {
        const double off = (double)(n_iso - 1) * 1.0033548350700006 / (double)pc.charge[i];
        const double q_lo = (double)(float)((double)pc.mz[i] + 0.0), q_hi = (double)(float)((double)pc.mz[i] + off);
        bool any_f = false, any_p = false;
        for (int row = 0; row < L && !(any_f && any_p); ++row)
            for (int64_t sc = ss; sc < ss + S; ++sc) {
                const double wl = T.cycle[2 * ((int64_t)row * SM + sc)], wh = T.cycle[2 * ((int64_t)row * SM + sc) + 1];
                any_f = any_f || (q_lo <= wh && q_hi >= wl);
                any_p = any_p || (-1.0 <= wh && -1.0 >= wl);
                if (any_f && any_p) break;
            }
        ok = any_f && any_p;
    }
    r.ok = ok ? 1 : 0;
    recs[i] = r;
    const unsigned long long bytes =
        selim::SEL_HEADER + (ok ? (unsigned long long)(fe - fs + (uint32_t)n_iso + 1u) * (unsigned long long)S * (unsigned long long)F * 4ull : 0ull);
    const unsigned long long aligned = (bytes + 255ull) / 256ull * 256ull;
    need[i] = aligned;
    atomicMax(biggest, aligned);
    if (ok) {
        atomicMax(&red[1], (int32_t)(S * F));
        atomicMax(&red[2], (int32_t)S);
        atomicMax(&red[3], (int32_t)F);
    }
}

// scratch offsets of the batches: cum = inclusive sums of need; batch b covers the precursors [first[b], first[b + 1])
// and its blocks start at 0: offset = cum[i] - need[i] - (cum[first[b] - 1] or 0)
__global__ void adh_select_offsets_im_kernel(selim::PrecRec *__restrict__ recs, const unsigned long long *__restrict__ cum,
                                             const unsigned long long *__restrict__ need, int64_t n,
                                             const int64_t *__restrict__ first, int n_batches) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int a = 0, b = n_batches;  // last batch with first[a] <= i
    while (b - a > 1) {
        const int m = (a + b) >> 1;
        if (first[m] <= i) a = m; else b = m;
    }
    const int64_t f = first[a];
    const unsigned long long base = f > 0 ? cum[f - 1] : 0ull;
    recs[i].scratch_off = cum[i] - need[i] - base;
}
