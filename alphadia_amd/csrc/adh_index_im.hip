// adh_index_im.hip - search indices of a staged ion-mobility run (see DevTims in adh_device.h).
//
// The reference finds the events of a candidate with np.searchsorted over mz_values
// (alphadia/search/jitclasses/bruker_jit.py:273-278) and a merge-join of the push list with every
// TOF bin from the bin's first event (bruker_jit.py:415-502).  On the GPU a search is a chain of
// dependent loads, and a candidate's wavefront spends its time waiting for them; 288 GB of HBM buys
// the answers in advance: 4 bytes per (TOF bin, cycle) - 3.2 GB for the 400 000 bin x 2 000 cycle run
// of BASELINE configs[3] - turn "first event of this bin in cycle c" into one load.
#include "adh_device.h"

namespace index_im {

// number of index columns cb in [0, n_blocks] whose first push is <= p
__device__ __forceinline__ uint32_t columns_le(uint32_t p, uint32_t S_max, uint32_t L, uint32_t z, int shift,
                                               uint32_t n_blocks) {
    const uint32_t frame = p / S_max;
    if (frame < z) return 0u;
    const uint32_t blk = ((frame - z) / L) >> shift;
    return (blk < n_blocks ? blk : n_blocks) + 1u;
}

// searchsorted(mz_values, x, "left") (bruker_jit.py:273-278): the staged lookup table brackets the answer to
// a few bins; an answer on the edge of the bracket is verified and searched again over the whole table if
// the bracket was wrong (it never is unless the table and the arithmetic here disagree in the last bit).
__device__ __forceinline__ int tof_lower_bound(const DevTims &run, double x) {
    const int n_tof = (int)run.n_tof;
    int a = 0, b = n_tof;
    if (run.mz_lut) {
        const double t = (x - run.lut_min) * run.lut_inv_step;
        const int bk = !(t >= 0.0) ? 0 : (t >= (double)run.lut_n ? run.lut_n - 1 : (int)t);
        const int a0 = (int)run.mz_lut[max(bk - 1, 0)], b0 = (int)run.mz_lut[min(bk + 2, run.lut_n)];
        a = a0, b = b0;
        while (a < b) {
            const int m = (a + b) >> 1;
            if (run.mz[m] < x) a = m + 1; else b = m;
        }
        const bool ok_lo = a > a0 || a == 0 || run.mz[a - 1] < x;
        const bool ok_hi = a < b0 || a == n_tof || !(run.mz[a] < x);
        if (ok_lo && ok_hi) return a;
        a = 0, b = n_tof;
    }
    while (a < b) {
        const int m = (a + b) >> 1;
        if (run.mz[m] < x) a = m + 1; else b = m;
    }
    return a;
}

// The events of TOF bin `tof` whose push lies in [push_lo, push_hi) - the pushes of cycles c0 ... c0 + F - 1:
// one contiguous range [lo, lo2), because a bin's events ascend by push.  With one index column per cycle
// the two look-ups are the answer; with one column per block of cycles they bracket two short searches;
// without the index both ends are searched in the whole bin.
__device__ __forceinline__ void event_range(const DevTims &run, int tof, int c0, int F, uint32_t push_lo,
                                            uint32_t push_hi, int64_t &lo, int64_t &lo2) {
    int64_t hi_lo, hi_hi;
    if (run.cyc_idx) {
        // (the columns count from the bin's first event: 32 bits per column whatever the size of the run)
        const int64_t first = run.tof_indptr[tof];
        const int sh = run.cyc_shift, nb = run.cyc_cols - 1;
        const int ba = min(c0 >> sh, nb), bb = min((c0 + F) >> sh, nb);
        lo = first + (int64_t)run.cyc_word(tof, ba);
        lo2 = first + (int64_t)run.cyc_word(tof, bb);
        if (sh == 0) return;
        hi_lo = ba < nb ? first + (int64_t)run.cyc_word(tof, ba + 1) : run.tof_indptr[tof + 1];
        hi_hi = bb < nb ? first + (int64_t)run.cyc_word(tof, bb + 1) : run.tof_indptr[tof + 1];
    } else {
        lo = run.tof_indptr[tof];
        hi_lo = hi_hi = run.tof_indptr[tof + 1];
        lo2 = lo;
    }
    int64_t hi = hi_lo;
    while (lo < hi) {
        const int64_t m = (lo + hi) >> 1;
        if (run.push[m] < push_lo) lo = m + 1; else hi = m;
    }
    if (lo2 < lo) lo2 = lo;
    hi = hi_hi;
    while (lo2 < hi) {
        const int64_t m = (lo2 + hi) >> 1;
        if (run.push[m] < push_hi) lo2 = m + 1; else hi = m;
    }
}

// ---- the event stream of the sparse gathers (adh_gather_im_kernel, adh_select_gather_im_kernel), one wavefront
// per candidate / precursor.  LDS arrays of the caller: w_p0[W + 1] first (window, TOF bin) pair of every
// window, w_base[W] first event of the window's first TOF bin (64 bits: a run may hold more than 2^32 events),
// p_lo[ADH_IM_PAIR_CAP] first event of a pair's range COUNTED FROM w_base of its window (the bins of a window are
// neighbours, so 32 bits do), p_off[ADH_IM_PAIR_CAP + 1] events before the pair, p_win[ADH_IM_PAIR_CAP] window
// of the pair.

// Pairs of all W windows (window w covers the TOF bins [t_lo[slot_of(w)], t_hi[slot_of(w)])) and their event
// ranges.  Returns the number of pairs; more than ADH_IM_PAIR_CAP: nothing else is set up.
template <typename SlotOf>
__device__ __forceinline__ int pair_setup(const DevTims &run, int W, const int *t_lo, const int *t_hi, SlotOf slot_of,
                                          int c0, int F, uint32_t push_lo, uint32_t push_hi, int *w_p0,
                                          int64_t *w_base, uint32_t *p_lo, uint32_t *p_off, uint8_t *p_win, int lane) {
    for (int w = lane; w < W; w += ADH_WAVE) w_base[w] = run.tof_indptr[max(min(t_lo[slot_of(w)], (int)run.n_tof), 0)];
    if (lane == 0) {
        int acc = 0;
        for (int w = 0; w < W; ++w) {
            const int slot = slot_of(w);
            w_p0[w] = acc;
            acc += t_hi[slot] - t_lo[slot];
        }
        w_p0[W] = acc;
    }
    __syncthreads();
    const int P = w_p0[W];
    if (P > ADH_IM_PAIR_CAP) return P;
    for (int p = lane; p < P; p += ADH_WAVE) {
        int w = 0;
        while (w_p0[w + 1] <= p) ++w;
        const int tof = t_lo[slot_of(w)] + (p - w_p0[w]);
        int64_t lo, lo2;
        event_range(run, tof, c0, F, push_lo, push_hi, lo, lo2);
        p_lo[p] = (uint32_t)(lo - w_base[w]);
        p_win[p] = (uint8_t)w;
        p_off[p + 1] = (uint32_t)(lo2 - lo);
    }
    __syncthreads();
    uint32_t carry = 0;  // inclusive scan of the counts, 64 at a time
    for (int base = 0; base < P; base += ADH_WAVE) {
        uint32_t v = base + lane < P ? p_off[base + lane + 1] : 0u;
        for (int off = 1; off < ADH_WAVE; off <<= 1) {
            const uint32_t u = __shfl_up(v, off);
            if (lane >= off) v += u;
        }
        if (base + lane < P) p_off[base + lane + 1] = carry + v;
        carry += __shfl(v, ADH_WAVE - 1);
    }
    if (lane == 0) p_off[0] = 0u;
    __syncthreads();
    return P;
}

// The tile layout (DevTims::tile_ev): the tiles that the candidate's box of cycles [c0, c0 + F) x scans
// [scan_lo, scan_hi) meets.  nT = number of tiles, tile j (j < nT) = tile_at(j).
struct TileBox {
    int cb0, sb0, n_sb, nT, stride;  // nT: (cycle block, scan block) boxes the candidate meets
    __device__ __forceinline__ int tile_at(int j) const { return (cb0 + j / n_sb) * stride + sb0 + j % n_sb; }
};
// the i-th set bit of a frame mask
__device__ __forceinline__ int nth_frame(uint32_t mask, int i) {
    for (int k = 0; k < i; ++k) mask &= mask - 1u;
    return __ffs((int)mask) - 1;
}
__device__ __forceinline__ TileBox tile_box(const DevTims &run, int c0, int F, int scan_lo, int scan_hi) {
    TileBox b;
    const int cbl = run.tile_cblocks - 1;
    b.cb0 = min(c0 >> run.tile_cshift, cbl);
    const int cb1 = min((c0 + F - 1) >> run.tile_cshift, cbl);
    b.sb0 = min(max(scan_lo, 0) >> run.tile_sshift, run.tile_sblocks - 1);
    const int sb1 = min(max(scan_hi - 1, 0) >> run.tile_sshift, run.tile_sblocks - 1);
    b.n_sb = sb1 - b.sb0 + 1;
    b.nT = (cb1 - b.cb0 + 1) * b.n_sb;
    b.stride = run.tile_sblocks;
    return b;
}

// Pairs of the tile layout: (window, tile) - the events of the window's TOF bins inside one tile are ONE run
// (TOF ascending, then push).  Pair p = w * nT + j; p_lo holds the absolute first event (the layout is only built
// below 2^31 events), w_base is not used.  Returns the number of pairs.
// `n_frag`: windows [0, n_frag) are fragment windows, the rest isotope windows; with a frame-keyed layout
// (run.tile_frames > 1) a window visits the boxes of ITS frames (masks frames_f / frames_p of the plan record): pair
// p of window w = (box j, k-th frame of the window's mask), boxes outer.
template <typename SlotOf>
__device__ __forceinline__ int pair_setup_tiled(const DevTims &run, const TileBox &box, int W, const int *t_lo,
                                                const int *t_hi, SlotOf slot_of, int *w_p0, uint32_t *p_lo,
                                                uint32_t *p_off, uint8_t *p_win, int lane, int n_frag = 0,
                                                uint32_t frames_f = 1u, uint32_t frames_p = 1u) {
    const bool keyed = run.tile_frames > 1;
    const int nf_f = keyed ? __popc(frames_f) : 1, nf_p = keyed ? __popc(frames_p) : 1;
    const int per_f = box.nT * nf_f, per_p = box.nT * nf_p;
    const int n_fw = keyed ? n_frag : W;  // (not keyed: every window counts as a "fragment" window with one frame)
    const int P = n_fw * per_f + (W - n_fw) * per_p;
    for (int w = lane; w <= W; w += ADH_WAVE) w_p0[w] = w <= n_fw ? w * per_f : n_fw * per_f + (w - n_fw) * per_p;
    if (P > ADH_IM_PAIR_CAP) return P;
    for (int p = lane; p < P; p += ADH_WAVE) {
        const bool fw = p < n_fw * per_f;
        const int per = fw ? per_f : per_p, q = fw ? p : p - n_fw * per_f;
        const int w = (fw ? 0 : n_fw) + q / per, jj = q - (q / per) * per;
        const int nf = fw ? nf_f : nf_p;
        const int j = jj / nf, k = jj - j * nf;
        const int slot = slot_of(w);
        int tile = box.tile_at(j);
        if (keyed) tile = tile * run.tile_frames + nth_frame(fw ? frames_f : frames_p, k);
        const uint32_t lo = run.tile_word(tile, t_lo[slot]), hi = run.tile_word(tile, t_hi[slot]);
        p_lo[p] = lo;
        p_win[p] = (uint8_t)w;
        p_off[p + 1] = hi - lo;
    }
    __syncthreads();
    uint32_t carry = 0;  // inclusive scan of the counts, 64 at a time
    for (int base = 0; base < P; base += ADH_WAVE) {
        uint32_t v = base + lane < P ? p_off[base + lane + 1] : 0u;
        for (int off = 1; off < ADH_WAVE; off <<= 1) {
            const uint32_t u = __shfl_up(v, off);
            if (lane >= off) v += u;
        }
        if (base + lane < P) p_off[base + lane + 1] = carry + v;
        carry += __shfl(v, ADH_WAVE - 1);
    }
    if (lane == 0) p_off[0] = 0u;
    __syncthreads();
    return P;
}

// Stage 1 of a batch of windows (pairs [pa0, pb0), raw events [r0, r1)): eight pushes per lane and step
// (eight independent loads in flight); the events inside the scan range [scan_lo, scan_hi) - ~3 % - are
// queued behind the m entries of the list: s_key = push, s_int = raw number - r0, s_pair = pair.  Returns the
// number of queued events (entries beyond CAP are counted, not stored).
// TILED: the events come from the tile layout (p_lo absolute; s_key = frame << tile_sbits | scan); a tile also
// holds cycles outside the candidate's, so the frame range [frame_lo, frame_hi) is tested here as well.
template <bool TILED = false, int CAP = ADH_IM_SORT_CAP>
__device__ __forceinline__ int queue_scan_range(const DevTims &run, int pa0, int pb0, uint32_t r0, uint32_t r1,
                                                int scan_lo, int scan_hi, int m, const int64_t *w_base,
                                                const uint8_t *p_win, const uint32_t *p_lo, const uint32_t *p_off,
                                                uint32_t *s_key, uint16_t *s_int, uint8_t *s_pair, int lane,
                                                uint32_t frame_lo = 0u, uint32_t frame_hi = 0xFFFFFFFFu) {
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t S_max = (uint32_t)run.scan_max;
    const double inv_smax = 1.0 / (double)S_max;
    const uint32_t smask = TILED ? (1u << run.tile_sbits) - 1u : 0u;
    int nq = 0;
    constexpr int U = 8;
    for (uint32_t e0 = r0; e0 < r1; e0 += U * ADH_WAVE) {
        uint32_t pv[U];
        int pa_u[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t eu = e0 + (uint32_t)(u * ADH_WAVE + lane);
            const uint32_t e = eu < r1 ? eu : r0;
            int pa = pa0, pb = pb0;  // the pair of raw event e: last pair with p_off <= e
            while (pb - pa > 1) {
                const int mid = (pa + pb) >> 1;
                if (p_off[mid] <= e) pa = mid; else pb = mid;
            }
            pa_u[u] = pa;
            if (TILED) pv[u] = run.tile_ev[(size_t)p_lo[pa] + (size_t)(e - p_off[pa])].x;
            else pv[u] = run.push[w_base[p_win[pa]] + (int64_t)p_lo[pa] + (int64_t)(e - p_off[pa])];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t eu = e0 + (uint32_t)(u * ADH_WAVE + lane);
            int scan;
            bool in_frames = true;
            if (TILED) {
                scan = (int)(pv[u] & smask);
                const uint32_t frame = pv[u] >> run.tile_sbits;
                in_frames = frame >= frame_lo && frame < frame_hi;
            } else {
                // (exact quotient without the integer-division sequence: float64 estimate, one fix-up)
                uint32_t fq = (uint32_t)((double)pv[u] * inv_smax);
                if (pv[u] - fq * S_max >= S_max) ++fq;
                scan = (int)(pv[u] - fq * S_max);
            }
            const bool pass = eu < r1 && scan >= scan_lo && scan < scan_hi && in_frames;
            const unsigned long long mask = __ballot(pass);
            if (pass) {
                const int at = m + nq + __popcll(mask & lt);
                if (at < CAP) {
                    s_key[at] = pv[u];
                    s_int[at] = (uint16_t)(eu - r0);
                    s_pair[at] = (uint8_t)pa_u[u];
                }
            }
            nq += __popcll(mask);
        }
    }
    return nq;
}

}  // namespace index_im

// one wavefront per TOF bin: the bin's events are read 64 at a time; an event that is the first one at or
// past a column's threshold writes that column
__global__ __launch_bounds__(ADH_WAVE) void adh_index_im_kernel(const int64_t *__restrict__ tof_indptr,
                                                                  const uint32_t *__restrict__ push, int64_t n_tof,
                                                                  uint32_t S_max, uint32_t L, uint32_t z, int shift,
                                                                  uint32_t n_blocks, uint32_t *__restrict__ idx) {
    using namespace index_im;
    const int lane = threadIdx.x;
    const uint32_t cols = n_blocks + 1u;
    for (int64_t tof = blockIdx.x; tof < n_tof; tof += gridDim.x) {
        const int64_t a = tof_indptr[tof], b = tof_indptr[tof + 1];
        // (column-major: word (cb, tof) at cb * n_tof + tof.  Neighbouring bins are written by neighbouring
        // blocks at about the same time, so the stores of a column meet in the L2 line they share.)
        uint32_t *row = idx + (size_t)tof;
        const size_t cs = (size_t)n_tof;  // stride of a column
        uint32_t done = 0u;  // columns written so far (wave-uniform)
        for (int64_t base = a; base < b; base += ADH_WAVE) {
            const int64_t e = base + lane;
            const bool in = e < b;
            const uint32_t g = in ? columns_le(push[e], S_max, L, z, shift, n_blocks) : cols;
            uint32_t g_prev = __shfl_up(g, 1);
            if (lane == 0) g_prev = done;
            if (in)
                for (uint32_t cb = g_prev; cb < g; ++cb) row[cb * cs] = (uint32_t)(e - a);  // (counted from the bin's first event)
            const int last = (int)min((int64_t)ADH_WAVE, b - base) - 1;
            done = __shfl(g, last);
        }
        for (uint32_t cb = done + lane; cb < cols; cb += ADH_WAVE) row[cb * cs] = (uint32_t)(b - a);
    }
}

// ---- the tile layout (DevTims::tile_ev / tile_idx): key = tile * (n_tof + 1) + TOF bin of every event, the event
// packed as the sort's value, and the histogram of the keys (its exclusive scan is tile_idx).  One wavefront per
// TOF bin as above; the stable sort keeps the push order inside a (tile, bin).
__global__ __launch_bounds__(ADH_WAVE) void adh_tile_key_kernel(const int64_t *__restrict__ tof_indptr,
                                                                  const uint32_t *__restrict__ push,
                                                                  const uint16_t *__restrict__ inten, int64_t n_tof,
                                                                  uint32_t S_max, uint32_t L, uint32_t z, int csh, int ssh,
                                                                  int sbits, uint32_t ncb, uint32_t nsb, uint32_t n_fr, uint32_t *__restrict__ keys,
                                                                  uint64_t *__restrict__ vals, uint32_t *__restrict__ hist) {
    const int lane = threadIdx.x;
    for (int64_t tof = blockIdx.x; tof < n_tof; tof += gridDim.x) {
        const int64_t a = tof_indptr[tof], b = tof_indptr[tof + 1];
        for (int64_t e = a + lane; e < b; e += ADH_WAVE) {
            const uint32_t p = push[e];
            const uint32_t frame = p / S_max, scan = p - frame * S_max;
            const uint32_t cyc = frame < z ? 0u : (frame - z) / L;
            const uint32_t fr = (n_fr > 1u && frame >= z) ? (frame - z) - cyc * L : 0u;  // frame inside the cycle (keyed layout)
            const uint32_t cb = min(cyc >> csh, ncb - 1u), sb = min(scan >> ssh, nsb - 1u);
            const uint32_t key = ((cb * nsb + sb) * n_fr + fr) * (uint32_t)(n_tof + 1) + (uint32_t)tof;
            keys[e] = key;
            vals[e] = (uint64_t)(frame << sbits | scan) | (uint64_t)inten[e] << 32 | (uint64_t)((uint32_t)tof & 0xFFFFu) << 48;
            atomicAdd(&hist[key], 1u);
        }
    }
}
