// adh_gather_im.hip - fragment selection + XIC gather for ion-mobility (timsTOF) runs.
//
// Replaces, per candidate,
//   FragmentContainer filters                   alphadia/search/jitclasses/fragment_container.py:56-102
//   TimsTOFTransposeJIT.get_dense (fragments    alphadia/search/jitclasses/bruker_jit.py:273-504,586-615
//   and isotopes)
//
// The reference builds, per call, a Python list of every (frame, scan) push of the candidate
// box whose quadrupole window overlaps, then merge-joins it with the event list of every TOF
// bin in the m/z window.  Here membership of an event in that list is decided arithmetically
// (frame range, scan range, quadrupole test on the cycle table).  An ion-mobility tile is ~1 %
// full, so it is not built: per (window, TOF bin) the bin's events of the candidate's cycles are one
// contiguous range (found through the staged (bin, cycle) index), the ranges are streamed, the few
// events that pass the tests are sorted by (cell, stream position) and folded per cell in that
// order - TOF ascending, then push ascending, the reference's order, so the running
// intensity-weighted m/z is reproduced exactly - and the non-zero cells leave as sorted
// (cell, intensity, m/z) entries (adh_device.h: ADH_IM_MODE_COMPACT).  A candidate that does not fit
// the LDS lists gets the dense tiles in its HBM scratch block instead: zero fill, then independent
// (window, cycle) tasks, one per lane, each the only one to touch its cells.
#include "adh_device.h"

// Events of one batch of windows the sort list holds (the isotope windows always go together; a candidate over the
// limit takes the materialised tiles).  Round 6 measured 1 024 against 512 on configs[3] (the isotope windows of a
// candidate ON a peptide hold 300 - 800 events, so most of those candidates are over 512): features 8.84 -> 8.34 ms
// (fewer materialised tiles), gather 4.73 -> 5.72 ms (3.6 KB more LDS per block) - worse in sum, so it stays 512.
#ifndef ADH_IM_GATHER_SORT_CAP
#define ADH_IM_GATHER_SORT_CAP ADH_IM_SORT_CAP
#endif
namespace gather_im {
constexpr int GCAP = ADH_IM_GATHER_SORT_CAP;
constexpr int GBITS = GCAP <= 512 ? 9 : 10;  // a list entry: cell << GBITS | position in the list
static_assert(GCAP <= 1024 && (GCAP & (GCAP - 1)) == 0, "list positions are 9 or 10 bits");
constexpr double ISOTOPE_DELTA = 1.0033548350700006;  // candidate.py:160

// Bitonic sort of the first m keys of an LDS array (m <= 64 * NR) by one wavefront, in registers:
// element e = r * 64 + lane.  A partner 64 or more elements away lives in the same lane (a register
// swap), a closer one in lane ^ j (one cross-lane read): no LDS traffic and no barrier between the
// log^2 stages, which is what the sort costs when it is done in LDS.
template <int NR>
__device__ __forceinline__ void sort_keys(uint32_t *keys, int m, int lane) {
    uint32_t k[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) k[r] = r * ADH_WAVE + lane < m ? keys[r * ADH_WAVE + lane] : 0xFFFFFFFFu;
#pragma unroll
    for (int kk = 2; kk <= ADH_WAVE * NR; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            if (j >= ADH_WAVE) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int pr = r ^ (j / ADH_WAVE);
                    if (pr > r) {
                        const bool up = ((r * ADH_WAVE) & kk) == 0;  // (kk > 64 here: the lane bits do not matter)
                        const uint32_t a = k[r], b = k[pr];
                        if ((a > b) == up) {
                            k[r] = b;
                            k[pr] = a;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const uint32_t other = __shfl_xor(k[r], j);
                    const bool up = ((r * ADH_WAVE + lane) & kk) == 0;
                    const bool lower = (lane & j) == 0;
                    k[r] = (lower == up) ? min(k[r], other) : max(k[r], other);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
        if (r * ADH_WAVE + lane < m) keys[r * ADH_WAVE + lane] = k[r];
}
}

// LDS of one block: window tables, then ONE region used twice - by the library slice while the
// fragments are selected, by the pair ranges and the event list of the compact mode afterwards.
// (Occupancy is what this latency-bound kernel lives on: 10 KB -> 20 KB per block costs 70 %.)
namespace gather_im {
constexpr size_t kCompactBytes = (size_t)ADH_IM_PAIR_CAP * 4 + (size_t)(ADH_IM_PAIR_CAP + 1) * 4 + 4 +  // p_lo, p_off
                                 (size_t)GCAP * (4 + 2 + 1) + ADH_IM_PAIR_CAP;                // s_key, s_int, s_pair, p_win
}
size_t adh_gather_im_lds_bytes(const Caps &c) {
    size_t b = (size_t)(c.k + c.i) * (4 + 4 + 4);  // window m/z, tof start, tof stop
    b += (size_t)(c.k + c.i + 1) * 4;              // first pair of every window
    b = (b + 15) / 16 * 16;
    b += (size_t)(c.k + c.i) * 8;                  // first event of every window's first TOF bin
    b = (b + 15) / 16 * 16;
    const size_t lib = (size_t)c.n_lib * 16;       // l_int, l_mz, l_rank, l_ok
    b += lib > gather_im::kCompactBytes ? lib : gather_im::kCompactBytes;
    return (b + 15) / 16 * 16;
}

#ifndef ADH_GATHER_IM_WAVES
#define ADH_GATHER_IM_WAVES 1  // wavefronts per SIMD the register allocation is held to (A/B of round 6: 8)
#endif
__global__ __launch_bounds__(ADH_WAVE, ADH_GATHER_IM_WAVES) void adh_gather_im_kernel(
    DevTims run, const LibRec *__restrict__ lib, const CandRecIM *__restrict__ plan,
    adh_scoring_config_t cfg, int32_t n_iso_cols, unsigned char *__restrict__ scratch, DevOut out,
    Caps caps) {
    using namespace gather_im;
    extern __shared__ __align__(16) unsigned char smem[];
    const int n_win = caps.k + caps.i;
    float *w_mz = reinterpret_cast<float *>(smem);
    int *t_lo = reinterpret_cast<int *>(w_mz + n_win);
    int *t_hi = t_lo + n_win;
    int *w_p0 = t_hi + n_win;  // first (window, bin) pair of every window
    int64_t *w_base = reinterpret_cast<int64_t *>(smem + ((size_t)n_win * 12 + (size_t)(n_win + 1) * 4 + 15) / 16 * 16);
    unsigned char *region = reinterpret_cast<unsigned char *>(w_base) + ((size_t)n_win * 8 + 15) / 16 * 16;
    // fragment selection
    float *l_int = reinterpret_cast<float *>(region);
    float *l_mz = l_int + caps.n_lib;
    int *l_rank = reinterpret_cast<int *>(l_mz + caps.n_lib);
    int *l_ok = l_rank + caps.n_lib;
    // compact mode: event ranges of the (window, TOF bin) pairs, and the list of surviving events
    uint32_t *p_lo = reinterpret_cast<uint32_t *>(region);   // [ADH_IM_PAIR_CAP] first event of the range
    uint32_t *p_off = p_lo + ADH_IM_PAIR_CAP;                 // [ADH_IM_PAIR_CAP + 1] events before the pair
    uint32_t *s_key = p_off + ADH_IM_PAIR_CAP + 2;            // [GCAP] cell << GBITS | position in the list
    uint16_t *s_int = reinterpret_cast<uint16_t *>(s_key + GCAP);
    uint8_t *s_pair = reinterpret_cast<uint8_t *>(s_int + GCAP);
    uint8_t *p_win = s_pair + GCAP;                // [ADH_IM_PAIR_CAP] window of the pair

    const int lane = threadIdx.x;
    const CandRecIM &r = plan[blockIdx.x];
    if (r.flags & ADH_FLAG_SKIP) return;
    const uint32_t row = r.row;
    if (lane == 0 && out.precursor_idx) {
        out.precursor_idx[row] = r.precursor_idx;  // candidate.py:175-176
        out.rank[row] = r.rank;
    }
    unsigned char *block = scratch + r.scratch_off;
    uint32_t *header = reinterpret_cast<uint32_t *>(block);
    LibRec *sel = reinterpret_cast<LibRec *>(block + 32);

    // ---- fragments: slice, cardinality filter, top-k by intensity, sort by m/z
    const int64_t frag_start = r.frag_start;
    const int n_lib = (int)(r.frag_stop - r.frag_start);
    for (int j = lane; j < n_lib; j += ADH_WAVE) {
        LibRec rec = lib[frag_start + j];
        l_int[j] = rec.intensity;
        l_mz[j] = rec.mz;
        l_ok[j] = !(cfg.exclude_shared_ions && rec.cardinality > 1);
    }
    __syncthreads();
    for (int a = lane; a < n_lib; a += ADH_WAVE) {
        int rk = -1;
        if (l_ok[a]) {
            rk = 0;
            float ia = l_int[a];
            for (int b = 0; b < n_lib; ++b) {
                if (!l_ok[b]) continue;
                float ib = l_int[b];
                rk += (ib > ia) || (ib == ia && b > a);
            }
            if (rk >= (int)cfg.top_k_fragments) rk = -1;
        }
        l_rank[a] = rk;
    }
    __syncthreads();
    int K = 0;
    if (n_lib <= ADH_WAVE) K = __popcll(__ballot(lane < n_lib && l_rank[lane] >= 0));  // (l_rank[lane]: this lane's own write)
    else
        for (int a = 0; a < n_lib; ++a) K += l_rank[a] >= 0;
    const int L = run.cycle_len, S_max = run.scan_max, z = run.zeroth;
    const int c0 = (r.frame_start - z) / L;
    const int F = (r.frame_stop - z) / L - c0;
    const int S = r.scan_stop - r.scan_start;
    const int O = r.n_obs, Op = r.n_ms1;
    const int I = min(n_iso_cols, (int)cfg.top_k_isotopes);
    if ((K <= 3 && caps.stop_phase != ADH_DEBUG_DENSE) || K <= 0 || F <= 0 || S <= 0 || O <= 0) {  // candidate.py:190,230; no push matches the quadrupole
        if (lane == 0) {
            header[0] = 0;
            header[1] = 0;
        }
        return;
    }
    for (int a = lane; a < n_lib; a += ADH_WAVE) {
        int ra = l_rank[a];
        if (ra < 0) continue;
        float ma = l_mz[a];
        int slot = 0;
        for (int b = 0; b < n_lib; ++b) {
            int rb = l_rank[b];
            if (rb < 0) continue;
            float mb = l_mz[b];
            slot += (mb < ma) || (mb == ma && rb < ra);
        }
        LibRec pick = lib[frag_start + a];
        pick.pad0 = (uint8_t)(a & 0xFF);  // position inside the library slice (adh_output_t.fragment_lib_slot)
        pick.pad1 = (uint8_t)(a >> 8);
        sel[slot] = pick;
        w_mz[slot] = ma;
    }
    if (lane < I) {
        double off = (double)lane * ISOTOPE_DELTA / (double)r.charge;
        w_mz[caps.k + lane] = (float)off + r.precursor_mz;
    }
    __syncthreads();
    // TOF index limits: searchsorted(mz_values, mass_range(...), "left") (bruker_jit.py:273-278)
    if (K + I <= ADH_WAVE / 2) {
        // two lanes per window, one per end: the two searches are dependent look-ups of a wavefront that has nothing else
        // to do (lane w: the lower end, lane w + 32: the upper end)
        const int w = lane & (ADH_WAVE / 2 - 1);
        const bool upper = lane >= ADH_WAVE / 2, act = w < K + I;
        const bool prec = w >= K;
        const int slot = prec ? caps.k + (w - K) : w;
        int bound = 0;
        if (act) {
            float mzq = w_mz[slot];
            float tol = prec ? cfg.precursor_mz_tolerance : cfg.fragment_mz_tolerance;
            float t = tol * mzq;
            float q = t / 1000000.0f;
            bound = index_im::tof_lower_bound(run, (double)(upper ? mzq + q : mzq - q));
        }
        const int other = __shfl(bound, lane ^ (ADH_WAVE / 2));  // the lower end, seen from the upper lane
        if (act && !upper) t_lo[slot] = bound;
        if (act && upper) t_hi[slot] = bound > other ? bound : other;
    } else
    for (int w = lane; w < K + I; w += ADH_WAVE) {
        const bool prec = w >= K;
        const int slot = prec ? caps.k + (w - K) : w;
        float mzq = w_mz[slot];
        float tol = prec ? cfg.precursor_mz_tolerance : cfg.fragment_mz_tolerance;
        float t = tol * mzq;
        float q = t / 1000000.0f;
        const int a = index_im::tof_lower_bound(run, (double)(mzq - q));
        const int b = index_im::tof_lower_bound(run, (double)(mzq + q));
        t_lo[slot] = a;
        t_hi[slot] = b > a ? b : a;
    }
    // zero the tile
    float2 *fcells = reinterpret_cast<float2 *>(block + adh_scratch_frag_off(r.k_cap));
    float2 *pcells = reinterpret_cast<float2 *>(block + adh_im_prec_off(r.k_cap, O, S, F));
    const int n_fc = K * O * S * F, n_pc = I * Op * S * F;
    __syncthreads();
    if (caps.stop_phase == 7) {  // developer ablation: selection + window limits only
        if (lane == 0) header[0] = 0;
        return;
    }

    // quadrupole range of the fragments (candidate.py:203-205)
    float iso_min = w_mz[caps.k], iso_max = w_mz[caps.k];
    for (int i = 1; i < I; ++i) {
        iso_min = fminf(iso_min, w_mz[caps.k + i]);
        iso_max = fmaxf(iso_max, w_mz[caps.k + i]);
    }
    double fq_lo = (double)(float)((double)iso_min - 0.5), fq_hi = (double)(float)((double)iso_max + 0.5);
    if (caps.stop_phase == ADH_DEBUG_DENSE) {
        fq_lo = (double)caps.dbg_q_lo;
        fq_hi = (double)caps.dbg_q_hi;
    }

    // one step of the running sums of a cell, bruker_jit.py:440-485 (absolute_masses=True): uint16
    // intensity, float64 m/z
    auto fold = [](float &vx, float &vy, int64_t ni, double measured) {
        float am = vy * vx;
        double num = (double)am + (double)ni * measured + 1e-36;
        double den = ((double)vx + (double)ni) + 1e-36;
        vy = (float)(num / den);
        vx = (float)((double)vx + (double)ni);
    };
    uint32_t hits = 0;
    bool dense = caps.stop_phase == ADH_DEBUG_DENSE || caps.stop_phase == 8;  // (8: developer switch, dense mode only)
    const int W = K + I;
    if (!dense) {
        // ---- compact mode: the tile is never materialised.
        //  1. one lane per (window, TOF bin) pair: two binary searches give the events of the bin that lie
        //     in the candidate's cycles - ONE contiguous range, because a bin's events ascend by push
        //  2. the ranges of a group of windows form one stream of raw events, 64 per round, one per lane:
        //     scan / quadrupole tests, survivors compacted in stream order (bin, then push: the order in
        //     which the reference folds the events of a cell) into an LDS list
        //  3. the list is sorted by (cell, stream position), one lane per cell folds its events, and the
        //     non-zero cells leave as (cell, intensity, m/z) entries, already in cell order
        // The isotope windows form the last group; its cells are keyed (scan, cycle, isotope, MS1 row) and
        // the MS1 rows are collapsed here (candidate.py:248-269), so the feature kernel finds all isotopes
        // of a precursor cell next to each other.
        const uint64_t ph64 = (uint64_t)((int64_t)(c0 + F) * L + z) * (uint64_t)S_max;
        const uint32_t push_lo = (uint32_t)(c0 * L + z) * (uint32_t)S_max;
        const uint32_t push_hi = ph64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ph64;
        // the tile layout when the run has it and the candidate's (window, tile) pairs fit (a window of more than 256
        // TOF bins does not: list entries name their bin by one byte): otherwise the (window, TOF bin) ranges
        auto slot_of = [&](int w) { return w >= K ? caps.k + (w - K) : w; };
        index_im::TileBox box{};
        bool tiled = run.tile_ev != nullptr;
        if (tiled) {
            box = index_im::tile_box(run, c0, F, r.scan_start, r.scan_stop);
            bool wide = false;
            for (int w = lane; w < W; w += ADH_WAVE) wide |= t_hi[slot_of(w)] - t_lo[slot_of(w)] > 256;
            // (a frame-keyed layout: a window visits the boxes of its own frames)
            const int per_f = run.tile_frames > 1 ? __popc(r.frames_f) : 1, per_p = run.tile_frames > 1 ? __popc(r.frames_p) : 1;
            tiled = (K * per_f + I * per_p) * box.nT <= ADH_IM_PAIR_CAP && !__any(wide);
        }
        const int P = tiled ? index_im::pair_setup_tiled(run, box, W, t_lo, t_hi, slot_of, w_p0, p_lo, p_off, p_win, lane, K,
                                                         r.frames_f, r.frames_p)
                            : index_im::pair_setup(run, W, t_lo, t_hi, slot_of, c0, F, push_lo, push_hi, w_p0, w_base, p_lo,
                                                   p_off, p_win, lane);
        if (caps.stop_phase == 15) {  // developer ablation: ... + pair ranges
            if (lane == 0) header[0] = 0;
            return;
        }
        bool over = P > ADH_IM_PAIR_CAP || W > 255 || run.n_events >= 0xFFFFFFFFll || (int64_t)n_fc + n_pc >= (1 << (32 - GBITS)) || I > 12 || F >= 4096 || S >= 32768;  // (limits of the packed cell ids)
        const double inv_smax = 1.0 / (double)S_max, inv_l = 1.0 / (double)L;
        ImEntry *out_list = reinterpret_cast<ImEntry *>(block + adh_scratch_frag_off(r.k_cap));
        const uint32_t out_cap =
            (uint32_t)((adh_im_tiles_end(r.k_cap, O, S, F, I, Op) - adh_scratch_frag_off(r.k_cap)) / sizeof(ImEntry));
        uint32_t out_n = 0, n_fe = 0;
        const unsigned long long lt = (1ull << lane) - 1ull;
        int m = 0;  // events in the list (wave-uniform)

        // sort the list by (cell, position) and fold it: one lane per output cell adds up the events of the
        // cell in list order.  A fragment cell is one tile cell; a precursor cell (scan, cycle, isotope)
        // collapses its MS1 rows (candidate.py:248-269: sum of the intensities, mean of the non-zero m/z).
        auto flush = [&]() {
            __syncthreads();
            if (m > 1) {  // in registers, element r * 64 + lane
                if (m <= ADH_WAVE) sort_keys<1>(s_key, m, lane);
                else if (m <= 2 * ADH_WAVE) sort_keys<2>(s_key, m, lane);
                else if (m <= 4 * ADH_WAVE) sort_keys<4>(s_key, m, lane);
                else if (m <= 8 * ADH_WAVE) sort_keys<8>(s_key, m, lane);
                else sort_keys<16>(s_key, m, lane);
            }
            __syncthreads();
            auto group_of = [&](uint32_t cell) -> uint32_t {
                return cell < (uint32_t)n_fc ? cell : (uint32_t)n_fc + (cell - (uint32_t)n_fc) / (uint32_t)Op;
            };
            for (int e0 = 0; e0 < m; e0 += ADH_WAVE) {
                const int e = e0 + lane;
                bool owner = false, frag = false;
                ImEntry en;
                en.cell = 0u, en.x = 0.0f, en.y = 0.0f;
                if (e < m) {
                    const uint32_t cell = s_key[e] >> GBITS;
                    const uint32_t gid = group_of(cell);
                    owner = e == 0 || group_of(s_key[e - 1] >> GBITS) != gid;
                    frag = cell < (uint32_t)n_fc;
                    if (owner) {
                        // TOF bin of an event: first bin of its window + pair - first pair of the window
                        const int w = frag ? (int)cell / (O * S * F) : K + (int)((gid - (uint32_t)n_fc) % (uint32_t)I);
                        // (tile layout: the list entry holds bin - first bin of the window)
                        const int tof0 = t_lo[frag ? w : caps.k + (w - K)] - (tiled ? 0 : w_p0[w]);
                        float acc = 0.0f, last_y = 0.0f;
                        double sum = 0.0;
                        int count = 0;
                        int q = e;
                        while (q < m && group_of(s_key[q] >> GBITS) == gid) {
                            const uint32_t c = s_key[q] >> GBITS;
                            float vx = 0.0f, vy = 0.0f;
                            for (; q < m && (s_key[q] >> GBITS) == c; ++q) {
                                const int pos = (int)(s_key[q] & (uint32_t)(GCAP - 1));
                                fold(vx, vy, (int64_t)s_int[pos], run.mz[tof0 + (int)s_pair[pos]]);
                            }
                            acc += vx;
                            sum += (double)vy;
                            count += vy > 0.0f;
                            last_y = vy;
                        }
                        en.cell = frag ? cell : gid - (uint32_t)n_fc;
                        en.x = acc;  // (a fragment cell: 0 + vx)
                        en.y = frag ? last_y : (float)(sum / ((double)count + 1e-6));
                    }
                }
                const unsigned long long mask = __ballot(owner);
                if (owner) {
                    const uint32_t at = out_n + (uint32_t)__popcll(mask & lt);
                    if (at < out_cap) out_list[at] = en;
                }
                out_n += (uint32_t)__popcll(mask);
                n_fe += (uint32_t)__popcll(__ballot(owner && frag));
            }
            __syncthreads();
            if (out_n > out_cap) over = true;
            m = 0;
        };

        // ---- stage 1 of the windows [wa, wb): the raw events; those in the scan range (and, in a tile, in the
        // candidate's cycles) queue up behind the list.  Returns their number (queue entries beyond the list's
        // capacity are counted, not stored).
        auto stage_one = [&](int wa, int wb, int scan_a, int scan_b) -> int {
            const int pa0 = w_p0[wa], pb0 = w_p0[wb];
            const uint32_t r0 = p_off[pa0], r1 = p_off[pb0];
            return tiled ? index_im::queue_scan_range<true, GCAP>(run, pa0, pb0, r0, r1, scan_a, scan_b, m, w_base, p_win, p_lo,
                                                            p_off, s_key, s_int, s_pair, lane, (uint32_t)(c0 * L + z),
                                                            (uint32_t)((c0 + F) * L + z))
                         : index_im::queue_scan_range<false, GCAP>(run, pa0, pb0, r0, r1, scan_a, scan_b, m, w_base, p_win, p_lo,
                                                             p_off, s_key, s_int, s_pair, lane);
        };
        // ---- stage 2 of a batch whose first raw event is r0: the nq queued events look up their quadrupole row, MS1 /
        // MS2 observation and intensity (independent loads) and the survivors take their place in the list, in stream order
        auto stage_two = [&](const int nq, const uint32_t r0) {
            const int q_base = m;
            for (int q0 = 0; q0 < nq; q0 += ADH_WAVE) {
                const int qi = q0 + lane;
                bool ok = false;
                uint32_t cell = 0u, pair = 0u;
                uint16_t ni = 0;
                if (qi < nq) {
                    const uint32_t pvq = s_key[q_base + qi];
                    const int pa = (int)s_pair[q_base + qi];
                    const uint32_t e = r0 + (uint32_t)s_int[q_base + qi];
                    const int w = (int)p_win[pa];
                    const bool prec = w >= K;
                    int frame, scan;
                    if (tiled) {  // (frame << tile_sbits | scan)
                        frame = (int)(pvq >> run.tile_sbits);
                        scan = (int)(pvq & ((1u << run.tile_sbits) - 1u));
                    } else {
                        uint32_t fq = (uint32_t)((double)pvq * inv_smax);
                        if (pvq - fq * (uint32_t)S_max >= (uint32_t)S_max) ++fq;
                        frame = (int)fq, scan = (int)(pvq - fq * (uint32_t)S_max);
                    }
                    uint32_t cq = (uint32_t)((double)(frame - z) * inv_l);
                    if ((uint32_t)(frame - z) - cq * (uint32_t)L >= (uint32_t)L) ++cq;
                    const int f = (int)cq - c0;
                    const int crow = (frame - z - (int)cq * L) * S_max + scan;
                    const double cy0 = run.cycle[2 * crow], cy1 = run.cycle[2 * crow + 1];
                    const int pc = run.dpc[crow];
                    if (tiled) {  // (the line was fetched by stage 1)
                        const uint32_t iw = run.tile_ev[(size_t)p_lo[pa] + (size_t)(e - p_off[pa])].y;
                        ni = (uint16_t)(iw & 0xFFFFu);
                        pair = ((iw >> 16) - (uint32_t)t_lo[prec ? caps.k + (w - K) : w]) & 0xFFFFu;
                    } else {
                        ni = run.inten[w_base[w] + (int64_t)p_lo[pa] + (int64_t)(e - p_off[pa])];
                        pair = (uint32_t)pa;
                    }
                    const double q_lo = prec ? -1.0 : fq_lo, q_hi = prec ? -1.0 : fq_hi;
                    const int n_o = prec ? Op : O;
                    const uint16_t *obs = prec ? r.ms1_obs : r.obs;
                    int o = 0;
                    while (o < n_o && (int)obs[o] != pc) ++o;
                    const int sc = scan - r.scan_start;
                    ok = q_lo <= cy1 && q_hi >= cy0 && o < n_o;  // (o < n_o always: the plan lists every overlapping row)
                    cell = prec ? (uint32_t)(n_fc + (((sc * F + f) * I + (w - K)) * Op + o))
                                : (uint32_t)(((w * O + o) * S + sc) * F + f);
                }
                __syncthreads();  // the queue slots of this step are in registers: the list may grow over them
                const unsigned long long mask = __ballot(ok);
                if (ok) {
                    const int pos = m + __popcll(mask & lt);
                    s_key[pos] = (cell << GBITS) | (uint32_t)pos;
                    s_pair[pos] = (uint8_t)pair;
                    s_int[pos] = ni;
                }
                m += __popcll(mask);
                hits += (uint32_t)__popcll(mask);
            }
        };
        int w0 = 0;
        bool try_all = caps.stop_phase != 14;  // (14: developer switch, certain batches only)
        while (w0 < W && !over) {
            // The next batch of windows.  First guess: ALL that are left - few raw events survive stage 1, and a
            // candidate that gets through its windows in one batch has one chain of dependent loads (events ->
            // quadrupole rows -> m/z of the bins) instead of one per batch, which is what this kernel waits for.
            // If the survivors do not fit the list: as many windows as are certain to fit its free part (every raw
            // event might survive); the isotope windows go together, whatever their size.
            const bool pg = w0 >= K;
            int w1 = W, nq = -1;
            if (try_all && p_off[w_p0[W]] - p_off[w_p0[w0]] <= 0xFFFFu) {
                nq = stage_one(w0, W, r.scan_start, r.scan_stop);
                if (m + nq > GCAP) {
                    nq = -1;
                    try_all = false;
                    __syncthreads();
                }
            }
            if (nq < 0) {
                w1 = pg ? W : w0 + 1;
                if (m > 0 && (uint32_t)m + (p_off[w_p0[w1]] - p_off[w_p0[w0]]) > GCAP) {
                    flush();
                    if (over) break;
                }
                if (!pg)
                    while (w1 < K && (uint32_t)m + (p_off[w_p0[w1 + 1]] - p_off[w_p0[w0]]) <= GCAP) ++w1;
                if (p_off[w_p0[w1]] - p_off[w_p0[w0]] > 0xFFFFu) {  // (raw numbers are queued as 16-bit offsets)
                    over = true;
                    break;
                }
                nq = stage_one(w0, w1, r.scan_start, r.scan_stop);
                if (m + nq > GCAP) {
                    // Only a single window or the isotope group can be this full (m = 0 here: what was in the list has
                    // been flushed) - a candidate ON a peptide: 300 - 800 isotope events.  Round 6: its scans in 2, 4,
                    // ... 32 parts, each sorted, folded and emitted by itself, lowest scans first.  The entries of the
                    // isotope group are in (scan, cycle, isotope) order and those of one fragment window with one
                    // observation in (scan, cycle) order, so the parts' entries ARE the sorted list; only what is left
                    // (two observations in one overfull window, a part of one scan that does not fit) takes the
                    // materialised tiles, which cost 110 KB of zeros written and read per candidate and the slowest
                    // path of the feature kernels.
                    bool fit = false;
                    int parts = 2;
                    if (pg || O == 1) {
                        for (; parts <= 32 && !fit; parts *= 2) {
                            fit = true;
                            for (int q = 0; q < parts && fit; ++q) {
                                const int sa = r.scan_start + (int)((int64_t)S * q / parts), sb = r.scan_start + (int)((int64_t)S * (q + 1) / parts);
                                if (sb > sa && stage_one(w0, w1, sa, sb) > GCAP) fit = false;
                                __syncthreads();
                            }
                            if (fit) break;
                        }
                    }
                    if (!fit) {
                        over = true;
                        break;
                    }
                    const uint32_t r0p = p_off[w_p0[w0]];
                    for (int q = 0; q < parts && !over; ++q) {
                        const int sa = r.scan_start + (int)((int64_t)S * q / parts), sb = r.scan_start + (int)((int64_t)S * (q + 1) / parts);
                        if (sb <= sa) continue;
                        const int nq_p = stage_one(w0, w1, sa, sb);
                        __syncthreads();
                        stage_two(nq_p, r0p);
                        __syncthreads();
                        if (m > 0) flush();
                    }
                    if (over) break;
                    w0 = w1;
                    continue;
                }
            }
            const uint32_t r0 = p_off[w_p0[w0]];
            __syncthreads();
            if (caps.stop_phase == 16) {  // developer ablation: ... + stage 1 (the raw events of the first batch)
                if (lane == 0) header[0] = 0;
                return;
            }
            stage_two(nq, r0);
            __syncthreads();
            w0 = w1;
        }
        if (caps.stop_phase == 17) {  // developer ablation: ... + stage 2, no sort / fold / entries
            if (lane == 0) header[0] = 0;
            return;
        }
        if (!over && m > 0) flush();
        dense = over;  // too many events for the lists: this candidate takes the dense path below
        if (!dense) {
            if (lane == 0) {
                header[0] = (uint32_t)K;
                header[1] = hits;
                header[2] = n_fe;
                header[3] = ADH_IM_MODE_COMPACT;
                header[4] = out_n - n_fe;
            }
            return;
        }
        if (caps.stop_phase == 13 || caps.dbg_drop_dense) {  // developer ablation: candidates that overflow the lists are dropped
            if (lane == 0) header[0] = 0;
            return;
        }
        hits = 0;
        __syncthreads();
    }

    // ---- dense mode: zero the tiles; (window, cycle) tasks, one per lane: a task walks the TOF bins of its
    // window in ascending order and, inside a bin, the events of its cycle.  A tile cell is only ever
    // touched by one task, in the reference's order (TOF ascending, then push ascending).
    for (int c = lane; c < n_fc; c += ADH_WAVE) fcells[c] = make_float2(0.0f, 0.0f);
    for (int c = lane; c < n_pc; c += ADH_WAVE) pcells[c] = make_float2(0.0f, 0.0f);
    __syncthreads();
    const bool per_cycle = run.cyc_idx != nullptr && run.cyc_shift == 0;
    for (int t = lane; t < (K + I) * F; t += ADH_WAVE) {
        const int w = t / F, f = t - w * F;
        const bool prec = w >= K;
        const int slot = prec ? caps.k + (w - K) : w;
        const int j = prec ? (w - K) : w;
        const double q_lo = prec ? -1.0 : fq_lo, q_hi = prec ? -1.0 : fq_hi;
        const int n_o = prec ? Op : O;
        const uint16_t *obs = prec ? r.ms1_obs : r.obs;
        float2 *cells = prec ? pcells : fcells;
        const int frame_lo = (c0 + f) * L + z;
        const uint32_t push_lo = (uint32_t)frame_lo * (uint32_t)S_max;
        const uint32_t push_hi = (uint32_t)(frame_lo + L) * (uint32_t)S_max;
        for (int tof = t_lo[slot]; tof < t_hi[slot]; ++tof) {
            const double measured = run.mz[tof];
            int64_t lo, b;
            if (per_cycle) {  // the staged (bin, cycle) index holds both ends
                const int64_t first = run.tof_indptr[tof];  // (the columns count from the bin's first event)
                lo = first + (int64_t)run.cyc_word(tof, c0 + f);
                b = first + (int64_t)run.cyc_word(tof, min(c0 + f + 1, run.cyc_cols - 1));
            } else {
                b = run.tof_indptr[tof + 1];
                lo = run.tof_indptr[tof];
                int64_t hi = b;
                while (lo < hi) {
                    int64_t m = (lo + hi) >> 1;
                    if (run.push[m] < push_lo) lo = m + 1; else hi = m;
                }
            }
            for (int64_t idx = lo; idx < b; ++idx) {
                const uint32_t p = run.push[idx];
                if (p >= push_hi) break;
                const int frame = (int)(p / (uint32_t)S_max), scan = (int)(p % (uint32_t)S_max);
                if (scan < r.scan_start || scan >= r.scan_stop) continue;
                const int crow = (frame - frame_lo) * S_max + scan;
                if (!(q_lo <= run.cycle[2 * crow + 1] && q_hi >= run.cycle[2 * crow])) continue;
                const int pc = run.dpc[crow];
                int o = 0;
                while (o < n_o && (int)obs[o] != pc) ++o;
                if (o >= n_o) continue;  // cannot happen: the plan lists every overlapping row
                const int cell = ((j * n_o + o) * S + (scan - r.scan_start)) * F + f;
                float2 v = cells[cell];
                fold(v.x, v.y, (int64_t)run.inten[idx], measured);
                cells[cell] = v;
                ++hits;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) hits += __shfl_xor(hits, off);
    if (lane == 0) {
        header[0] = (uint32_t)K;
        header[1] = hits;
        header[2] = 0u;
        header[3] = ADH_IM_MODE_DENSE;
        header[4] = 0u;
    }
}
