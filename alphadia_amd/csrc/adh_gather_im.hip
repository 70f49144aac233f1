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
// (frame range, scan range, quadrupole test on the cycle table), and the work is split into
// independent (fragment, cycle) tasks, one per lane: a task walks the TOF bins of its fragment
// in ascending order and, inside a bin, binary-searches the first event of its cycle.  A tile
// cell (fragment, observation, scan, cycle) is only ever touched by one task, in the same
// order as in the reference (TOF ascending, then push ascending), so the running
// intensity-weighted m/z is reproduced exactly.  The TOF-major event lists are read with
// 4-byte / 2-byte loads; the tile itself lives in the candidate's HBM scratch block
// (zero-filled first) and is read back coalesced by the feature kernel.
#include "adh_device.h"

namespace gather_im {
constexpr double ISOTOPE_DELTA = 1.0033548350700006;  // candidate.py:160
}

size_t adh_gather_im_lds_bytes(const Caps &c) {
    size_t b = (size_t)c.n_lib * 16;          // l_int, l_mz, l_rank, l_ok
    b += (size_t)(c.k + c.i) * (4 + 4 + 4);   // window m/z, tof start, tof stop
    return (b + 15) / 16 * 16;
}

__global__ __launch_bounds__(ADH_WAVE) void adh_gather_im_kernel(
    DevTims run, const LibRec *__restrict__ lib, const CandRecIM *__restrict__ plan,
    adh_scoring_config_t cfg, int32_t n_iso_cols, unsigned char *__restrict__ scratch, DevOut out,
    Caps caps) {
    using namespace gather_im;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned n_touched;  // dense mode: fragment cells that received their first event
    __shared__ unsigned n_list;     // compact mode: entries in `list`
    __shared__ ImEntry list[ADH_IM_LIST_CAP];
    float *l_int = reinterpret_cast<float *>(smem);
    float *l_mz = l_int + caps.n_lib;
    int *l_rank = reinterpret_cast<int *>(l_mz + caps.n_lib);
    int *l_ok = l_rank + caps.n_lib;
    float *w_mz = reinterpret_cast<float *>(l_ok + caps.n_lib);
    int *t_lo = reinterpret_cast<int *>(w_mz + caps.k + caps.i);
    int *t_hi = t_lo + caps.k + caps.i;

    const int lane = threadIdx.x;
    const CandRecIM &r = plan[blockIdx.x];
    if (r.flags & ADH_FLAG_SKIP) return;
    const uint32_t row = r.row;
    if (lane == 0) {
        out.precursor_idx[row] = r.precursor_idx;  // candidate.py:175-176
        out.rank[row] = r.rank;
        n_touched = 0u;
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
    for (int w = lane; w < K + I; w += ADH_WAVE) {
        const bool prec = w >= K;
        const int slot = prec ? caps.k + (w - K) : w;
        float mzq = w_mz[slot];
        float tol = prec ? cfg.precursor_mz_tolerance : cfg.fragment_mz_tolerance;
        float t = tol * mzq;
        float q = t / 1000000.0f;
        double lo = (double)(mzq - q), hi = (double)(mzq + q);
        int a = 0, b = (int)run.n_tof;
        while (a < b) {
            int m = (a + b) >> 1;
            if (run.mz[m] < lo) a = m + 1; else b = m;
        }
        t_lo[slot] = a;
        b = (int)run.n_tof;
        while (a < b) {
            int m = (a + b) >> 1;
            if (run.mz[m] < hi) a = m + 1; else b = m;
        }
        t_hi[slot] = a;
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
    if (!dense) {
        // ---- compact mode: the same (window, cycle) tasks, one per lane, but a task keeps the few cells
        // it touches in registers (a fragment window sees ~0.4 events per cycle) and appends them to an
        // LDS list when it is done: no tile, no zero fill (116 KB per candidate at 38 scans x 29 cycles),
        // no read-modify-write traffic.  A cell belongs to exactly one task and receives its events in
        // (TOF, push) order: the reference's order.  The (small) precursor tile stays dense.
        for (int c = lane; c < n_pc; c += ADH_WAVE) pcells[c] = make_float2(0.0f, 0.0f);
        if (lane == 0) n_list = 0u;
        __syncthreads();
        bool over = false;
        for (int t = lane; t < (K + I) * F; t += ADH_WAVE) {
            const int w = t / F, f = t - w * F;
            const bool prec = w >= K;
            const int slot = prec ? caps.k + (w - K) : w;
            const int j = prec ? (w - K) : w;
            const double q_lo = prec ? -1.0 : fq_lo, q_hi = prec ? -1.0 : fq_hi;
            const int n_o = prec ? Op : O;
            const uint16_t *obs = prec ? r.ms1_obs : r.obs;
            const int frame_lo = (c0 + f) * L + z;
            const uint32_t push_lo = (uint32_t)frame_lo * (uint32_t)S_max;
            const uint32_t push_hi = (uint32_t)(frame_lo + L) * (uint32_t)S_max;
            uint32_t ec[ADH_IM_TASK_CAP];
            float ex[ADH_IM_TASK_CAP], ey[ADH_IM_TASK_CAP];
            int ne = 0;
            for (int tof = t_lo[slot]; tof < t_hi[slot]; ++tof) {
                const double measured = run.mz[tof];
                const int64_t b = run.tof_indptr[tof + 1];
                int64_t lo = run.tof_indptr[tof], hi = b;
                while (lo < hi) {
                    int64_t m = (lo + hi) >> 1;
                    if (run.push[m] < push_lo) lo = m + 1; else hi = m;
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
                    const int64_t ni = run.inten[idx];
                    ++hits;
                    if (prec) {
                        float2 v = pcells[cell];
                        fold(v.x, v.y, ni, measured);
                        pcells[cell] = v;
                        continue;
                    }
                    int e = -1;
#pragma unroll
                    for (int q = 0; q < ADH_IM_TASK_CAP; ++q)
                        if (q < ne && ec[q] == (uint32_t)cell) e = q;
                    if (e < 0) {
                        if (ne == ADH_IM_TASK_CAP) {
                            over = true;
                            break;
                        }
                        e = ne++;
#pragma unroll
                        for (int q = 0; q < ADH_IM_TASK_CAP; ++q)
                            if (q == e) ec[q] = (uint32_t)cell, ex[q] = 0.0f, ey[q] = 0.0f;
                    }
#pragma unroll
                    for (int q = 0; q < ADH_IM_TASK_CAP; ++q)
                        if (q == e) fold(ex[q], ey[q], ni, measured);
                }
                if (over) break;
            }
            if (ne > 0 && !over) {
                const unsigned base = atomicAdd(&n_list, (unsigned)ne);
                if (base + (unsigned)ne > ADH_IM_LIST_CAP) {
                    over = true;
                } else {
#pragma unroll
                    for (int q = 0; q < ADH_IM_TASK_CAP; ++q)
                        if (q < ne) {
                            ImEntry en;
                            en.cell = ec[q];
                            en.x = ex[q];
                            en.y = ey[q];
                            list[base + q] = en;
                        }
                }
            }
        }
        dense = __ballot(over) != 0ull;  // too many non-zero cells: this candidate takes the dense path below
        __syncthreads();
        if (!dense) {
            // sort the entries by cell (bitonic, padded with the largest key) and write the list out
            const int n_ent = (int)n_list;
            int n_p = ADH_WAVE;
            while (n_p < n_ent) n_p <<= 1;
            for (int e = n_ent + lane; e < n_p; e += ADH_WAVE) list[e].cell = 0xFFFFFFFFu;
            __syncthreads();
            for (int k = 2; k <= n_p; k <<= 1)
                for (int jj = k >> 1; jj > 0; jj >>= 1) {
                    for (int e = lane; e < n_p; e += ADH_WAVE) {
                        const int q = e ^ jj;
                        if (q > e) {
                            const ImEntry ea = list[e], eb = list[q];
                            if ((ea.cell > eb.cell) == ((e & k) == 0)) {
                                list[e] = eb;
                                list[q] = ea;
                            }
                        }
                    }
                    __syncthreads();
                }
            ImEntry *out_list = reinterpret_cast<ImEntry *>(block + adh_im_touch_off(r.k_cap, O, S, F, I, Op));
            for (int e = lane; e < n_ent; e += ADH_WAVE) out_list[e] = list[e];
            for (int off = 32; off > 0; off >>= 1) hits += __shfl_xor(hits, off);
            if (lane == 0) {
                header[0] = (uint32_t)K;
                header[1] = hits;
                header[2] = (uint32_t)n_ent;
                header[3] = ADH_IM_MODE_COMPACT;
            }
            return;
        }
        hits = 0;
    }

    // ---- dense mode: zero the tile, (window, cycle) tasks, list of the touched fragment cells
    for (int c = lane; c < n_fc; c += ADH_WAVE) fcells[c] = make_float2(0.0f, 0.0f);
    for (int c = lane; c < n_pc; c += ADH_WAVE) pcells[c] = make_float2(0.0f, 0.0f);
    __syncthreads();
    uint32_t *touched = reinterpret_cast<uint32_t *>(block + adh_im_touch_off(r.k_cap, O, S, F, I, Op));
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
            int64_t a = run.tof_indptr[tof];
            const int64_t b = run.tof_indptr[tof + 1];
            int64_t lo = a, hi = b;
            while (lo < hi) {
                int64_t m = (lo + hi) >> 1;
                if (run.push[m] < push_lo) lo = m + 1; else hi = m;
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
                if (!prec && v.x == 0.0f && v.y == 0.0f) {  // pristine: after any event the m/z plane is > 0
                    const unsigned pos = atomicAdd(&n_touched, 1u);
                    if (pos < ADH_IM_TOUCH_CAP) touched[pos] = (uint32_t)cell;
                }
                fold(v.x, v.y, (int64_t)run.inten[idx], measured);
                cells[cell] = v;
                ++hits;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) hits += __shfl_xor(hits, off);
    __syncthreads();
    if (lane == 0) {
        header[0] = (uint32_t)K;
        header[1] = hits;
        header[2] = n_touched <= ADH_IM_TOUCH_CAP ? n_touched : ADH_IM_TOUCH_OVERFLOW;
        header[3] = ADH_IM_MODE_DENSE;
    }
}
