// adh_gather.hip - kernel 1 of the scoring path: fragment selection + XIC gather.
//
// One 64-lane wavefront per candidate.  Replaces, for every candidate,
//   FragmentContainer.slice / filter_by_cardinality / filter_top_k / sort_by_mz
//                                 alphadia/search/jitclasses/fragment_container.py:56-102
//   AlphaRawJIT.get_dense (x2)    alphadia/search/jitclasses/alpharaw_jit.py:208-337
//   MS1 observation collapse      alphadia/search/scoring/containers/candidate.py:248-269
//
// The reference answers "which peaks of spectrum s fall into m/z window k" with one binary
// search per (window, spectrum): a candidate of F cycles costs F random probes per
// fragment.  A chromatogram is contiguous in time, so the run is staged TRANSPOSED:
// peaks are sorted by (cycle row, block of B consecutive cycles, m/z bin, cycle, m/z), the
// m/z bin being the upper bits of the float32 pattern (exactly monotone, ~30-60 ppm
// wide).  The XIC of one fragment over F cycles is then one or two short contiguous
// runs of 8-byte entries, found through one table lookup each:
//   * the reference's monotone cursor (alpharaw_jit.py:290-297) makes window k start at
//     the first peak with m/z >= lo_k and m/z > max(hi_j, j < k); every peak is tested
//     against exactly these float32 bounds, so the same peaks are selected
//   * the entries of a cell (same cycle) are visited in ascending m/z (bins ascending,
//     m/z ascending inside a bin, equal m/z in input order), so the running
//     intensity-weighted m/z (alpharaw_jit.py:299-335) is reproduced bit for bit
//   * one lane owns one (window, observation, block) task; its cells are private, so
//     continuing a cell across bins is a plain load of the cell's state
// Results (selected fragments + the two-channel tile) go to a per-candidate scratch
// block in HBM that the feature kernel reads back coalesced.
#include "adh_device.h"

namespace gather {

constexpr double ISOTOPE_DELTA = 1.0033548350700006;  // candidate.py:160

struct Window {
    float lo, hi, excl;
    int b_lo, b_hi;  // bin range relative to run.bin0; b_hi < b_lo: nothing can match
};

__device__ __forceinline__ void bins_of(const DevRun &run, Window &w) {
    const int nb = run.n_bins;
    int lo = 0, hi = nb - 1;
    if (w.lo > run.mz_min) lo = (int)(__float_as_uint(w.lo) >> ADH_BIN_SHIFT) - run.bin0;
    if (w.hi < run.mz_max) hi = (int)(__float_as_uint(w.hi) >> ADH_BIN_SHIFT) - run.bin0;
    if (!(w.hi >= run.mz_min) || !(w.lo <= run.mz_max)) hi = -1, lo = 0;  // also NaN windows
    w.b_lo = min(max(lo, 0), nb);
    w.b_hi = min(hi, nb - 1);
}

// one step of the running sums, alpharaw_jit.py:299-335 with absolute_masses=True
__device__ __forceinline__ void fold(float &acc_i, float &acc_m, float mz, float ni) {
    ni = ((double)ni > 1e-26) ? ni : ni * 0.0f;
    float a = acc_m * acc_i;
    float b = ni * mz;
    float n32 = a + b;
    float d32 = acc_i + ni;
    // (float)(((double)n32 + 1e-36) / ((double)d32 + 1e-36)).  Above 1e-20 the float64 sums are the operands
    // themselves (1e-36 is below half an ulp of their 53-bit form), and a float64 quotient of two float32
    // values rounded to float32 is the correctly rounded float32 quotient (53 >= 2 * 24 + 2): one float32
    // division (IEEE, hipcc's default) instead of a float64 one.
    if (n32 > 1e-20f && d32 > 1e-20f)
        acc_m = n32 / d32;
    else
        acc_m = (float)(((double)n32 + 1e-36) / ((double)d32 + 1e-36));
    acc_i = d32;
}

// Gather one (window, cycle row, block) task.  `cells` is indexed cells[f * stride] for
// f = cycle - c0; the cells must be zero on entry.
template <typename CellPtr>
__device__ __forceinline__ void gather_task(const DevRun &run, const Window &w, int row, int blk,
                                            int c0, int F, CellPtr cells, int stride,
                                            uint32_t &hits) {
    if (w.b_hi < w.b_lo) return;
    const int bs = run.block_shift;
    const int cyc_base = blk << bs;
    const int grp_base = (blk >> ADH_SUB_SHIFT) << (bs + ADH_SUB_SHIFT);
    const int f_lo = max(c0, cyc_base) - grp_base;              // group-relative cycle range inside the block
    const int f_hi = min(c0 + F, cyc_base + (1 << bs)) - grp_base;
    const uint32_t *t = adh_tab_row(run, row, blk) + (blk & (ADH_SUB - 1));
    const uint2 *ent = adh_group_entries(run, blk);
    for (int b = w.b_lo; b <= w.b_hi; ++b) {  // bin after bin: a cell keeps ascending m/z
        uint32_t idx = t[b * ADH_SUB];
        const uint32_t end = t[b * ADH_SUB + 1];
        int cur = -1;  // open cell (group-relative cycle), -1: none
        float acc_i = 0.0f, acc_m = 0.0f;
        while (idx < end) {
            // four entries in flight; the tail repeats the last one (harmless, skipped below)
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
                if (cyc != cur) {
                    if (cur >= 0) cells[(cur + grp_base - c0) * stride] = make_float2(acc_i, acc_m);
                    float2 v = make_float2(0.0f, 0.0f);
                    if (b > w.b_lo) v = cells[(cyc + grp_base - c0) * stride];  // continue from an earlier bin
                    acc_i = v.x;
                    acc_m = v.y;
                    cur = cyc;
                }
                fold(acc_i, acc_m, mz, __uint_as_float(e[u].y));
                ++hits;
            }
            idx += 4;
        }
        if (cur >= 0) cells[(cur + grp_base - c0) * stride] = make_float2(acc_i, acc_m);
    }
}

}  // namespace gather

size_t adh_gather_lds_bytes(const Caps &c, int n_ms1_obs) {
    size_t b = (size_t)c.n_lib * (4 + 4 + 4 + 4);        // l_int, l_mz, l_rank, l_ok (as int)
    b += (size_t)(c.k + c.i) * sizeof(gather::Window);     // windows of fragments and isotopes
    b = (b + 7) / 8 * 8;
    b += (size_t)c.i * (size_t)std::max(n_ms1_obs, 1) * (size_t)c.f * sizeof(float2);  // raw MS1 cells
    return (b + 15) / 16 * 16;
}

__global__ __launch_bounds__(ADH_WAVE) void adh_gather_kernel(
    DevRun run, const LibRec *__restrict__ lib, const CandRec *__restrict__ plan,
    adh_scoring_config_t cfg, int32_t n_iso_cols, unsigned char *__restrict__ scratch, DevOut out,
    Caps caps) {
    using namespace gather;
    extern __shared__ __align__(16) unsigned char smem[];
    float *l_int = reinterpret_cast<float *>(smem);
    float *l_mz = l_int + caps.n_lib;
    int *l_rank = reinterpret_cast<int *>(l_mz + caps.n_lib);
    int *l_ok = l_rank + caps.n_lib;
    Window *win = reinterpret_cast<Window *>(l_ok + caps.n_lib);
    float2 *raw1 = reinterpret_cast<float2 *>(
        smem + ((size_t)caps.n_lib * 16 + (size_t)(caps.k + caps.i) * sizeof(Window) + 7) / 8 * 8);

    const int lane = threadIdx.x;
    const CandRec &r = plan[blockIdx.x];
    if (r.flags & ADH_FLAG_SKIP) return;
    const uint32_t row = r.row;
    if (lane == 0 && out.precursor_idx) {
        out.precursor_idx[row] = r.precursor_idx;  // candidate.py:175-176
        out.rank[row] = r.rank;
    }
    unsigned char *block = scratch + r.scratch_off;
    uint32_t *header = reinterpret_cast<uint32_t *>(block);
    LibRec *sel = reinterpret_cast<LibRec *>(block + 32);

    // ---- zero the tile right away: fragment cells ((o * F + f) * K + k) in scratch - K <= k_cap of them per
    // (o, f) - and the raw MS1 cells in LDS.  The stores drain while the library slice is loaded and
    // ranked, so the barrier in front of the gather tasks finds them done.
    float2 *fcells = reinterpret_cast<float2 *>(block + adh_scratch_frag_off(r.k_cap));
    const int M1 = run.n_ms1_obs;
    {
        const int Lz = run.cycle_len;
        const int Fz = r.frame_stop / Lz - r.frame_start / Lz;
        const int n_z = (int)r.k_cap * (int)r.n_obs * max(Fz, 0);
        for (int c = lane; c < n_z; c += ADH_WAVE) fcells[c] = make_float2(0.0f, 0.0f);
        const int n_r = min(n_iso_cols, (int)cfg.top_k_isotopes) * M1 * max(Fz, 0);
        for (int c = lane; c < n_r; c += ADH_WAVE) raw1[c] = make_float2(0.0f, 0.0f);
    }

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
                rk += (ib > ia) || (ib == ia && b > a);  // position in argsort()[::-1]
            }
            if (rk >= (int)cfg.top_k_fragments) rk = -1;
        }
        l_rank[a] = rk;
    }
    __syncthreads();
    int K = 0;
    for (int a = 0; a < n_lib; ++a) K += l_rank[a] >= 0;
    const int L = run.cycle_len;
    const int c0 = r.frame_start / L, c1 = r.frame_stop / L;
    const int F = c1 - c0;
    const int O = r.n_obs;
    if ((K <= 3 && caps.stop_phase != ADH_DEBUG_DENSE) || K <= 0 || F <= 0 || O <= 0) {  // candidate.py:190,230 / no overlapping window (:323)
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
            slot += (mb < ma) || (mb == ma && rb < ra);  // stable argsort(mz) of the top-k list
        }
        LibRec pick = lib[frag_start + a];
        pick.pad0 = (uint8_t)(a & 0xFF);  // position inside the library slice (adh_output_t.fragment_lib_slot)
        pick.pad1 = (uint8_t)(a >> 8);
        sel[slot] = pick;
        // mass_range (jitclasses/utils.py:15-20): float32 throughout
        float t = cfg.fragment_mz_tolerance * ma;
        float q = t / 1000000.0f;
        win[slot].lo = ma - q;
        win[slot].hi = ma + q;
    }
    // isotope m/z (candidate.py:151-163) and their windows
    const int I = min(n_iso_cols, (int)cfg.top_k_isotopes);
    if (lane < I) {
        double off = (double)lane * ISOTOPE_DELTA / (double)r.charge;
        float mzq = (float)off + r.precursor_mz;
        float t = cfg.precursor_mz_tolerance * mzq;
        float q = t / 1000000.0f;
        win[caps.k + lane].lo = mzq - q;
        win[caps.k + lane].hi = mzq + q;
    }
    __syncthreads();
    if (lane == 0) {
        float e = -INFINITY;
        for (int k = 0; k < K; ++k) {
            win[k].excl = e;
            e = fmaxf(e, win[k].hi);
        }
        e = -INFINITY;
        for (int i = 0; i < I; ++i) {
            win[caps.k + i].excl = e;
            e = fmaxf(e, win[caps.k + i].hi);
        }
    }
    for (int w = lane; w < K + I; w += ADH_WAVE) bins_of(run, win[w < K ? w : caps.k + (w - K)]);
    if (caps.stop_phase == 1) {  // developer ablation switches (ADH_DEBUG_GATHER)
        if (lane == 0) header[0] = 0;
        return;
    }

    __syncthreads();  // (the tile was zeroed at the top)
    if (caps.stop_phase == 2) {
        if (lane == 0) header[0] = 0;
        return;
    }

    // ---- (window, observation, block) tasks
    uint32_t hits = 0;
    const int bs = run.block_shift;
    const int blk0 = c0 >> bs;
    const int n_blk = ((c0 + F - 1) >> bs) - blk0 + 1;
    const int n_ft = K * O * n_blk;
    const int n_tasks = n_ft + I * M1 * n_blk;
    for (int t = lane; t < n_tasks; t += ADH_WAVE) {
        if (t < n_ft) {
            const int bi = t % n_blk, ko = t / n_blk;
            const int k = ko % K, o = ko / K;
            gather_task(run, win[k], (int)r.obs[o], blk0 + bi, c0, F, fcells + (o * F) * K + k, K, hits);
        } else {
            const int u = t - n_ft;
            const int bi = u % n_blk, ij = u / n_blk;
            const int i = ij / M1, j = ij - i * M1;
            gather_task(run, win[caps.k + i], run.ms1_obs[j], blk0 + bi, c0, F, raw1 + (i * M1 + j) * F, 1, hits);
        }
    }
    __syncthreads();
    // ---- precursor cells (i * F + f): MS1 observations collapsed (candidate.py:248-269)
    float2 *pcells = reinterpret_cast<float2 *>(block + adh_scratch_prec_off(r.k_cap, O, F));
    const int n_pc = I * F;
    for (int c = lane; c < n_pc; c += ADH_WAVE) {
        int i = c / F, f = c - i * F;
        float acc = 0.0f;
        double sum = 0.0;
        int count = 0;
        for (int j = 0; j < M1; ++j) {
            const float2 v = raw1[(i * M1 + j) * F + f];
            acc += v.x;
            sum += (double)v.y;
            count += v.y > 0.0f;
        }
        pcells[c] = make_float2(acc, (float)(sum / ((double)count + 1e-6)));
    }
    for (int off = 32; off > 0; off >>= 1) hits += __shfl_xor(hits, off);
    if (lane == 0) {
        header[0] = (uint32_t)K;
        header[1] = hits;
    }
}

// ------------------------------------------------------------------ staging kernels
// sort key of every peak: high word = ((group * L + row) * n_bins + bin), low word =
// (cycle inside the group << ADH_BIN_SHIFT) | low m/z bits.  One workgroup per spectrum of the slab
// [spec0, spec0 + n_spec): peak j of the run is item j - peak0 of the slab.
__global__ void adh_peak_key_kernel(const float *__restrict__ mz, const int64_t *__restrict__ pstart,
                                    const int64_t *__restrict__ pstop, int64_t spec0, int64_t n_spec, int64_t peak0, int L,
                                    int block_shift, int bin0, int n_bins,
                                    uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                    int *__restrict__ bad) {
    if ((int64_t)blockIdx.x >= n_spec) return;
    const int64_t spec = spec0 + blockIdx.x;
    const int64_t cyc = spec / L;
    const int row = (int)(spec - cyc * L);
    const int gs = block_shift + ADH_SUB_SHIFT;
    const int64_t grp = cyc >> gs;
    const uint32_t cin = (uint32_t)(cyc - (grp << gs));
    const uint64_t seg = (uint64_t)(grp * L + row) * (uint64_t)n_bins;
    const int64_t ps = pstart[blockIdx.x] - peak0, pe = pstop[blockIdx.x] - peak0;
    for (int64_t j = ps + threadIdx.x; j < pe; j += blockDim.x) {
        const uint32_t bits = __float_as_uint(mz[j]);
        const int b = (int)(bits >> ADH_BIN_SHIFT) - bin0;
        if (b < 0 || b >= n_bins) {  // outside [first, last] peak of some spectrum: not sorted
            *bad = 1;
            continue;
        }
        keys[j] = ((seg + (uint64_t)b) << 32) |
                  (uint64_t)((cin << ADH_BIN_SHIFT) | (bits & ((1u << ADH_BIN_SHIFT) - 1u)));
        vals[j] = (uint32_t)j;
    }
}

// sorted keys of one slab (whole groups of blocks) -> 8-byte entries (low key word, intensity) + the slab's part of
// the bin table.  With the fine key g = (group, row, bin) * ADH_SUB + block inside the group, the word of g is the
// index of the first entry whose fine key is >= g, counted from the first entry of g's group (grp_entry0); the
// word sits at g + group (every group's segment ends with a spare word: adh_tab_row).  The slab's entries start at
// entry0 of the run; its fine keys are [g_first, g_last].
__global__ void adh_entries_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                   const float *__restrict__ inten, int64_t n, int64_t entry0, uint2 *__restrict__ entries,
                                   uint32_t *__restrict__ tab, int64_t g_first, int64_t g_last, int64_t words_per_group,
                                   const int64_t *__restrict__ grp_entry0, int block_shift) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto fine = [block_shift](uint64_t key) -> int64_t {
        const uint32_t cin = (uint32_t)key >> ADH_BIN_SHIFT;
        return (int64_t)((key >> 32) << ADH_SUB_SHIFT) + (int64_t)(cin >> block_shift);
    };
    for (; i <= n; i += stride) {
        const int64_t g_prev = (i == 0) ? g_first - 1 : fine(keys[i - 1]);
        const int64_t g_cur = (i == n) ? g_last : fine(keys[i]);
        for (int64_t g = g_prev + 1; g <= g_cur; ++g) {
            const int64_t grp = g / words_per_group;
            tab[g + grp] = (uint32_t)(entry0 + i - grp_entry0[grp]);
        }
        if (i < n) entries[entry0 + i] = make_uint2((uint32_t)keys[i], __float_as_uint(inten[vals[i]]));
    }
}

// the spare word that ends every group's segment of the table: the group's entry count
__global__ void adh_tab_spare_kernel(uint32_t *__restrict__ tab, int64_t n_groups, int64_t words_per_group,
                                     const int64_t *__restrict__ grp_entry0) {
    const int64_t grp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (grp < n_groups) tab[(grp + 1) * words_per_group + grp] = (uint32_t)(grp_entry0[grp + 1] - grp_entry0[grp]);
}
