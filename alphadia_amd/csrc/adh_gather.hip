// adh_gather.hip - kernel 1 of the scoring path: fragment selection + XIC gather.
//
// One 64-lane wavefront per candidate.  Replaces, for every candidate,
//   FragmentContainer.slice / filter_by_cardinality / filter_top_k / sort_by_mz
//                                 alphadia/search/jitclasses/fragment_container.py:56-102
//   AlphaRawJIT.get_dense (x2)    alphadia/search/jitclasses/alpharaw_jit.py:208-337
//   MS1 observation collapse      alphadia/search/scoring/containers/candidate.py:248-269
//
// This kernel is the latency-/HBM-bound half of the path, so it is kept free of
// large LDS tiles (high occupancy) and is written for memory-level parallelism:
//   * the reference walks the K sorted m/z windows of a spectrum with one monotone
//     cursor and a binary search per window (alpharaw_jit.py:290-297).  The cursor
//     after window k is "first peak with m/z > max(hi_0..hi_k)", so every
//     (fragment, observation, cycle) cell is independent: start at the first peak
//     with m/z >= lo_k and m/z > max(hi_j, j < k)
//   * the window start comes from a per-spectrum m/z bucket table (one 4-byte load
//     with an absolute peak offset) followed by a short forward scan; peaks are
//     (m/z, intensity) pairs so a hit costs no extra dependent load
//   * each lane keeps four cells in flight: the four table loads are issued
//     together, then the four first-peak loads, then the scans are resolved
//   * lanes that are adjacent in k read the same spectrum (same sectors)
// Results (selected fragments + the two-channel tile) go to a per-candidate scratch
// block in HBM that the feature kernel reads back coalesced.
#include "adh_device.h"

namespace gather {

constexpr double ISOTOPE_DELTA = 1.0033548350700006;  // candidate.py:160
constexpr int UNROLL = 4;

struct Cell {
    uint32_t idx, pe;
    float2 p;
    float lo, hi, excl, first_mz;
    bool live;
};

// accumulate one window starting at (idx, p): alpharaw_jit.py:299-335, absolute_masses=True
__device__ __forceinline__ void resolve(const DevRun &run, Cell &c, float &acc_i, float &acc_m,
                                        uint32_t &hits) {
    acc_i = 0.0f;
    acc_m = 0.0f;
    if (!c.live) return;
    uint32_t idx = c.idx;
    float2 p = c.p;
    while (idx < c.pe && !(p.x >= c.lo && p.x > c.excl)) {
        ++idx;
        if (idx < c.pe) p = run.peaks[idx];
    }
    while (idx < c.pe && p.x <= c.hi) {
        float ni = p.y;
        ni = ((double)ni > 1e-26) ? ni : ni * 0.0f;
        float a = acc_m * acc_i;
        float b = ni * p.x;
        float n32 = a + b;
        float d32 = acc_i + ni;
        acc_m = (float)(((double)n32 + 1e-36) / ((double)d32 + 1e-36));
        acc_i = d32;
        ++hits;
        ++idx;
        if (idx < c.pe) p = run.peaks[idx];
    }
}

__device__ __forceinline__ void issue_tab(const DevRun &run, Cell &c, int64_t spec) {
    const uint2 *t = run.tab + spec * (int64_t)(run.n_buckets + 2);
    const int b = adh_bucket_of(c.lo, run.bucket_min, run.bucket_inv_width, run.n_buckets);
    const uint2 e = t[b];
    c.idx = e.x;
    c.first_mz = __uint_as_float(e.y);  // +inf when no peak at/after this bucket
    c.pe = t[run.n_buckets + 1].x;
}

__device__ __forceinline__ void issue_peak(const DevRun &run, Cell &c) {
    c.p = make_float2(INFINITY, 0.0f);
    // the table already knows the m/z of peaks[idx]: beyond the window -> nothing to gather,
    // and no second sector has to be fetched
    if (c.first_mz > c.hi) c.live = false;
    if (c.live && c.idx < c.pe) c.p = run.peaks[c.idx];
}

}  // namespace gather

size_t adh_gather_lds_bytes(const Caps &c) {
    size_t b = (size_t)c.n_lib * (4 + 4 + 4 + 4);  // l_int, l_mz, l_rank, l_ok (as int)
    b += (size_t)(c.k + c.i) * 3 * 4;               // lo, hi, excl for fragments and isotopes
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
    float *w_lo = reinterpret_cast<float *>(l_ok + caps.n_lib);
    float *w_hi = w_lo + caps.k + caps.i;
    float *w_ex = w_hi + caps.k + caps.i;

    const int lane = threadIdx.x;
    const CandRec &r = plan[blockIdx.x];
    if (r.flags & ADH_FLAG_SKIP) return;
    const uint32_t row = r.row;
    if (lane == 0) {
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
    if (K <= 3 || F <= 0 || O <= 0) {  // candidate.py:190,230 / no overlapping window (:323)
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
        sel[slot] = lib[frag_start + a];
        // mass_range (jitclasses/utils.py:15-20): float32 throughout
        float t = cfg.fragment_mz_tolerance * ma;
        float q = t / 1000000.0f;
        w_lo[slot] = ma - q;
        w_hi[slot] = ma + q;
    }
    // isotope m/z (candidate.py:151-163) and their windows
    const int I = min(n_iso_cols, (int)cfg.top_k_isotopes);
    if (lane < I) {
        double off = (double)lane * ISOTOPE_DELTA / (double)r.charge;
        float mzq = (float)off + r.precursor_mz;
        float t = cfg.precursor_mz_tolerance * mzq;
        float q = t / 1000000.0f;
        w_lo[caps.k + lane] = mzq - q;
        w_hi[caps.k + lane] = mzq + q;
    }
    __syncthreads();
    if (lane == 0) {
        float e = -INFINITY;
        for (int k = 0; k < K; ++k) {
            w_ex[k] = e;
            e = fmaxf(e, w_hi[k]);
        }
        e = -INFINITY;
        for (int i = 0; i < I; ++i) {
            w_ex[caps.k + i] = e;
            e = fmaxf(e, w_hi[caps.k + i]);
        }
    }
    __syncthreads();

    // ---- fragment cells, index ((o * F + f) * K + k): k fastest
    uint32_t hits = 0;
    float2 *fcells = reinterpret_cast<float2 *>(block + adh_scratch_frag_off(r.k_cap));
    const int n_fc = K * O * F;
    for (int base = 0; base < n_fc; base += ADH_WAVE * UNROLL) {
        Cell cell[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            int c = base + u * ADH_WAVE + lane;
            cell[u].live = c < n_fc;
            int cc = cell[u].live ? c : 0;
            int k = cc % K, of = cc / K;
            int o = of / F, f = of - o * F;
            cell[u].lo = w_lo[k];
            cell[u].hi = w_hi[k];
            cell[u].excl = w_ex[k];
            int64_t spec = (int64_t)r.obs[o] + (int64_t)(c0 + f) * L;
            issue_tab(run, cell[u], spec);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) issue_peak(run, cell[u]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            float ai, am;
            resolve(run, cell[u], ai, am, hits);
            int c = base + u * ADH_WAVE + lane;
            if (c < n_fc) fcells[c] = make_float2(ai, am);
        }
    }
    // ---- precursor cells (i * F + f), MS1 observations collapsed (candidate.py:248-269)
    float2 *pcells = reinterpret_cast<float2 *>(block + adh_scratch_prec_off(r.k_cap, O, F));
    const int n_pc = I * F;
    for (int c = lane; c < n_pc; c += ADH_WAVE) {
        int i = c / F, f = c - i * F;
        float acc = 0.0f;
        double sum = 0.0;
        int count = 0;
        for (int j = 0; j < run.n_ms1_obs; ++j) {
            Cell cell;
            cell.live = true;
            cell.lo = w_lo[caps.k + i];
            cell.hi = w_hi[caps.k + i];
            cell.excl = w_ex[caps.k + i];
            int64_t spec = (int64_t)run.ms1_obs[j] + (int64_t)(c0 + f) * L;
            issue_tab(run, cell, spec);
            issue_peak(run, cell);
            float ai, am;
            resolve(run, cell, ai, am, hits);
            acc += ai;
            sum += (double)am;
            count += am > 0.0f;
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
// (m/z, intensity) -> float2 pairs
__global__ void adh_interleave_kernel(const float *__restrict__ mz, const float *__restrict__ inten,
                                      int64_t n, float2 *__restrict__ peaks) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) peaks[i] = make_float2(mz[i], inten[i]);
}

// bucket table of one spectrum per block: tab[b] = absolute index of the first peak whose
// bucket is >= b (b = 0..nb), tab[nb + 1] = end of the spectrum
__global__ void adh_bucket_build_kernel(const float2 *__restrict__ peaks,
                                        const int64_t *__restrict__ pstart,
                                        const int64_t *__restrict__ pstop, int64_t n_spectra,
                                        uint2 *__restrict__ tab, int nb, float bmin, float binv) {
    int64_t spec = blockIdx.x;
    if (spec >= n_spectra) return;
    const int64_t ps = pstart[spec], pe = pstop[spec];
    uint2 *t = tab + spec * (int64_t)(nb + 2);
    const int64_t n = pe - ps;
    for (int64_t j = threadIdx.x; j <= n; j += blockDim.x) {
        int b_prev = (j == 0) ? -1 : adh_bucket_of(peaks[ps + j - 1].x, bmin, binv, nb);
        int b_cur = (j == n) ? nb : adh_bucket_of(peaks[ps + j].x, bmin, binv, nb);
        const float mzj = (j == n) ? INFINITY : peaks[ps + j].x;
        for (int b = b_prev + 1; b <= b_cur; ++b)
            t[b] = make_uint2((uint32_t)(ps + j), __float_as_uint(mzj));
    }
    if (threadIdx.x == 0) t[nb + 1] = make_uint2((uint32_t)pe, __float_as_uint(INFINITY));
}
