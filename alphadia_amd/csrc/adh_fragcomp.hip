// adh_fragcomp.hip - fragment competition on gfx950.
//
// Replaces `_compete_for_fragments` + `_get_fragment_overlap`
// (alphadia/fragcomp/fragcomp.py:19-143).  The reference gives every DIA window to a thread
// and walks the window with two nested sequential loops:
//
//     for i (best proba first):  if valid[i]:
//         for j != i:            if valid[j] and |rt_i - rt_j| < 3 s and overlap(i, j) >= 3:  valid[j] = False
//
// The rule is greedy and order dependent, which is why round 1-3 kept `i` sequential (one
// workgroup per window, 34 000 barriers for a window of 17 000 PSMs, every `j` of the window
// visited for every `i`: O(n^2) per window on 60 of 256 CUs).  Round 4 separates what is
// parallel from what is ordered:
//
//   "i can remove j"  C(i -> j) = |rt_i - rt_j| < tol  and  #{(a, b): |mz_a - mz_b| / mz_a * 1e6 < ppm} >= 3
//   depends on the two PSMs only.  PSM i acts iff it is still valid when its turn comes:
//       active[i] = valid0[i] and no active i' < i with C(i' -> i)
//   and what the loops leave behind is
//       valid[j]  = active[j] and no active i (earlier OR later - the reference's j runs over the
//                   whole window) with C(i -> j).
//
//   1. every PSM's RT neighbourhood is a contiguous range of its window sorted by RT (the float32
//      subtraction is monotone, the exact predicate is what the binary searches evaluate): one radix
//      sort by (window, rt), two binary searches per PSM - ~14 neighbours instead of 17 000;
//   2. `adh_fc_edges_kernel`: 16 lanes per PSM j, a lane per neighbour, K x K m/z comparisons with the
//      reference's arithmetic -> the in-edges of j as one bit per neighbour; PSMs without an earlier
//      in-edge are active at once (nearly all of them);
//   3. `adh_fc_resolve_kernel`: the few PSMs with an earlier in-edge wait for those neighbours' states;
//      the lowest unresolved PSM can always be decided, so rounds are bounded by the longest chain of
//      conflicts (2-4 on search results), not by the window;
//   4. `adh_fc_final_kernel`: valid[j] from the states of j's in-neighbours.
//
// > 10 000 workgroups instead of <= 60, no barrier, identical survivors (reference KATs, 5 000-PSM
// golden, random tables against the oracle).  The one-workgroup-per-window kernel stays as
// `adh_fc_serial_kernel` for tables whose neighbour bitmap would not fit (every PSM at one RT).
#include "adh_device.h"

#define ADH_FC_THREADS 256
#define ADH_FC_MAXFRAG 512

__global__ __launch_bounds__(ADH_FC_THREADS) void adh_fc_serial_kernel(
    int64_t n_windows, const int64_t *window_start, const int64_t *window_stop, const float *rt,
    const int64_t *frag_start, const int64_t *frag_stop, const float *fragment_mz,
    double rt_tol_seconds, double mass_tol_ppm, uint8_t *valid) {
    __shared__ float s_mz[ADH_FC_MAXFRAG];
    const int64_t w = blockIdx.x;
    if (w >= n_windows) return;
    const int64_t p0 = window_start[w], p1 = window_stop[w];
    for (int64_t i = p0; i < p1; ++i) {
        if (!valid[i]) continue;  // uniform: every lane reads the same byte after the barrier
        const int64_t a0 = frag_start[i], a1 = frag_stop[i];
        const int na = (int)(a1 - a0);
        const bool in_lds = na <= ADH_FC_MAXFRAG;
        if (in_lds)
            for (int a = threadIdx.x; a < na; a += ADH_FC_THREADS) s_mz[a] = fragment_mz[a0 + a];
        __syncthreads();
        const float rt_i = rt[i];
        for (int64_t j = p0 + threadIdx.x; j < p1; j += ADH_FC_THREADS) {
            if (j == i || !valid[j]) continue;
            float delta_rt = fabsf(rt_i - rt[j]);
            if (!((double)delta_rt < rt_tol_seconds)) continue;
            const int64_t b0 = frag_start[j], b1 = frag_stop[j];
            int overlap = 0;
            for (int a = 0; a < na; ++a) {
                float ma = in_lds ? s_mz[a] : fragment_mz[a0 + a];
                for (int64_t b = b0; b < b1; ++b) {
                    float delta = fabsf(ma - fragment_mz[b]);
                    float rel = delta / ma;
                    double ppm = (double)rel * 1e6;
                    overlap += ppm < mass_tol_ppm;
                }
            }
            if (overlap >= 3) valid[j] = 0;
        }
        __syncthreads();
    }
}

namespace fragcomp {

constexpr int kGroup = 16;      // lanes per PSM row in the edge kernel
constexpr int kLdsFrag = 32;    // fragment m/z of the row's PSM kept in LDS (longer lists: global memory)
constexpr uint8_t kInactive = 0, kActive = 1, kUnknown = 2;

// window of every PSM (-1: in none, never touched - as in the reference, whose loops only see windows)
__global__ void window_of_kernel(int64_t n_windows, const int64_t *__restrict__ ws, const int64_t *__restrict__ we,
                                 int32_t *__restrict__ win_of) {
    const int64_t w = blockIdx.x;
    if (w >= n_windows) return;
    for (int64_t i = ws[w] + threadIdx.x; i < we[w]; i += blockDim.x) win_of[i] = (int32_t)w;
}

// first sorted position of every window: windows in id order, each as long as its row range
__global__ void segment_kernel(int64_t n_windows, const int64_t *__restrict__ ws, const int64_t *__restrict__ we,
                               int64_t *__restrict__ seg) {
    if (blockIdx.x || threadIdx.x) return;
    int64_t at = 0;
    for (int64_t w = 0; w < n_windows; ++w) {
        seg[w] = at;
        at += we[w] - ws[w];
    }
    seg[n_windows] = at;
}

// sort key (window, rt): float32 bits made monotone, every NaN last
__global__ void key_kernel(int64_t n, const int32_t *__restrict__ win_of, const float *__restrict__ rt,
                           uint64_t *__restrict__ key, uint32_t *__restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = rt[i];
    uint32_t b = __float_as_uint(v);
    b = (v != v) ? 0xFFFFFFFFu : ((b & 0x80000000u) ? ~b : (b | 0x80000000u));
    const uint32_t w = win_of[i] < 0 ? 0xFFFFFFFFu : (uint32_t)win_of[i];
    key[i] = ((uint64_t)w << 32) | b;
    idx[i] = (uint32_t)i;
}

__device__ __forceinline__ bool rt_close(float a, float b, double tol) {
    // fragcomp.py:126-127: delta_rt = abs(i_rt - j_rt) in the array's dtype, compared with a float64
    return (double)fabsf(a - b) < tol;
}

// neighbourhood of the PSM at sorted position p: [lo, lo + cnt) of its window's segment
__global__ void range_kernel(int64_t n_in, const uint32_t *__restrict__ rs_idx, const int32_t *__restrict__ win_of,
                             const int64_t *__restrict__ seg, const float *__restrict__ rt, double tol,
                             const uint8_t *__restrict__ valid0, uint32_t *__restrict__ lo_out,
                             uint32_t *__restrict__ cnt_out, uint64_t *__restrict__ words_out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_in) return;
    const uint32_t j = rs_idx[p];
    const int32_t w = win_of[j];
    const float r = rt[j];
    uint32_t lo = (uint32_t)p, cnt = 0;
    if (w >= 0 && valid0[j] && rt_close(r, r, tol)) {
        // the predicate holds at p and is monotone towards both ends of the segment
        int64_t a = seg[w], b = p;  // first position in [a, p] where it holds
        while (a < b) {
            const int64_t m = (a + b) >> 1;
            if (rt_close(rt[rs_idx[m]], r, tol)) b = m;
            else a = m + 1;
        }
        lo = (uint32_t)a;
        a = p;
        b = seg[w + 1] - 1;  // last position in [p, b] where it holds
        while (a < b) {
            const int64_t m = (a + b + 1) >> 1;
            if (rt_close(rt[rs_idx[m]], r, tol)) a = m;
            else b = m - 1;
        }
        cnt = (uint32_t)(a + 1) - lo;
    }
    lo_out[p] = lo;
    cnt_out[p] = cnt;
    words_out[p] = (cnt + kGroup - 1) / kGroup;  // 16-bit words of the row's bitmap
}

// overlap >= 3 ?  frag_mz_1 = the acting PSM's list (a), frag_mz_2 = the other's (b): fragcomp.py:43-48
//   ppm = |mz_a - mz_b| / mz_a * 1e6   (float32 quotient; the literal makes the product float64)
// `skip_above`: relative distance beyond which the exact test cannot pass (0 disables the shortcut).
template <bool LDS>
__device__ __forceinline__ bool removes(const float *__restrict__ mz_a, int na, const float *s_b,
                                        const float *__restrict__ g_b, int nb, double ppm_tol, float skip_above) {
    int overlap = 0;
    for (int a = 0; a < na; ++a) {
        const float ma = mz_a[a];
        const float cut = (ma > 0.0f) ? ma * skip_above : 0.0f;
        const bool fast = cut > 0.0f;
        for (int b = 0; b < nb; ++b) {
            const float mb = LDS ? s_b[b] : g_b[b];
            const float delta = fabsf(ma - mb);
            if (fast && delta > cut) continue;  // far above the tolerance: the exact test fails too
            const float rel = delta / ma;
            const double ppm = (double)rel * 1e6;
            overlap += ppm < ppm_tol;
        }
    }
    return overlap >= 3;
}

// in-edges of every PSM: bit q of row p = "the neighbour at sorted position lo + q can remove this PSM"
__global__ __launch_bounds__(256) void adh_fc_edges_kernel(
    int64_t n_in, const uint32_t *__restrict__ rs_idx, const uint32_t *__restrict__ lo_arr,
    const uint32_t *__restrict__ cnt_arr, const uint64_t *__restrict__ word_off, const int64_t *__restrict__ fs,
    const int64_t *__restrict__ fe, const float *__restrict__ mz, double ppm_tol, float skip_above,
    const uint8_t *__restrict__ valid0, uint16_t *__restrict__ bits, uint8_t *__restrict__ state,
    uint8_t *__restrict__ has_in, uint32_t *__restrict__ unknown, uint32_t *__restrict__ n_unknown) {
    __shared__ float s_mz[256 / kGroup][kLdsFrag];
    const int g = threadIdx.x / kGroup, l = threadIdx.x % kGroup;
    const int64_t p = (int64_t)blockIdx.x * (256 / kGroup) + g;
    if (p >= n_in) return;  // whole 16-lane groups leave together; there is no block barrier below
    const uint32_t j = rs_idx[p];
    const uint32_t cnt = cnt_arr[p];
    if (cnt == 0) {  // outside every window, invalid on entry, or a NaN rt: no neighbours at all
        if (l == 0) {
            state[j] = valid0[j] ? kActive : kInactive;
            has_in[j] = 0;
        }
        return;
    }
    const int64_t b0 = fs[j];
    const int nb = (int)(fe[j] - b0);
    const bool in_lds = nb <= kLdsFrag;
    if (in_lds)
        for (int b = l; b < nb; b += kGroup) s_mz[g][b] = mz[b0 + b];
    adh_wave_sync();  // a 16-lane group lives in one wavefront: its LDS writes precede its reads
    const uint32_t lo = lo_arr[p];
    const uint64_t off = word_off[p];
    const int sh = (threadIdx.x % 64) / kGroup * kGroup;
    bool any_earlier = false, any = false;
    for (uint32_t q0 = 0; q0 < cnt; q0 += kGroup) {
        const uint32_t q = q0 + l;
        bool bit = false;
        if (q < cnt) {
            const uint32_t i = rs_idx[lo + q];
            if (i != j && valid0[i]) {
                const int64_t a0 = fs[i];
                const int na = (int)(fe[i] - a0);
                bit = in_lds ? removes<true>(mz + a0, na, s_mz[g], nullptr, nb, ppm_tol, skip_above)
                             : removes<false>(mz + a0, na, nullptr, mz + b0, nb, ppm_tol, skip_above);
                any |= bit;
                any_earlier |= bit && i < j;
            }
        }
        const uint64_t m = __ballot(bit);
        if (l == 0) bits[off + q0 / kGroup] = (uint16_t)(m >> sh);
    }
    const uint64_t m_any = __ballot(any), m_early = __ballot(any_earlier);
    if (l == 0) {
        const bool in_any = (m_any >> sh) & 0xFFFFu, in_early = (m_early >> sh) & 0xFFFFu;
        has_in[j] = in_any ? 1 : 0;
        if (in_early) {
            state[j] = kUnknown;
            unknown[atomicAdd(n_unknown, 1u)] = (uint32_t)p;
        } else {
            state[j] = kActive;  // valid on entry and nobody earlier can remove it
        }
    }
}

// One round over the PSMs whose fate hangs on earlier ones.  A state, once written, is final, so
// reading a neighbour that another thread decides in the same round is harmless (either value is true).
__global__ void adh_fc_resolve_kernel(const uint32_t *__restrict__ unknown, uint32_t n_unknown,
                                      const uint32_t *__restrict__ rs_idx, const uint32_t *__restrict__ lo_arr,
                                      const uint32_t *__restrict__ cnt_arr, const uint64_t *__restrict__ word_off,
                                      const uint16_t *__restrict__ bits, uint8_t *state, uint32_t *__restrict__ pending) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_unknown) return;
    const uint32_t p = unknown[u];
    const uint32_t j = rs_idx[p];
    if (__atomic_load_n(&state[j], __ATOMIC_RELAXED) != kUnknown) return;
    const uint32_t lo = lo_arr[p], nw = (cnt_arr[p] + kGroup - 1) / kGroup;
    const uint64_t off = word_off[p];
    for (int sweep = 0; sweep < 4; ++sweep) {  // short chains settle inside one launch
        bool removed = false, wait = false;
        for (uint32_t wd = 0; wd < nw && !removed; ++wd) {
            uint32_t m = bits[off + wd];
            while (m) {
                const int b = __builtin_ctz(m);
                m &= m - 1;
                const uint32_t i = rs_idx[lo + wd * kGroup + b];
                if (i >= j) continue;
                const uint8_t s = __atomic_load_n(&state[i], __ATOMIC_RELAXED);
                if (s == kActive) {
                    removed = true;
                    break;
                }
                wait |= s == kUnknown;
            }
        }
        if (removed) {
            __atomic_store_n(&state[j], kInactive, __ATOMIC_RELAXED);
            return;
        }
        if (!wait) {
            __atomic_store_n(&state[j], kActive, __ATOMIC_RELAXED);
            return;
        }
    }
    atomicAdd(pending, 1u);
}

// valid[j] = it acted and no acting neighbour, earlier or later, removes it
__global__ void adh_fc_final_kernel(int64_t n_in, const uint32_t *__restrict__ rs_idx,
                                    const uint32_t *__restrict__ lo_arr, const uint32_t *__restrict__ cnt_arr,
                                    const uint64_t *__restrict__ word_off, const uint16_t *__restrict__ bits,
                                    const uint8_t *__restrict__ state, const uint8_t *__restrict__ has_in,
                                    const int32_t *__restrict__ win_of, uint8_t *__restrict__ valid) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_in) return;
    const uint32_t j = rs_idx[p];
    if (win_of[j] < 0) return;  // not in any window: untouched
    bool ok = state[j] == kActive;
    if (ok && has_in[j]) {
        const uint32_t lo = lo_arr[p], nw = (cnt_arr[p] + kGroup - 1) / kGroup;
        const uint64_t off = word_off[p];
        for (uint32_t wd = 0; wd < nw && ok; ++wd) {
            uint32_t m = bits[off + wd];
            while (m) {
                const int b = __builtin_ctz(m);
                m &= m - 1;
                if (state[rs_idx[lo + wd * kGroup + b]] == kActive) {
                    ok = false;
                    break;
                }
            }
        }
    }
    valid[j] = ok ? 1 : 0;
}

struct Tmp {
    std::vector<void *> ptrs;
    ~Tmp() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    hipError_t alloc(T **p, size_t count) {
        hipError_t e = hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

struct Stats {
    int64_t pairs = 0;       // (PSM, RT neighbour) pairs examined
    int64_t unknown = 0;     // PSMs with an earlier in-edge
    int32_t rounds = 0;      // resolve launches
    int32_t serial = 0;      // 1: the bitmap would not fit, one workgroup per window ran instead
    double kernel_ms = 0.0;  // HIP events around everything below, on `st`
};

#define FC_TRY(expr)                 \
    do {                             \
        hipError_t _e = (expr);      \
        if (_e != hipSuccess) return _e; \
    } while (0)

// Every pointer is a device buffer; `valid` is input (who takes part) and output.  Synchronises `st`
// (the round loop reads a counter back).
inline hipError_t compete(hipStream_t st, int64_t n_windows, const int64_t *d_ws, const int64_t *d_we, int64_t n_psm,
                          const float *d_rt, const int64_t *d_fs, const int64_t *d_fe, const float *d_mz,
                          double rt_tol, double ppm_tol, uint8_t *d_valid, Stats *stats) {
    Stats local;
    Stats &S = stats ? *stats : local;
    S = Stats();
    if (n_windows <= 0 || n_psm <= 0) return hipSuccess;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    FC_TRY(hipEventCreate(&e0));
    hipError_t err = hipEventCreate(&e1);
    if (err != hipSuccess) {
        (void)hipEventDestroy(e0);
        return err;
    }
    auto body = [&]() -> hipError_t {
        Tmp t;
        const unsigned nb256 = (unsigned)((n_psm + 255) / 256);
        bool serial = n_psm >= (int64_t)0xFFFFFFF0ll || getenv("ADH_FRAGCOMP_SERIAL") != nullptr;
        int32_t *win_of = nullptr;
        int64_t *seg = nullptr;
        uint64_t *key_in = nullptr, *key_out = nullptr, *words = nullptr, *word_off = nullptr;
        uint32_t *idx_in = nullptr, *rs_idx = nullptr, *lo = nullptr, *cnt = nullptr, *unknown = nullptr, *counters = nullptr;
        uint8_t *valid0 = nullptr, *state = nullptr, *has_in = nullptr;
        uint16_t *bits = nullptr;
        uint64_t total_words = 0;
        FC_TRY(hipEventRecord(e0, st));
        if (!serial) {
            FC_TRY(t.alloc(&win_of, n_psm));
            FC_TRY(t.alloc(&seg, n_windows + 1));
            FC_TRY(t.alloc(&key_in, n_psm));
            FC_TRY(t.alloc(&key_out, n_psm));
            FC_TRY(t.alloc(&idx_in, n_psm));
            FC_TRY(t.alloc(&rs_idx, n_psm));
            FC_TRY(t.alloc(&lo, n_psm));
            FC_TRY(t.alloc(&cnt, n_psm));
            FC_TRY(t.alloc(&words, n_psm + 1));
            FC_TRY(t.alloc(&word_off, n_psm + 1));
            FC_TRY(t.alloc(&valid0, n_psm));
            FC_TRY(t.alloc(&counters, 4));
            FC_TRY(hipMemsetAsync(win_of, 0xFF, (size_t)n_psm * 4, st));
            FC_TRY(hipMemsetAsync(counters, 0, 16, st));
            FC_TRY(hipMemcpyAsync(valid0, d_valid, (size_t)n_psm, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(window_of_kernel, dim3((unsigned)n_windows), dim3(256), 0, st, n_windows, d_ws, d_we, win_of);
            hipLaunchKernelGGL(segment_kernel, dim3(1), dim3(1), 0, st, n_windows, d_ws, d_we, seg);
            hipLaunchKernelGGL(key_kernel, dim3(nb256), dim3(256), 0, st, n_psm, win_of, d_rt, key_in, idx_in);
            size_t cub_bytes = 0, scan_bytes = 0;
            FC_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, key_in, key_out, idx_in, rs_idx, (int)n_psm, 0, 64, st));
            FC_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, words, word_off, (int)(n_psm + 1), st));
            char *cub_tmp = nullptr;
            FC_TRY(t.alloc(&cub_tmp, std::max(cub_bytes, scan_bytes)));
            FC_TRY(hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, key_in, key_out, idx_in, rs_idx, (int)n_psm, 0, 64, st));
            // PSMs outside every window sort last (window key 0xFFFFFFFF) and have no neighbours
            hipLaunchKernelGGL(range_kernel, dim3(nb256), dim3(256), 0, st, n_psm, rs_idx, win_of, seg, d_rt, rt_tol, valid0, lo,
                               cnt, words);
            FC_TRY(hipMemsetAsync(words + n_psm, 0, 8, st));
            FC_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp, scan_bytes, words, word_off, (int)(n_psm + 1), st));
            FC_TRY(hipMemcpyAsync(&total_words, word_off + n_psm, 8, hipMemcpyDeviceToHost, st));
            FC_TRY(hipStreamSynchronize(st));
            size_t free_b = 0, total_b = 0;
            FC_TRY(hipMemGetInfo(&free_b, &total_b));
            if (total_words * 2 > free_b / 2) serial = true;  // every PSM at one RT: n^2 bits do not fit
        }
        if (serial) {
            S.serial = 1;
            hipLaunchKernelGGL(adh_fc_serial_kernel, dim3((unsigned)n_windows), dim3(ADH_FC_THREADS), 0, st, n_windows, d_ws,
                               d_we, d_rt, d_fs, d_fe, d_mz, rt_tol, ppm_tol, d_valid);
            FC_TRY(hipGetLastError());
            FC_TRY(hipEventRecord(e1, st));
            return hipStreamSynchronize(st);
        }
        S.pairs = (int64_t)total_words * kGroup;  // upper bound (rows round up to 16)
        FC_TRY(t.alloc(&bits, total_words));
        FC_TRY(t.alloc(&state, n_psm));
        FC_TRY(t.alloc(&has_in, n_psm));
        FC_TRY(t.alloc(&unknown, n_psm));
        // the shortcut needs a finite positive tolerance; 1.001 covers the float rounding of cut and quotient
        const float skip = (ppm_tol > 0.0 && ppm_tol < 1e12) ? (float)(ppm_tol * 1.001e-6) : 0.0f;
        const unsigned rows_per_block = 256 / kGroup;
        hipLaunchKernelGGL(adh_fc_edges_kernel, dim3((unsigned)((n_psm + rows_per_block - 1) / rows_per_block)), dim3(256), 0,
                           st, n_psm, rs_idx, lo, cnt, word_off, d_fs, d_fe, d_mz, ppm_tol, skip, valid0, bits, state, has_in,
                           unknown, counters);
        FC_TRY(hipGetLastError());
        uint32_t n_unknown = 0;
        FC_TRY(hipMemcpyAsync(&n_unknown, counters, 4, hipMemcpyDeviceToHost, st));
        FC_TRY(hipStreamSynchronize(st));
        S.unknown = n_unknown;
        uint32_t pending = n_unknown;
        while (pending) {
            FC_TRY(hipMemsetAsync(counters + 1, 0, 4, st));
            hipLaunchKernelGGL(adh_fc_resolve_kernel, dim3((n_unknown + 255) / 256), dim3(256), 0, st, unknown, n_unknown,
                               rs_idx, lo, cnt, word_off, bits, state, counters + 1);
            FC_TRY(hipGetLastError());
            FC_TRY(hipMemcpyAsync(&pending, counters + 1, 4, hipMemcpyDeviceToHost, st));
            FC_TRY(hipStreamSynchronize(st));
            ++S.rounds;
        }
        hipLaunchKernelGGL(adh_fc_final_kernel, dim3(nb256), dim3(256), 0, st, n_psm, rs_idx, lo, cnt, word_off, bits, state,
                           has_in, win_of, d_valid);
        FC_TRY(hipGetLastError());
        FC_TRY(hipEventRecord(e1, st));
        return hipStreamSynchronize(st);
    };
    err = body();
    if (err == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) S.kernel_ms = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return err;
}
#undef FC_TRY

}  // namespace fragcomp
