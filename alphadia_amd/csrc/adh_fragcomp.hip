// adh_fragcomp.hip - fragment competition on gfx950.
//
// Replaces `_compete_for_fragments` + `_get_fragment_overlap`
// (alphadia/fragcomp/fragcomp.py:19-143).  The reference parallelises over DIA
// windows with threads and walks each window with two nested sequential loops.
// The greedy rule is order dependent in `i` (a PSM that was removed can no longer
// remove others) but independent in `j`, so here one workgroup owns one window,
// walks `i` sequentially and spreads `j` across its 256 lanes; the m/z list of
// PSM `i` is staged once in LDS and re-used by every lane.
#include "adh_device.h"

#define ADH_FC_THREADS 256
#define ADH_FC_MAXFRAG 512

__global__ __launch_bounds__(ADH_FC_THREADS) void adh_fragcomp_kernel(
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
