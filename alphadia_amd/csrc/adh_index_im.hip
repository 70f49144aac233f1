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
        uint32_t *row = idx + (size_t)tof * cols;
        uint32_t done = 0u;  // columns written so far (wave-uniform)
        for (int64_t base = a; base < b; base += ADH_WAVE) {
            const int64_t e = base + lane;
            const bool in = e < b;
            const uint32_t g = in ? columns_le(push[e], S_max, L, z, shift, n_blocks) : cols;
            uint32_t g_prev = __shfl_up(g, 1);
            if (lane == 0) g_prev = done;
            if (in)
                for (uint32_t cb = g_prev; cb < g; ++cb) row[cb] = (uint32_t)e;
            const int last = (int)min((int64_t)ADH_WAVE, b - base) - 1;
            done = __shfl(g, last);
        }
        for (uint32_t cb = done + lane; cb < cols; cb += ADH_WAVE) row[cb] = (uint32_t)b;
    }
}
