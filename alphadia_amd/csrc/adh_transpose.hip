// adh_transpose.hip - timsTOF frame-major -> TOF-major transposition on the device.
//
// Replaces `_transpose` / `_transpose_chunk` (alphadia/raw_data/bruker.py:155-280): the
// alphatims layout is a CSR matrix with pushes as rows (push_indptr, tof_indices, values);
// scoring and selection want TOF bins as rows (tof_indptr, push_indices, values) with the
// pushes of a bin in ascending order.  That is a stable counting sort by TOF index: here a
// stable radix sort of (tof, event index) pairs followed by two gathers.
#include "adh_device.h"

// push index of every detector event: row p owns events [push_indptr[p], push_indptr[p + 1])
__global__ void adh_expand_push_kernel(const int64_t *__restrict__ push_indptr, int64_t n_push,
                                       uint32_t *__restrict__ push_of) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < n_push; p += stride) {
        const int64_t a = push_indptr[p], b = push_indptr[p + 1];
        for (int64_t e = a; e < b; ++e) push_of[e] = (uint32_t)p;
    }
}

__global__ void adh_iota_kernel(uint32_t *__restrict__ v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) v[i] = (uint32_t)i;
}

// sorted (tof, event) pairs -> push_indices, values in TOF-major order
__global__ void adh_transpose_gather_kernel(const uint32_t *__restrict__ event, const uint32_t *__restrict__ push_of,
                                            const uint16_t *__restrict__ values, int64_t n,
                                            uint32_t *__restrict__ push_out, uint16_t *__restrict__ values_out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t e = event[i];
        push_out[i] = push_of[e];
        values_out[i] = values[e];
    }
}

// tof_indptr[t] = first position whose TOF index is >= t (t = 0..n_tof)
__global__ void adh_tof_indptr_kernel(const uint32_t *__restrict__ tof_sorted, int64_t n, int64_t n_tof,
                                      int64_t *__restrict__ tof_indptr) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t <= n_tof; t += stride) {
        int64_t a = 0, b = n;
        while (a < b) {
            const int64_t m = (a + b) >> 1;
            if ((int64_t)tof_sorted[m] < t) a = m + 1; else b = m;
        }
        tof_indptr[t] = a;
    }
}
