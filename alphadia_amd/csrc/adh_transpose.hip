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

// ---- runs of 2^31 detector events and more: the pushes are cut into slabs of < 2^31 events (the radix sort
// and the event numbers stay 32-bit inside a slab); a first pass counts every slab's events per TOF bin, the
// counts give tof_indptr and, per slab, where its run of every bin starts inside the bin (slabs are in push
// order, so a bin's events stay ascending by push); a second pass sorts each slab and scatters its runs.

__global__ void adh_tof_count_kernel(const uint32_t *__restrict__ tof, int64_t n, int64_t n_tof,
                                     unsigned long long *__restrict__ count, int *__restrict__ bad) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t t = tof[i];
        if ((int64_t)t >= n_tof) {
            *bad = 1;
            continue;
        }
        atomicAdd(&count[t], 1ull);
    }
}

// prior[t] = events of bin t in earlier slabs (the running count BEFORE this slab), then the running count
// takes this slab in
__global__ void adh_tof_prior_kernel(const unsigned long long *__restrict__ slab_count, int64_t n_tof,
                                     unsigned long long *__restrict__ running, unsigned long long *__restrict__ prior) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tof) return;
    const unsigned long long r = running[t];
    prior[t] = r;
    running[t] = r + slab_count[t];
}

// sorted slab -> its place in the whole: position i of the slab (TOF index tof_sorted[i], the i - slab_indptr[tof]-th
// event of that bin in this slab) goes to tof_indptr[tof] + prior[tof] + (i - slab_indptr[tof])
__global__ void adh_transpose_scatter_kernel(const uint32_t *__restrict__ tof_sorted, const uint32_t *__restrict__ event,
                                             const uint32_t *__restrict__ push_of, const uint16_t *__restrict__ values,
                                             int64_t n_slab, const int64_t *__restrict__ slab_indptr,
                                             const int64_t *__restrict__ tof_indptr, const unsigned long long *__restrict__ prior,
                                             uint32_t *__restrict__ push_out, uint16_t *__restrict__ values_out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_slab; i += stride) {
        const uint32_t t = tof_sorted[i], e = event[i];
        const int64_t dest = tof_indptr[t] + (int64_t)prior[t] + (i - slab_indptr[t]);
        push_out[dest] = push_of[e];
        values_out[dest] = values[e];
    }
}

// push index of the events of pushes [p0, p1), numbered from the slab's first event; pushes keep their global number
__global__ void adh_expand_push_slab_kernel(const int64_t *__restrict__ push_indptr, int64_t p0, int64_t p1, int64_t e0,
                                            uint32_t *__restrict__ push_of) {
    int64_t p = p0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < p1; p += stride) {
        const int64_t a = push_indptr[p] - e0, b = push_indptr[p + 1] - e0;
        for (int64_t e = a; e < b; ++e) push_of[e] = (uint32_t)p;
    }
}

// kernels index tof_indptr with the TOF index unchecked: refuse a table whose indices leave it
__global__ void adh_tof_range_kernel(const uint32_t *__restrict__ tof, int64_t n, int64_t n_tof, int *__restrict__ bad) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool b = false;
    for (; i < n; i += stride) b |= (int64_t)tof[i] >= n_tof;
    if (b) *bad = 1;
}
