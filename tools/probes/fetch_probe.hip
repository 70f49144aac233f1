// Probe: what does rocprofv3's FETCH_SIZE report for the gather's access pattern (one narrow load per lane, each
// in its own 128-byte line, lines scattered over gigabytes), next to a wide coalesced stream?
//   hipcc --offload-arch=gfx950 -O3 -o fetch_probe fetch_probe.hip
//   rocprofv3 --pmc FETCH_SIZE -d out -o p -- ./fetch_probe      (then tools/rocpd_summary.py out/p_results.db)
// Every kernel prints the bytes it asked for and the distinct 128-byte lines it touched.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void stream16(const uint4 *__restrict__ a, size_t n16, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = a[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}

// lane t reads BYTES bytes at the start of line perm(t): every load its own line, lines in random order
template <int BYTES>
__global__ void scattered(const unsigned char *__restrict__ a, size_t n_lines, size_t n_loads, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_loads; t += (size_t)gridDim.x * blockDim.x) {
        const size_t line = (t * 2654435761ull + 12345ull) % n_lines;  // (a permutation-like scatter; n_lines is prime-ish)
        if (BYTES == 4) acc += *reinterpret_cast<const uint32_t *>(a + line * 128);
        if (BYTES == 8) {
            const uint2 v = *reinterpret_cast<const uint2 *>(a + line * 128);
            acc += v.x ^ v.y;
        }
        if (BYTES == 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(a + line * 128);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}

int main() {
    const size_t bytes = 6ull << 30;  // well past the 256 MB Infinity Cache
    unsigned char *d;
    uint32_t *o;
    hipMalloc(&d, bytes);
    hipMalloc(&o, 64);
    hipMemset(d, 1, bytes);
    const size_t n_lines = bytes / 128 - 7, n_loads = 1ull << 27;  // 134 M loads, (almost) all in distinct lines
    hipLaunchKernelGGL(stream16, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)d, bytes / 16, o);
    hipLaunchKernelGGL((scattered<4>), dim3(256 * 16), dim3(256), 0, 0, d, n_lines, n_loads, o);
    hipLaunchKernelGGL((scattered<8>), dim3(256 * 16), dim3(256), 0, 0, d, n_lines, n_loads, o);
    hipLaunchKernelGGL((scattered<16>), dim3(256 * 16), dim3(256), 0, 0, d, n_lines, n_loads, o);
    hipDeviceSynchronize();
    printf("stream16: %zu bytes asked; scattered<4|8|16>: %zu loads = %zu lines of 128 B (%zu bytes of lines, %zu of 64-B sectors)\n",
           bytes, n_loads, n_loads, n_loads * 128, n_loads * 64);
    return 0;
}
