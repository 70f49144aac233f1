// D2H probe: what does one copy-out cost by its size - hipMemcpyAsync (the DMA engines) against a kernel that
// stores into page-locked host memory - and how do a few of them behave back to back on one stream?
//   hipcc --offload-arch=gfx950 -O3 -o d2h_probe tools/probes/d2h_probe.hip && ./d2h_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void copy_out_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = src[i];
        __builtin_nontemporal_store(v.x, &dst[i].x);
        __builtin_nontemporal_store(v.y, &dst[i].y);
        __builtin_nontemporal_store(v.z, &dst[i].z);
        __builtin_nontemporal_store(v.w, &dst[i].w);
    }
}
__global__ void copy_out_plain_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

int main() {
    const size_t cap = 256ull << 20;
    void *d, *h;
    hipMalloc(&d, cap);
    hipMemset(d, 1, cap);
    hipHostMalloc(&h, cap, hipHostMallocPortable | hipHostMallocMapped);
    void *hd = nullptr;
    hipHostGetDevicePointer(&hd, h, 0);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t sizes[] = {256u << 10, 1u << 20, 2u << 20, 4u << 20, 8u << 20, 16u << 20, 32u << 20, 64u << 20, 128u << 20};
    for (size_t bytes : sizes) {
        const int reps = 8;
        // (1) DMA: reps copies back to back on one stream
        hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int r = 0; r < reps; ++r) hipMemcpyAsync((char *)h + (size_t)(r % 2) * bytes, (char *)d + (size_t)(r % 2) * bytes, bytes, hipMemcpyDeviceToHost, s);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double dma_us = ms * 1e3 / reps;
        printf("%9zu B  dma %8.1f us %6.1f GB/s", bytes, dma_us, bytes / dma_us / 1e3);
        // (2) kernels of several grid sizes
        for (int blocks : {16, 64, 256, 1024}) {
            for (int variant = 0; variant < 2; ++variant) {
                auto launch = [&] {
                    if (variant == 0) hipLaunchKernelGGL(copy_out_kernel, dim3(blocks), dim3(256), 0, s, (const uint4 *)d, (uint4 *)hd, bytes / 16);
                    else hipLaunchKernelGGL(copy_out_plain_kernel, dim3(blocks), dim3(256), 0, s, (const uint4 *)d, (uint4 *)hd, bytes / 16);
                };
                launch();
                hipStreamSynchronize(s);
                hipEventRecord(e0, s);
                for (int r = 0; r < reps; ++r) launch();
                hipEventRecord(e1, s);
                hipStreamSynchronize(s);
                hipEventElapsedTime(&ms, e0, e1);
                const double us = ms * 1e3 / reps;
                printf(" | %s%-4d %7.1f us %5.1f GB/s", variant ? "pl" : "nt", blocks, us, bytes / us / 1e3);
            }
        }
        printf("\n");
    }
    // host-side check that the kernel's stores arrived
    const unsigned char *hb = (const unsigned char *)h;
    size_t bad = 0;
    for (size_t i = 0; i < (128u << 20); i += 4097) bad += hb[i] != 1;
    printf("mismatches %zu\n", bad);
    return 0;
}
