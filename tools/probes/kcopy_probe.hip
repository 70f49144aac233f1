// kcopy_probe.hip - how fast can a KERNEL write device data into page-locked host memory (zero-copy stores over
// PCIe), next to hipMemcpyAsync of the same bytes?  Decides whether the copy-out of small chunks (nine DMA copies with
// ~10 us of idle engine between them) and the compacted, column-major copy-out of the operator path can be kernels.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/kcopy_probe tools/probes/kcopy_probe.hip && tools/probes/kcopy_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);    \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

template <typename T>
__global__ void copy_kernel(const T *__restrict__ src, T *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// transposing pack: rows [n][46] float -> columns [46][cap], lanes along rows (what the compact copy-out would do)
__global__ void transpose_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t n, size_t cap) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int f = 0; f < 46; ++f) dst[(size_t)f * cap + i] = src[i * 46 + f];
    }
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const size_t bytes = 64ull << 20;
    void *d = nullptr, *h = nullptr;
    CHECK(hipMalloc(&d, bytes));
    CHECK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CHECK(hipMemset(d, 1, bytes));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // DMA reference
    for (size_t sz : {(size_t)1 << 20, (size_t)8 << 20, bytes}) {
        for (int rep = 0; rep < 3; ++rep) CHECK(hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, st));
        CHECK(hipStreamSynchronize(st));
        const double t0 = now_ms();
        const int reps = 10;
        for (int rep = 0; rep < reps; ++rep) CHECK(hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, st));
        CHECK(hipStreamSynchronize(st));
        const double ms = (now_ms() - t0) / reps;
        printf("hipMemcpyAsync D2H %6zu KB: %.3f ms = %.1f GB/s\n", sz >> 10, ms, sz / ms / 1e6);
    }
    for (int width : {4, 8, 16}) {
        for (int blocks : {16, 32, 64, 128, 256, 1024}) {
            for (size_t sz : {(size_t)8 << 20, bytes}) {
                auto launch = [&] {
                    if (width == 4) hipLaunchKernelGGL(copy_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)d, (float *)h, sz / 4);
                    else if (width == 8) hipLaunchKernelGGL(copy_kernel<float2>, dim3(blocks), dim3(256), 0, st, (const float2 *)d, (float2 *)h, sz / 8);
                    else hipLaunchKernelGGL(copy_kernel<float4>, dim3(blocks), dim3(256), 0, st, (const float4 *)d, (float4 *)h, sz / 16);
                };
                for (int rep = 0; rep < 2; ++rep) launch();
                CHECK(hipStreamSynchronize(st));
                const double t0 = now_ms();
                const int reps = 6;
                for (int rep = 0; rep < reps; ++rep) launch();
                CHECK(hipStreamSynchronize(st));
                const double ms = (now_ms() - t0) / reps;
                printf("kernel copy %2d B/lane, %4d blocks x 256, %6zu KB: %.3f ms = %.1f GB/s\n", width, blocks, sz >> 10, ms,
                       sz / ms / 1e6);
            }
        }
    }
    {
        const size_t n = bytes / (46 * 4);
        for (int blocks : {32, 64, 128, 256}) {
            for (int rep = 0; rep < 2; ++rep)
                hipLaunchKernelGGL(transpose_kernel, dim3(blocks), dim3(256), 0, st, (const float *)d, (float *)h, n, n);
            CHECK(hipStreamSynchronize(st));
            const double t0 = now_ms();
            const int reps = 6;
            for (int rep = 0; rep < reps; ++rep)
                hipLaunchKernelGGL(transpose_kernel, dim3(blocks), dim3(256), 0, st, (const float *)d, (float *)h, n, n);
            CHECK(hipStreamSynchronize(st));
            const double ms = (now_ms() - t0) / reps;
            printf("kernel transpose-out [n][46] -> [46][n], %4d blocks: %.3f ms = %.1f GB/s\n", blocks, ms, n * 184 / ms / 1e6);
        }
    }
    // host-side check of the last kernel's bytes
    const unsigned char *hb = static_cast<const unsigned char *>(h);
    size_t bad = 0;
    for (size_t i = 0; i < (bytes / 184) * 184; i += 4097) bad += hb[i] != 1;
    printf("spot check: %zu bad bytes\n", bad);
    return 0;
}
