// Probe: is the shared-reciprocal division of adh_fused.hip (fused::Recip) bit-identical to the
// compiler's IEEE float64 division for the operand ranges of the scoring kernels?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o recip_probe recip_probe.hip && ./recip_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#include "../../alphadia_amd/csrc/adh_gather.hip"
#include "../../alphadia_amd/csrc/adh_features_fast.hip"
#include "../../alphadia_amd/csrc/adh_fused.hip"

__device__ uint64_t rng(uint64_t &s) {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
}

__global__ void probe(unsigned long long *bad, unsigned long long *n, int mode) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1) + mode;
    unsigned long long local_bad = 0, local_n = 0;
    for (int it = 0; it < 2000; ++it) {
        // divisor: a float32-derived positive number over a wide exponent range, or 1 + 1e-6
        float df = __uint_as_float((uint32_t)((rng(s) >> 9) & 0x7FFFFF) | ((uint32_t)(90 + rng(s) % 70) << 23));
        double d = mode == 0 ? (double)df / 3.0 : (mode == 1 ? 1.0 + 1e-6 : (double)df);
        fused::Recip r(d);
        for (int j = 0; j < 32; ++j) {
            float xf = __uint_as_float((uint32_t)((rng(s) >> 9) & 0x7FFFFF) | ((uint32_t)(90 + rng(s) % 70) << 23));
            double x = (double)xf;
            double a = x / d, b = r.div(x);
            local_bad += (__double_as_longlong(a) != __double_as_longlong(b));
            ++local_n;
        }
    }
    atomicAdd(bad, local_bad);
    atomicAdd(n, local_n);
}

// float32 division == float64 division of the same operands rounded to float32 (gather::fold)
__global__ void probe_div32(unsigned long long *bad, unsigned long long *n) {
    uint64_t s = 0xD1B54A32D192ED03ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long local_bad = 0, local_n = 0;
    for (int it = 0; it < 20000; ++it) {
        const float a = __uint_as_float((uint32_t)((rng(s) >> 9) & 0x7FFFFF) | ((uint32_t)(61 + rng(s) % 160) << 23));
        const float b = __uint_as_float((uint32_t)((rng(s) >> 9) & 0x7FFFFF) | ((uint32_t)(61 + rng(s) % 160) << 23));
        if (!(a > 1e-20f && b > 1e-20f)) continue;
        const float q32 = a / b;
        const float q64 = (float)(((double)a + 1e-36) / ((double)b + 1e-36));
        local_bad += (__float_as_uint(q32) != __float_as_uint(q64));
        ++local_n;
    }
    atomicAdd(bad, local_bad);
    atomicAdd(n, local_n);
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 16);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, d, d + 1, mode);
        unsigned long long h[2];
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("mode %d: %llu of %llu quotients differ from x / d\n", mode, h[0], h[1]);
    }
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(probe_div32, dim3(1024), dim3(256), 0, 0, d, d + 1);
    unsigned long long h[2];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("float32 a / b vs (float)((a + 1e-36) / (b + 1e-36)) in float64: %llu of %llu differ\n", h[0], h[1]);
    return 0;
}
