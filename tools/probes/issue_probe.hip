// Probe: VALU issue cost on gfx950 - cycles per wave64 instruction and per dependent step for the
// instruction kinds the scoring kernels are made of.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o issue_probe issue_probe.hip && ./issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ITER 2000
template <int KIND, int CHAINS>
__global__ __launch_bounds__(64) void probe(float *out, float seed) {
    float a[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x + i; d[i] = a[i]; }
    float s = seed * 0.5f + 1.0f;
    double sd = s;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int c = u % CHAINS;
            if (KIND == 0) a[c] = a[c] + s;                       // v_add_f32
            if (KIND == 1) d[c] = d[c] + sd;                      // v_add_f64
            if (KIND == 2) d[c] = d[c] * sd;                      // v_mul_f64
            if (KIND == 3) d[c] = __builtin_fma(d[c], sd, sd);    // v_fma_f64
            if (KIND == 4) a[c] = __builtin_fmaf(a[c], s, s);     // v_fma_f32
            if (KIND == 5) { d[c] = d[c] + (double)a[c]; }        // v_cvt_f64_f32 + v_add_f64
            if (KIND == 6) a[c] = (a[c] > s) ? a[c] - s : a[c] + 1.0f;  // cmp + cndmask-ish
            if (KIND == 7) a[c] = fminf(a[c], s) + 1.0f;          // min + add
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + (float)d[i];
    if (r == 12345.678f) out[0] = r;
}

template <int KIND, int CHAINS>
void run(const char *name, float *d_out) {
    for (int w = 1; w <= 4; w *= 2) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int blocks = 256 * 4 * w;
        hipLaunchKernelGGL((probe<KIND, CHAINS>), dim3(blocks), dim3(64), 0, 0, d_out, 1.0f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<KIND, CHAINS>), dim3(blocks), dim3(64), 0, 0, d_out, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)ITER * 32 * (KIND == 5 ? 2 : (KIND >= 6 ? 2 : 1));
        printf("%-28s chains %d waves/SIMD %d: %.3f ms, %.2f cycles per instruction and wave, %.2f per instruction and SIMD (2.4 GHz)\n", name,
               CHAINS, w, ms, ms * 1e-3 * 2.4e9 / insts, ms * 1e-3 * 2.4e9 / insts / w);
    }
}

int main() {
    float *d_out;
    hipMalloc(&d_out, 64);
    run<0, 8>("v_add_f32", d_out);
    run<0, 1>("v_add_f32", d_out);
    run<4, 8>("v_fma_f32", d_out);
    run<1, 8>("v_add_f64", d_out);
    run<1, 1>("v_add_f64", d_out);
    run<2, 8>("v_mul_f64", d_out);
    run<3, 8>("v_fma_f64", d_out);
    run<3, 1>("v_fma_f64", d_out);
    run<5, 8>("cvt_f64_f32 + add_f64", d_out);
    run<5, 1>("cvt_f64_f32 + add_f64", d_out);
    run<6, 8>("cmp + select (2 instr)", d_out);
    run<7, 8>("min + add (2 instr)", d_out);
    run<7, 1>("min + add (2 instr)", d_out);
    return 0;
}
