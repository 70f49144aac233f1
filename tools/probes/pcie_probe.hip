// PCIe probe: D2H / H2D rate of page-locked buffers, call by call (does the link need warming up?)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    size_t bytes = (argc > 1 ? atoll(argv[1]) : 2000) * 1000000ull;
    int pieces = argc > 2 ? atoi(argv[2]) : 1;
    void *d, *h;
    hipMalloc(&d, bytes);
    hipMemset(d, 1, bytes);
    hipHostMalloc(&h, bytes, hipHostMallocPortable);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int it = 0; it < 8; ++it) {
        double t0 = now();
        size_t per = bytes / pieces;
        for (int p = 0; p < pieces; ++p) hipMemcpyAsync((char *)h + p * per, (char *)d + p * per, per, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        double t1 = now();
        printf("D2H %d: %.2f ms %.1f GB/s (%d pieces)\n", it, t1 - t0, bytes / (t1 - t0) / 1e6, pieces);
    }
    for (int it = 0; it < 4; ++it) {
        double t0 = now();
        hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double t1 = now();
        printf("H2D %d: %.2f ms %.1f GB/s\n", it, t1 - t0, bytes / (t1 - t0) / 1e6);
    }
    // two streams, two halves
    hipStream_t s2;
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int it = 0; it < 3; ++it) {
        double t0 = now();
        hipMemcpyAsync(h, d, bytes / 2, hipMemcpyDeviceToHost, s);
        hipMemcpyAsync((char *)h + bytes / 2, (char *)d + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, s2);
        hipStreamSynchronize(s);
        hipStreamSynchronize(s2);
        double t1 = now();
        printf("D2H two streams %d: %.2f ms %.1f GB/s\n", it, t1 - t0, bytes / (t1 - t0) / 1e6);
    }
    // pageable
    std::vector<char> pg(bytes);
    for (int it = 0; it < 3; ++it) {
        double t0 = now();
        hipMemcpy(pg.data(), d, bytes, hipMemcpyDeviceToHost);
        double t1 = now();
        printf("D2H pageable %d: %.2f ms %.1f GB/s\n", it, t1 - t0, bytes / (t1 - t0) / 1e6);
    }
    return 0;
}
