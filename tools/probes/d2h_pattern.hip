// d2h_pattern: the copy pattern of adh_score_candidates without its kernels - per call a copy-in burst on one stream,
// per chunk eight copy-outs (one per table: features, five fragment tables, slots, valid) on another, from ONE device
// buffer into separate page-locked host arrays - for several table sizes in one process.  Prints the copy-out rate of
// every chunk.    hipcc --offload-arch=gfx950 -O3 -o d2h_pattern tools/probes/d2h_pattern.hip && ./d2h_pattern 1500000 3000000 1500000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>
#include <thread>
#include <atomic>

int main(int argc, char **argv) {
    std::vector<long> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back(atol(argv[i]));
    if (sizes.empty()) sizes = {1500000, 3000000, 1500000, 600000};
    const int widths[8] = {184, 48, 48, 48, 48, 48, 24, 1};  // bytes per row of the eight tables on the wire
    const int in_w = 91;
    char *dev = nullptr, *dev_in = nullptr, *host_in = nullptr, *host[8] = {};
    hipStream_t si, so, sk;
    hipStreamCreateWithFlags(&sk, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&si, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&so, hipStreamNonBlocking);
    const bool grow = getenv("GROW") != nullptr;  // allocate per size (grow-only) instead of once for the largest
    long cap = grow ? 0 : *std::max_element(sizes.begin(), sizes.end());
    auto reserve = [&](long n) {
        if (n <= cap && dev && host_in) return;
        cap = std::max(cap, n);
        if (!dev) {
            hipMalloc((void **)&dev, (size_t)cap * 646);
            hipMalloc((void **)&dev_in, (size_t)cap * in_w);
        }
        if (!host_in) {
            hipHostMalloc((void **)&host_in, (size_t)cap * in_w, hipHostMallocPortable);
            for (int f = 0; f < 8; ++f) hipHostMalloc((void **)&host[f], (size_t)cap * widths[f] * 9 / 8, hipHostMallocPortable);
        }
    };
    // EARLY=all|dev|host: these buffers are allocated before the ballast and its churn (default: all after)
    const char *early = getenv("EARLY");
    {
        const long mx = *std::max_element(sizes.begin(), sizes.end());
        if (early && (early[0] == 'a' || early[0] == 'd')) {
            hipMalloc((void **)&dev, (size_t)mx * 646);
            hipMalloc((void **)&dev_in, (size_t)mx * in_w);
        }
        if (early && (early[0] == 'a' || early[0] == 'h')) {
            hipHostMalloc((void **)&host_in, (size_t)mx * in_w, hipHostMallocPortable);
            for (int f = 0; f < 8; ++f) hipHostMalloc((void **)&host[f], (size_t)mx * widths[f] * 9 / 8, hipHostMallocPortable);
        }
    }
    if (getenv("BALLAST_GB")) {  // device memory held by the process, in pieces (a staged run is several buffers of GBs)
        const int gb = atoi(getenv("BALLAST_GB")), pieces = getenv("BALLAST_PIECES") ? atoi(getenv("BALLAST_PIECES")) : 4;
        for (int i = 0; i < pieces; ++i) {
            void *b = nullptr;
            hipMalloc(&b, ((size_t)gb << 30) / pieces);
            hipMemset(b, i, ((size_t)gb << 30) / pieces);
        }
        if (getenv("BALLAST_CHURN")) {  // temporaries of a staging sort: allocated, used, freed
            // BALLAST_CHURN = letters: m memset, f free, s the memset on a stream; CHURN_MB = size of a temporary
            const char *mode = getenv("BALLAST_CHURN");
            const size_t bytes = (size_t)(getenv("CHURN_MB") ? atol(getenv("CHURN_MB")) : 4096) << 20;
            for (int i = 0; i < 4; ++i) {
                void *b = nullptr;
                hipMalloc(&b, bytes);
                if (strchr(mode, 'm')) hipMemset(b, i, bytes);
                if (strchr(mode, 's')) { hipMemsetAsync(b, i, bytes, sk); hipStreamSynchronize(sk); }
                if (strchr(mode, 'f')) hipFree(b);
            }
        }
    }
    if (const char *r = getenv("RESTORE")) {  // candidates for what puts the copies back to the fast state after the churn
        const size_t two = (size_t)(getenv("RESTORE_MB") ? atol(getenv("RESTORE_MB")) : 2048) << 20;
        void *b = nullptr, *hb = nullptr;
        if (strchr(r, 'M') || strchr(r, 'S') || strchr(r, 'C')) hipMalloc(&b, two);
        if (strchr(r, 'S')) { hipMemsetAsync(b, 1, two, sk); hipStreamSynchronize(sk); }
        if (strchr(r, 'H') || strchr(r, 'C')) hipHostMalloc(&hb, two, hipHostMallocPortable);
        if (strchr(r, 'C')) { hipMemcpyAsync(hb, b, two, hipMemcpyDeviceToHost, so); hipStreamSynchronize(so); }
        if (strchr(r, 'D')) hipDeviceSynchronize();
        if (strchr(r, 'F')) { if (b) hipFree(b); if (hb) hipHostFree(hb); }
    }
    if (getenv("STREAMS_LATE")) {  // the three streams are made anew after the ballast and its churn
        hipStreamDestroy(si); hipStreamDestroy(so); hipStreamDestroy(sk);
        hipStreamCreateWithFlags(&sk, hipStreamNonBlocking);
        hipStreamCreateWithFlags(&si, hipStreamNonBlocking);
        hipStreamCreateWithFlags(&so, hipStreamNonBlocking);
    }
    char *meta = nullptr;
    hipHostMalloc((void **)&meta, 4096, hipHostMallocDefault);
    hipEvent_t ep;
    hipEventCreate(&ep);
    for (long n : sizes) {
        reserve(n);
        const long chunk = (n + ((n + 524287) / 524288) - 1) / ((n + 524287) / 524288);
        std::vector<long> cut{0};
        if (n > chunk) cut.push_back(chunk / (n >= 4 * chunk ? 2 : 4));
        while (cut.back() < n) cut.push_back(std::min(n, cut.back() + chunk));
        const int nc = (int)cut.size() - 1;
        for (int call = 0; call < 3; ++call) {
            std::vector<hipEvent_t> e0(nc), e1(nc), ek(nc);
            for (int c = 0; c < nc; ++c) { hipEventCreate(&e0[c]); hipEventCreate(&e1[c]); hipEventCreate(&ek[c]); }
            hipMemsetAsync(dev, 0, (size_t)n * 646, sk);
            hipMemcpyAsync(dev_in, host_in, (size_t)cut[1] * in_w, hipMemcpyHostToDevice, si);
            if (nc > 1) hipMemcpyAsync(dev_in + cut[1] * in_w, host_in + cut[1] * in_w, (size_t)(cut[2] - cut[1]) * in_w, hipMemcpyHostToDevice, si);
            for (int c = 0; c < nc; ++c) {
                if (c == 0 && nc > 2)  // the rest of the columns as 14 copies
                    for (int k = 0; k < 14; ++k) {
                        const size_t a = (size_t)cut[2] * in_w + (size_t)(n - cut[2]) * in_w * k / 14, b = (size_t)cut[2] * in_w + (size_t)(n - cut[2]) * in_w * (k + 1) / 14;
                        hipMemcpyAsync(dev_in + a, host_in + a, b - a, hipMemcpyHostToDevice, si);
                    }
                if (getenv("PLAN")) {  // the plan of the chunk: a few small operations on the copy-in stream, 64 bytes back, the host waits
                    hipMemsetAsync(dev_in, 0, 4096, si);
                    hipMemcpyAsync(meta, dev_in, 64, hipMemcpyDeviceToHost, si);
                    hipEventRecord(ep, si);
                    hipEventSynchronize(ep);
                }
                hipMemsetAsync(dev + (size_t)cut[c] * 8, 1, (size_t)(cut[c + 1] - cut[c]) * 8, sk);  // stands for the chunk's kernels
                hipEventRecord(ek[c], sk);
                hipStreamWaitEvent(so, ek[c], 0);
                hipEventRecord(e0[c], so);
                size_t off = 0;
                for (int f = 0; f < 8; ++f) {
                    hipMemcpyAsync(host[f] + (size_t)cut[c] * widths[f], dev + off + (size_t)cut[c] * widths[f],
                                   (size_t)(cut[c + 1] - cut[c]) * widths[f], hipMemcpyDeviceToHost, so);
                    off += ((size_t)n * widths[f] + 255) / 256 * 256;
                }
                hipEventRecord(e1[c], so);
            }
            if (getenv("THREADS")) {  // a team of host threads waits for every chunk's copy-out, as the column rebuild does
                std::vector<std::thread> team;
                for (int t = 0; t < 16; ++t)
                    team.emplace_back([&] {
                        for (int c = 0; c < nc; ++c) {
                            if (getenv("THREADS")[0] == 'q') while (hipEventQuery(e1[c]) == hipErrorNotReady) {}
                            else hipEventSynchronize(e1[c]);
                        }
                    });
                for (auto &t : team) t.join();
            }
            hipStreamSynchronize(so);
            hipStreamSynchronize(si);
            hipStreamSynchronize(sk);
            if (call == 2) {
                printf("n = %ld (%d chunks):", n, nc);
                for (int c = 0; c < nc; ++c) {
                    float ms = 0;
                    hipEventElapsedTime(&ms, e0[c], e1[c]);
                    printf(" %.1f", (double)(cut[c + 1] - cut[c]) * 449 / 1e6 / ms);
                }
                printf(" GB/s\n");
            }
            for (int c = 0; c < nc; ++c) { hipEventDestroy(e0[c]); hipEventDestroy(e1[c]); hipEventDestroy(ek[c]); }
        }
    }
    return 0;
}
