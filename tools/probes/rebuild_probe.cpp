// Probe: how fast can host threads rebuild the library / id columns of the fragment tables from
// fragment_lib_slot (instead of copying them over PCIe)?  g++ -O3 -march=native -pthread
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

struct LibRec {
    float mz_library, mz, intensity;
    uint8_t type, loss_type, charge, number, position, cardinality, pad0, pad1;
    uint32_t pad2[3];
};

int main(int argc, char **argv) {
    const int64_t n_prec = 1000000, per = 3, n = n_prec * per;
    const int K = 12, top_k = 12;
    const int threads = argc > 1 ? atoi(argv[1]) : 16;
    std::vector<LibRec> lib((size_t)n_prec * K);
    for (size_t i = 0; i < lib.size(); ++i) {
        lib[i].mz_library = lib[i].mz = 200.f + (float)(i % 977);
        lib[i].type = 98 + (i & 1) * 23;
        lib[i].charge = 1;
        lib[i].number = (uint8_t)(i % K);
        lib[i].position = (uint8_t)(i % 7);
        lib[i].loss_type = 0;
    }
    std::vector<uint32_t> frag_start(n), pidx(n);
    std::vector<uint8_t> rank(n);
    std::vector<uint16_t> slot((size_t)n * top_k);
    for (int64_t i = 0; i < n; ++i) {
        pidx[i] = (uint32_t)(i / per);
        frag_start[i] = (uint32_t)((i / per) * K);
        rank[i] = (uint8_t)(i % per);
        for (int j = 0; j < top_k; ++j) slot[(size_t)i * top_k + j] = (i % 5 && j < 9) ? (uint16_t)(1 + (j * 5 + i) % K) : 0;
    }
    std::vector<uint32_t> o_pidx((size_t)n * top_k);
    std::vector<uint8_t> o_rank((size_t)n * top_k), o_pos((size_t)n * top_k), o_num((size_t)n * top_k),
        o_type((size_t)n * top_k), o_ch((size_t)n * top_k), o_loss((size_t)n * top_k);
    std::vector<float> o_mzl((size_t)n * top_k), o_mz((size_t)n * top_k);
    auto work = [&](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) {
            const LibRec *base = lib.data() + frag_start[i];
            const uint32_t p = pidx[i];
            const uint8_t r = rank[i];
            for (int j = 0; j < top_k; ++j) {
                const size_t o = (size_t)i * top_k + j;
                const uint16_t s = slot[o];
                if (s) {
                    const LibRec &l = base[s - 1];
                    o_pidx[o] = p; o_rank[o] = r; o_mzl[o] = l.mz_library; o_mz[o] = l.mz;
                    o_pos[o] = l.position; o_num[o] = l.number; o_type[o] = l.type; o_ch[o] = l.charge; o_loss[o] = l.loss_type;
                } else {
                    o_pidx[o] = 0; o_rank[o] = 0; o_mzl[o] = 0; o_mz[o] = 0;
                    o_pos[o] = 0; o_num[o] = 0; o_type[o] = 0; o_ch[o] = 0; o_loss[o] = 0;
                }
            }
        }
    };
    for (int rep = 0; rep < 4; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t) th.emplace_back(work, n * t / threads, n * (t + 1) / threads);
        for (auto &x : th) x.join();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %d: rebuild of %lld candidates x %d slots in %.1f ms (%.1f ns per candidate)\n", threads, (long long)n, top_k, ms, ms * 1e6 / n);
    }
    return (int)o_mz[12345];
}
