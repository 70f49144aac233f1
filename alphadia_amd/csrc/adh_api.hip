// adh_api.hip - host side of libalphadia_hip.so: the C ABI declared in
// include/alphadia_hip.h.  Owns the HBM-resident copies of the run, the fragment
// library and the candidate table, sizes the LDS of the scoring kernel per batch
// and times the kernel with HIP events on its launch stream.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <string>
#include <vector>
#include <mutex>
#include <unordered_map>

#include <hip/hip_ext.h>
#include <hipcub/hipcub.hpp>

#include <dlfcn.h>
#include <immintrin.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include "adh_index_im.hip"
#include "adh_plan.hip"
#include "adh_gather.hip"
#include "adh_features.hip"
#include "adh_features_fast.hip"
#include "adh_fused.hip"
#include "adh_gather_im.hip"
#include "adh_features_im.hip"
#include "adh_features_im4.hip"
#include "adh_features_im2.hip"
#include "adh_fragcomp.hip"
#include "adh_select.hip"
#include "adh_select_im.hip"
#include "adh_transpose.hip"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            (void)hipGetLastError();                                                        \
            return fail(_e == hipErrorOutOfMemory ? ADH_ERR_OUT_OF_MEMORY : ADH_ERR_HIP,   \
                        std::string(#expr) + " (adh_api.hip:" + std::to_string(__LINE__) + "): " +   \
                            hipGetErrorString(_e));                                         \
        }                                                                                   \
    } while (0)

// ---- large device buffers are kept, not returned (round 5).  Freeing >= 2 GB of device memory leaves the runtime's DMA
// copies at half the link rate for the rest of the process (tools/probes/d2h_pattern.hip; round 4 cured that by
// page-locking 2 GB of host memory once after such a free - settle_copy_path, now only with ADH_COPY_PATH_RESET=1).  The
// trigger was the library itself: staging a run frees two 3.9 GB sort buffers, re-growing tables or scratch frees the old
// ones.  Every hipMalloc / hipFree of this translation unit goes through adh_dev_malloc / adh_dev_free (the macros
// below): a freed block of 256 MB or more is parked and handed to the next request it fits (size <= block <= 1.5 x
// size) - the second staging of a run of similar size allocates nothing - up to ADH_DEV_CACHE_GB (default 48) in all;
// adh_trim_device_cache() gives everything back.  A request that fails with the cache non-empty empties it and retries.
// Blocks carry the device they were allocated on: a parked block only serves a request made with the same device
// current (one process may hold a handle per GPU), and the limit counts per device.  Parking keeps hipFree's implicit
// device synchronisation (kernels of any stream may still touch the block; parks are rare - a re-staged run, a table
// that outgrew its slab), so a reused block is never written while its old owner's work is in flight.
struct DevBlock {
    void *p;
    size_t bytes;
    int device;
};
struct DevBlockCache {
    std::mutex m;
    std::vector<DevBlock> parked;
    std::unordered_map<void *, std::pair<size_t, int>> live;  // blocks of >= kMin bytes handed out: (bytes, device)
    static constexpr size_t kMin = (size_t)256 << 20;
    size_t parked_on(int device) const {
        size_t b = 0;
        for (const DevBlock &k : parked)
            if (k.device == device) b += k.bytes;
        return b;
    }
};
DevBlockCache &dev_cache() {
    static DevBlockCache c;
    return c;
}
std::atomic<bool> g_big_free{false};  // a block of 1 GB or more did go back to the runtime

int dev_current() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
}

// per device: ADH_DEV_CACHE_GB (default 48), never more than a quarter of the device's memory
size_t dev_cache_limit() {
    static const size_t lim = [] {
        const char *env = getenv("ADH_DEV_CACHE_GB");
        size_t l = (size_t)(env ? std::max(atof(env), 0.0) : 48.0) << 30;
        size_t free_b = 0, total_b = 0;
        if (::hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) l = std::min(l, total_b / 4);
        return l;
    }();
    return lim;
}

// gives back the parked blocks of one device (device < 0: of every device)
hipError_t adh_dev_trim_device(int device) {
    DevBlockCache &c = dev_cache();
    std::vector<DevBlock> drop;
    {
        std::lock_guard<std::mutex> g(c.m);
        std::vector<DevBlock> keep;
        for (const DevBlock &b : c.parked) (device < 0 || b.device == device ? drop : keep).push_back(b);
        c.parked.swap(keep);
    }
    hipError_t e = hipSuccess;
    int cur = dev_current();
    for (const DevBlock &b : drop) {
        if (b.bytes >= ((size_t)1 << 30)) g_big_free.store(true);
        if (b.device != cur) (void)hipSetDevice(b.device);
        const hipError_t f = hipFree(b.p);
        if (b.device != cur) (void)hipSetDevice(cur);
        if (f != hipSuccess) e = f;
    }
    return e;
}
hipError_t adh_dev_trim() { return adh_dev_trim_device(-1); }

hipError_t adh_dev_malloc(void **p, size_t bytes) {
    DevBlockCache &c = dev_cache();
    const int device = dev_current();
    if (bytes >= DevBlockCache::kMin) {
        std::lock_guard<std::mutex> g(c.m);
        size_t best = SIZE_MAX, at = SIZE_MAX;
        for (size_t i = 0; i < c.parked.size(); ++i)
            if (c.parked[i].device == device && c.parked[i].bytes >= bytes && c.parked[i].bytes <= bytes + bytes / 2 &&
                c.parked[i].bytes < best)
                best = c.parked[i].bytes, at = i;
        if (at != SIZE_MAX) {
            *p = c.parked[at].p;
            c.live[*p] = {best, device};
            c.parked.erase(c.parked.begin() + (long)at);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        bool any;
        {
            std::lock_guard<std::mutex> g(c.m);
            any = c.parked_on(device) > 0;
        }
        if (any) {
            (void)adh_dev_trim_device(device);
            e = hipMalloc(p, bytes);
        }
    }
    if (e == hipSuccess && bytes >= DevBlockCache::kMin) {
        std::lock_guard<std::mutex> g(c.m);
        c.live[*p] = {bytes, device};
    }
    return e;
}

hipError_t adh_dev_free(void *p) {
    if (!p) return hipSuccess;
    DevBlockCache &c = dev_cache();
    size_t bytes = 0;
    int device = 0;
    bool park = false;
    {
        std::lock_guard<std::mutex> g(c.m);
        auto it = c.live.find(p);
        if (it != c.live.end()) {
            bytes = it->second.first;
            device = it->second.second;
            c.live.erase(it);
            park = c.parked_on(device) + bytes <= dev_cache_limit();
        }
    }
    if (park) {
        // what hipFree would have done: nothing in flight on the block's device touches it any more
        const int cur = dev_current();
        if (device != cur) (void)hipSetDevice(device);
        const hipError_t e = hipDeviceSynchronize();
        if (device != cur) (void)hipSetDevice(cur);
        if (e != hipSuccess) (void)hipGetLastError();  // (a failed device: hand the block on anyway, the next call reports)
        std::lock_guard<std::mutex> g(c.m);
        c.parked.push_back(DevBlock{p, bytes, device});
        return hipSuccess;
    }
    if (bytes >= ((size_t)1 << 30)) g_big_free.store(true);
    return hipFree(p);
}

// hipMemGetInfo as the sizing heuristics mean it: the parked blocks of the current device are free memory (a request
// that does not fit trims them)
hipError_t adh_mem_get_info(size_t *free_b, size_t *total_b) {
    const hipError_t e = hipMemGetInfo(free_b, total_b);
    if (e != hipSuccess) return e;
    DevBlockCache &c = dev_cache();
    std::lock_guard<std::mutex> g(c.m);
    *free_b = std::min(*total_b, *free_b + c.parked_on(dev_current()));
    return hipSuccess;
}
#define hipMalloc(ptr, bytes) adh_dev_malloc((void **)(ptr), (bytes))
#define hipFree(ptr) adh_dev_free((void *)(ptr))
#define hipMemGetInfo(f, t) adh_mem_get_info((f), (t))

// CPU cores this process may actually use: the smallest of the hardware threads, the scheduler affinity mask and the
// cgroup CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).  The pool's GPU boxes show 256 hardware
// threads under a quota of 16 cores: a team sized by the former only takes turns.
// ADH_CGROUP_CPU_MAX names another file in cpu.max format ("<quota> <period>" or "max <period>") - the tests fake one.
int host_cpu_budget() {
    int hw = (int)std::thread::hardware_concurrency();
    if (hw <= 0) hw = 16;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int aff = CPU_COUNT(&set);
        if (aff > 0) hw = std::min(hw, aff);
    }
    auto read_two = [](const char *path, char *a, size_t na, double *b) -> bool {
        FILE *f = fopen(path, "r");
        if (!f) return false;
        char fmt[32];
        snprintf(fmt, sizeof(fmt), "%%%zus %%lf", na - 1);
        const int k = fscanf(f, fmt, a, b);
        fclose(f);
        return k >= 1;
    };
    char q[64] = {0};
    double period = 100000.0;
    const char *fake = getenv("ADH_CGROUP_CPU_MAX");
    if (read_two(fake && *fake ? fake : "/sys/fs/cgroup/cpu.max", q, sizeof(q), &period)) {
        if (strcmp(q, "max") != 0 && period > 0) {
            const double cores = atof(q) / period;
            if (cores > 0) hw = std::min(hw, std::max((int)cores, 1));
        }
    } else {
        double quota = -1, per = -1;
        char dummy[64];
        FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        if (f) {
            if (fscanf(f, "%lf", &quota) != 1) quota = -1;
            fclose(f);
        }
        f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (f) {
            if (fscanf(f, "%lf", &per) != 1) per = -1;
            fclose(f);
        }
        (void)dummy;
        if (quota > 0 && per > 0) hw = std::min(hw, std::max((int)(quota / per), 1));
    }
    return std::max(hw, 1);
}

struct DeviceBuffers {
    std::vector<void *> ptrs;
    void release() {
        for (void *p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
};

}  // namespace

// device copies of the candidate columns (adh_candidates_t): one grow-only slab
struct CandSlab {
    void *base = nullptr;
    size_t bytes = 0;
    DevCands d{};
    float *iso = nullptr;
    uint8_t *flags = nullptr;   // storage of the flags column (d.flags stays NULL when the caller passes none)
    int64_t n = 0;
    int32_t n_iso_cols = 0;
};

// device buffers of one plan under construction (adh_plan.hip); two slots alternate between the
// chunks of adh_score_candidates
struct PlanSlot {
    int64_t cap = 0;
    size_t rec_bytes = 0;
    void *recs = nullptr, *ordered = nullptr;
    uint32_t *keys_in = nullptr, *keys_out = nullptr, *idx_in = nullptr, *idx_out = nullptr;
    uint64_t *bytes = nullptr, *sorted_bytes = nullptr, *offs = nullptr;
    void *cub_tmp = nullptr;
    size_t cub_bytes = 0;
    uint32_t *hist = nullptr;       // counting sort: one counter / cursor per (class, first cycle) key
    int64_t hist_cap = 0;
    PlanMeta *d_meta = nullptr, *h_meta = nullptr;  // device / pinned host
    hipEvent_t done = nullptr;
    DeviceBuffers buf;
};

// one processing plan = CandRec table in processing order, for a given (top_k_fragments,
// top_k_isotopes, kernel family) and row range
struct Plan {
    bool ready = false;
    uint32_t top_k_fragments = 0, top_k_isotopes = 0;
    bool fast_ok = false;          // register kernels enabled (experimental_xic)
    bool fused_ok = false;         // fused gather + feature kernel enabled
    bool wide_ok = false;          // wide register kernels (17 ... 64 fragments kept) enabled
    bool quant_all = false;
    int64_t row0 = 0, n = 0;
    CandRec *d_recs = nullptr;
    CandRecIM *d_recs_im = nullptr;
    uint64_t scratch_bytes = 0;
    // candidates per kernel class (adh_plan.hip): fused kernel / register kernels by cycle count (a
    // kernel per four cycles: every cycle loop is unrolled to the class size) / the LDS kernel
    int64_t n_class[ADH_N_CLASSES] = {0};
    Caps caps_generic;
    Caps caps_all;
};

// the OutputPsmDF tables of one call as ONE packed device buffer (computed tables first)
struct DevTables {
    void *base = nullptr;
    size_t bytes = 0, used = 0, wire_bytes = 0;
    int64_t rows = 0;
    int top_k = 0;
    adh_output_t view{};
    // the last call left the columns that repeat ids / the library unwritten (adh_score_candidates rebuilds them
    // on the host): a reader of the device tables has them filled in first (materialise_tables)
    bool partial = false;
};

struct adh_comm_state;

struct adh_handle {
    int device = 0;
    hipStream_t stream = nullptr;        // compute
    hipStream_t stream_in = nullptr, stream_out = nullptr;  // H2D + plan / D2H of adh_score_candidates
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // side streams of the fused launches of one chunk (adh_score_host.hip: launch_scoring), created on first use
    hipStream_t stream_aux[2] = {nullptr, nullptr};
    hipEvent_t ev_aux[2] = {nullptr, nullptr};
    hipEvent_t ev_k[2] = {nullptr, nullptr};  // kernels of the chunk that used plan slot s are done
    DevRun run{};
    DevTims tims{};
    bool tims_staged = false;
    std::vector<double> h_cycle;    // host copy (candidate selection sizes its tiles on the host)
    std::vector<int32_t> h_dpc;     // host copy of dia_precursor_cycle (adh_debug_get_dense)
    const LibRec *d_lib = nullptr;
    std::vector<LibRec> h_lib;            // host copy of the staged library: the library columns of the fragment
                                          // tables are rebuilt from it on the host instead of crossing PCIe
    void *slot_stage = nullptr;           // page-locked fragment_lib_slot staging when the caller passes none
    size_t slot_stage_bytes = 0;
    // compacted copy-out of the fragment tables (adh_score_host.hip): per-row offsets + the filled slots of the six
    // wire columns, on the device and in page-locked host memory; scan scratch
    void *cmp_dev = nullptr, *cmp_host = nullptr, *cmp_scan = nullptr;
    size_t cmp_dev_bytes = 0, cmp_host_bytes = 0, cmp_scan_bytes = 0;
    // adh_score_candidates_compact: per-row counts / offsets on the device, scan scratch, page-locked staging block
    void *cop_cnt = nullptr, *cop_scan = nullptr, *cop_stage = nullptr, *cop_dev = nullptr;
    size_t cop_cnt_bytes = 0, cop_scan_bytes = 0, cop_stage_bytes = 0, cop_dev_bytes = 0;
    uint64_t *cop_tot_pinned = nullptr;     // totals of up to 4096 chunks, page-locked
    std::vector<uint64_t> cop_tot_host;
    int64_t n_lib = 0;
    double *d_wtp = nullptr;        // precursor weight table [2][64]
    uint64_t im_scratch_budget = 0; // bytes the scratch of one ion-mobility chunk may reserve (0: not asked yet)
    std::vector<float> h_rt;        // host copy of the run's rt_values (selection sizes its tiles with it)
    std::vector<double> h_rt_im, h_mobility_im;  // the same for an ion-mobility run
    double last_select_ms = 0.0;    // duration of the last adh_select_kernel launch
    fragcomp::Stats last_fragcomp;  // of the last adh_fragcomp / adh_fdr_resident call
    // The runtime's DMA copies run at half the link rate (25-29 instead of 55 GB/s) from the moment this process has
    // freed a device buffer of 2 GB or more - the temporaries of staging a run do - until it page-locks 2 GB of host
    // memory in one piece (tools/probes/d2h_pattern.hip: BALLAST_CHURN=mf against RESTORE=HF).
    // Whatever frees large device memory sets the flag, the host -> host entry point settles it.
    bool big_staged = false;
    void *sel_slab = nullptr;       // precursor columns + candidate table of adh_select_candidates (grow-only)
    size_t sel_slab_bytes = 0;
    void *scratch_slab = nullptr;   // per-candidate scratch blocks (grow-only, shared by all chunks)
    uint64_t scratch_slab_bytes = 0;
    DevTables tables[2];            // slot 1 only with a communicator (double-buffered all-gather)
    int table_slot = 1, last_tables = -1;
    int64_t last_rows = 0;
    CandSlab cs;
    PlanSlot slots[2];
    Plan plan;                      // plan of the resident table (adh_upload_candidates / adh_score_uploaded)
    bool run_staged = false, lib_staged = false, cands_uploaded = false;
    DeviceBuffers run_buf, lib_buf;
    adh_comm_state *comm = nullptr;
    int64_t comm_rows = 0;          // rows of the largest shard (table layout under a communicator)
    bool comm_attached() const { return comm != nullptr; }
    struct Timed {
        hipEvent_t e0, e1, e2;
    };
    std::vector<Timed> timed;  // per launch: before gather, between, after features
    std::vector<hipEvent_t> free_events;
    double sum_gather_ms = 0.0, sum_feature_ms = 0.0;
    int64_t n_timed = 0;
    uint64_t d2h_bytes = 0;  // bytes this library copied device -> host (adh_transfer_counters)
    // page-locked staging of upload_staged (H2D of a caller's pageable arrays): per lane two buffers, their events, a stream
    struct UpLane {
        void *buf[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        hipStream_t st = nullptr;
    };
    std::vector<UpLane> up_lanes;
};

namespace {

template <typename T>
int upload(DeviceBuffers &owner, const T *host, int64_t n, const T **dev, hipStream_t) {
    *dev = nullptr;
    void *p = nullptr;
    size_t bytes = (size_t)std::max<int64_t>(n, 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    owner.ptrs.push_back(p);
    if (n > 0) HIP_TRY(hipMemcpy(p, host, (size_t)n * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<const T *>(p);
    return ADH_OK;
}

// H2D of a caller's PAGEABLE arrays at the link's rate instead of the runtime's (~8 GB/s for pageable memory on these
// boxes): every job is cut into one slice per lane; a lane's thread copies its slice, 4 MB at a time, into one of its two
// page-locked buffers and enqueues the DMA copy of that buffer on its own stream - the memcpy of the next piece runs
// beside the DMA of the last one, and the lanes beside each other.  Small jobs take hipMemcpy.  Returns with every
// byte on the device.
struct UpJob {
    void *dst;
    const void *src;
    size_t bytes;
};
constexpr size_t UP_PIECE = (size_t)4 << 20;
int upload_staged(adh_handle *h, const std::vector<UpJob> &jobs) {
    size_t total = 0;
    for (const UpJob &j : jobs) total += j.bytes;
    int lanes = 4;
    if (const char *env = getenv("ADH_UPLOAD_LANES")) lanes = atoi(env);  // (0: the runtime's pageable copy)
    {
        int ranks = 1;
        if (const char *lw = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(atoi(lw), 1);
        lanes = std::min<int>(lanes, std::max<int>(host_cpu_budget() / ranks, 1));
    }
    lanes = std::min(lanes, 16);
    size_t least = (size_t)16 << 20;  // (below this the runtime's copy is as fast; ADH_UPLOAD_MIN_MB: how the tests get here)
    if (const char *env = getenv("ADH_UPLOAD_MIN_MB")) least = (size_t)atoll(env) << 20;
    if (lanes <= 0 || total < std::max<size_t>(least, 1)) {
        for (const UpJob &j : jobs)
            if (j.bytes) HIP_TRY(hipMemcpy(j.dst, j.src, j.bytes, hipMemcpyHostToDevice));
        return ADH_OK;
    }
    while ((int)h->up_lanes.size() < lanes) {
        adh_handle::UpLane l;
        hipError_t e = hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking);
        for (int k = 0; k < 2 && e == hipSuccess; ++k) {
            e = hipHostMalloc(&l.buf[k], UP_PIECE, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&l.ev[k], hipEventDisableTiming);
        }
        if (e != hipSuccess) {  // (no page-locked memory to be had: fewer lanes, or the runtime's copy)
            for (int k = 0; k < 2; ++k) {
                if (l.buf[k]) (void)hipHostFree(l.buf[k]);
                if (l.ev[k]) (void)hipEventDestroy(l.ev[k]);
            }
            if (l.st) (void)hipStreamDestroy(l.st);
            (void)hipGetLastError();
            lanes = (int)h->up_lanes.size();
            break;
        }
        h->up_lanes.push_back(l);
    }
    if (lanes == 0) {
        for (const UpJob &j : jobs)
            if (j.bytes) HIP_TRY(hipMemcpy(j.dst, j.src, j.bytes, hipMemcpyHostToDevice));
        return ADH_OK;
    }
    std::atomic<int> err{(int)hipSuccess};
    auto work = [&](int t) {
        adh_handle::UpLane &l = h->up_lanes[(size_t)t];
        hipError_t e = hipSetDevice(h->device);
        bool used[2] = {false, false};
        int k = 0;
        for (const UpJob &j : jobs) {
            // slices on 4 KB borders
            const size_t per = ((j.bytes + (size_t)lanes - 1) / (size_t)lanes + 4095) & ~(size_t)4095;
            const size_t a = std::min(j.bytes, per * (size_t)t), b = std::min(j.bytes, a + per);
            for (size_t o = a; o < b && e == hipSuccess; o += UP_PIECE, k ^= 1) {
                const size_t n = std::min(UP_PIECE, b - o);
                if (used[k]) e = hipEventSynchronize(l.ev[k]);
                if (e != hipSuccess) break;
                memcpy(l.buf[k], static_cast<const char *>(j.src) + o, n);
                e = hipMemcpyAsync(static_cast<char *>(j.dst) + o, l.buf[k], n, hipMemcpyHostToDevice, l.st);
                if (e == hipSuccess) e = hipEventRecord(l.ev[k], l.st);
                used[k] = true;
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(l.st);
        if (e != hipSuccess) err.store((int)e);
    };
    std::vector<std::thread> team;
    for (int t = 1; t < lanes; ++t) team.emplace_back(work, t);
    work(0);
    for (std::thread &t : team) t.join();
    if (err.load() != (int)hipSuccess)
        return fail(ADH_ERR_HIP, std::string("upload_staged: ") + hipGetErrorString((hipError_t)err.load()));
    return ADH_OK;
}

#define UP(owner, host, n, dev)                                            \
    do {                                                                   \
        int _rc = upload(owner, host, n, dev, h->stream);                  \
        if (_rc != ADH_OK) return _rc;                                     \
    } while (0)

int get_event(adh_handle *h, hipEvent_t *e) {
    if (!h->free_events.empty()) {
        *e = h->free_events.back();
        h->free_events.pop_back();
        return ADH_OK;
    }
    HIP_TRY(hipEventCreate(e));
    return ADH_OK;
}

}  // namespace

extern "C" {

namespace {
// adh_score_host.hip: fills the columns a host -> host call left out of the device tables.  They are rebuilt
// from the candidate table and the library that are in HBM at that moment, so every entry point that
// replaces either (or the run) settles the tables of the previous call first.
int materialise_tables(adh_handle *h);
}  // namespace

const char *adh_last_error(void) { return g_last_error.c_str(); }

int adh_device_count(int *count) {
    if (!count) return fail(ADH_ERR_INVALID_ARGUMENT, "count is NULL");
    HIP_TRY(hipGetDeviceCount(count));
    return ADH_OK;
}

int adh_create(adh_handle_t **handle, int device) {
    if (!handle) return fail(ADH_ERR_INVALID_ARGUMENT, "handle is NULL");
    *handle = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n)
        return fail(ADH_ERR_INVALID_ARGUMENT, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    adh_handle *h = new adh_handle();
    h->device = device;
    const char *what = "hipStreamCreate";
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream_out, hipStreamNonBlocking);
    if (e == hipSuccess) what = "hipEventCreate";
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_k[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_k[1], hipEventDisableTiming);
    if (e == hipSuccess) what = "hipMalloc(weight table)";
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_wtp, 2 * 64 * sizeof(double));
    if (e == hipSuccess) {
        what = "adh_wtp_table_kernel";
        hipLaunchKernelGGL(adh_wtp_table_kernel, dim3(1), dim3(128), 0, h->stream, h->d_wtp);
        e = hipGetLastError();
        // callers may score on a stream of their own: the table must be complete before the handle is used
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        const std::string msg = std::string(what) + ": " + hipGetErrorString(e);
        adh_destroy(h);  // releases whatever was created so far
        return fail(e == hipErrorOutOfMemory ? ADH_ERR_OUT_OF_MEMORY : ADH_ERR_HIP, msg);
    }
    // the kernels may need more than the default 64 KiB of dynamic LDS
    (void)hipFuncSetAttribute((const void *)adh_feature_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)adh_gather_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // (this kernel also has ADH_IM_STATIC_LDS bytes of static LDS)
    (void)hipFuncSetAttribute((const void *)adh_feature_im_kernel<featim::Layout>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - ADH_IM_STATIC_LDS);
    (void)hipFuncSetAttribute((const void *)adh_feature_im_kernel<featim::LayoutCommon>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - ADH_IM_STATIC_LDS);
    (void)hipFuncSetAttribute((const void *)adh_feature_im_kernel<featim::LayoutSmall>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - ADH_IM_STATIC_LDS);
    (void)hipFuncSetAttribute((const void *)adh_feature_im_kernel<featim::LayoutCommon, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - ADH_IM_STATIC_LDS);
    (void)hipFuncSetAttribute((const void *)adh_feature_im_kernel<featim::LayoutSmall, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - ADH_IM_STATIC_LDS);
    (void)hipFuncSetAttribute((const void *)adh_feature_im_kernel<featim::LayoutCommon2, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - ADH_IM_STATIC_LDS);
    (void)hipFuncSetAttribute((const void *)adh_gather_im_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    *handle = h;
    return ADH_OK;
}

int adh_destroy(adh_handle_t *h) {
    if (!h) return ADH_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    (void)adh_comm_destroy(h);
    h->run_buf.release();
    h->lib_buf.release();
    for (PlanSlot &s : h->slots) {
        s.buf.release();
        if (s.hist) (void)hipFree(s.hist);
        if (s.h_meta) (void)hipHostFree(s.h_meta);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (auto &p : h->timed) {
        (void)hipEventDestroy(p.e0);
        (void)hipEventDestroy(p.e1);
        (void)hipEventDestroy(p.e2);
    }
    for (auto e : h->free_events) (void)hipEventDestroy(e);
    if (h->d_wtp) (void)hipFree(h->d_wtp);
    if (h->slot_stage) (void)hipHostFree(h->slot_stage);
    if (h->cmp_host) (void)hipHostFree(h->cmp_host);
    if (h->cop_stage) (void)hipHostFree(h->cop_stage);
    if (h->cop_cnt) (void)hipFree(h->cop_cnt);
    if (h->cop_dev) (void)hipFree(h->cop_dev);
    if (h->cop_tot_pinned) (void)hipHostFree(h->cop_tot_pinned);
    for (adh_handle::UpLane &l : h->up_lanes) {
        for (int k = 0; k < 2; ++k) {
            if (l.buf[k]) (void)hipHostFree(l.buf[k]);
            if (l.ev[k]) (void)hipEventDestroy(l.ev[k]);
        }
        if (l.st) (void)hipStreamDestroy(l.st);
    }
    if (h->cop_scan) (void)hipFree(h->cop_scan);
    if (h->cmp_dev) (void)hipFree(h->cmp_dev);
    if (h->cmp_scan) (void)hipFree(h->cmp_scan);
    for (DevTables &t : h->tables)
        if (t.base) (void)hipFree(t.base);
    if (h->cs.base) (void)hipFree(h->cs.base);
    if (h->scratch_slab) (void)hipFree(h->scratch_slab);
    if (h->sel_slab) (void)hipFree(h->sel_slab);
    for (hipStream_t st : h->stream_aux)
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t e : h->ev_aux)
        if (e) (void)hipEventDestroy(e);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    for (hipEvent_t e : h->ev_k)
        if (e) (void)hipEventDestroy(e);
    if (h->stream_in) (void)hipStreamDestroy(h->stream_in);
    if (h->stream_out) (void)hipStreamDestroy(h->stream_out);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    (void)hipGetLastError();
    delete h;
    return ADH_OK;
}

namespace {

inline uint32_t float_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

// Sort the peaks of the run into the transposed order and build entries + bin table, slab by slab: a slab is a
// range of whole groups of blocks (= a range of spectra = a range of peaks, spectra being stored in time order)
// with fewer than 2^31 peaks, so that the radix sort and the peak numbers inside it stay 32-bit while the run as
// a whole may hold 2^32 peaks and more (round 4).  grp_entry0[g] = first entry of group g (host prefix sums).
int stage_transposed(adh_handle *h, const adh_alpharaw_t *d, const DevRun &r, int64_t n_ref, int64_t n_tab,
                     const std::vector<int64_t> &grp_entry0, const int64_t *d_grp_entry0, uint2 *entries, uint32_t *tab,
                     DeviceBuffers &tmp) {
    hipStream_t st = h->stream;
    const int64_t L = d->cycle_len;
    const int64_t n_groups = (int64_t)grp_entry0.size() - 1;
    const int64_t words_per_group = L * (int64_t)r.n_bins * ADH_SUB;
    HIP_TRY(hipMemsetAsync(tab, 0, (size_t)n_tab * sizeof(uint32_t), st));
    if (n_ref == 0 || d->n_spectra == 0) {
        HIP_TRY(hipStreamSynchronize(st));
        return ADH_OK;
    }
    const int64_t spectra_per_group = L << (r.block_shift + ADH_SUB_SHIFT);
    // peak range of a group: first referenced peak of its first spectrum .. last of its last (spectra do not share
    // peaks and ascend, checked by the caller); unreferenced peaks in between get the all-ones key and sort last
    auto group_peaks = [&](int64_t g0, int64_t g1, int64_t &p0, int64_t &p1) {
        const int64_t s0 = std::min(g0 * spectra_per_group, d->n_spectra), s1 = std::min(g1 * spectra_per_group, d->n_spectra);
        p0 = p1 = 0;
        bool any = false;
        for (int64_t sp = s0; sp < s1; ++sp)
            if (d->peak_stop_idx[sp] > d->peak_start_idx[sp]) {
                if (!any) p0 = d->peak_start_idx[sp];
                p1 = d->peak_stop_idx[sp];
                any = true;
            }
    };
    int64_t slab_cap = (int64_t)0x7FFFFFFFll - 1;
    if (const char *env = getenv("ADH_STAGE_SLAB_PEAKS")) slab_cap = std::max<int64_t>(atoll(env), 1);  // (tests)
    // slabs of whole groups within the cap (a single group above it is refused)
    std::vector<int64_t> cut{0};
    while (cut.back() < n_groups) {
        int64_t g1 = cut.back() + 1, p0, p1;
        group_peaks(cut.back(), g1, p0, p1);
        if (p1 - p0 >= (int64_t)0x7FFFFFFFll)
            return fail(ADH_ERR_UNSUPPORTED, "one group of cycle blocks holds 2^31 peaks or more");
        while (g1 < n_groups) {
            int64_t q0, q1;
            group_peaks(cut.back(), g1 + 1, q0, q1);
            if (q1 - q0 > slab_cap) break;
            ++g1;
        }
        cut.push_back(g1);
    }
    int64_t max_peaks = 1, max_spec = 1;
    for (size_t sl = 0; sl + 1 < cut.size(); ++sl) {
        int64_t p0, p1;
        group_peaks(cut[sl], cut[sl + 1], p0, p1);
        max_peaks = std::max(max_peaks, p1 - p0);
        max_spec = std::max(max_spec, std::min(cut[sl + 1] * spectra_per_group, d->n_spectra) - std::min(cut[sl] * spectra_per_group, d->n_spectra));
    }
    float *d_mz = nullptr, *d_int = nullptr;
    int64_t *d_ps = nullptr, *d_pe = nullptr;
    uint64_t *k_in = nullptr, *k_out = nullptr;
    uint32_t *v_in = nullptr, *v_out = nullptr;
    int *d_bad = nullptr;
    void *sort_tmp = nullptr;
    int rc;
    auto dev_alloc = [&](void **p, size_t bytes) -> int {
        HIP_TRY(hipMalloc(p, std::max<size_t>(bytes, 16)));
        tmp.ptrs.push_back(*p);
        return ADH_OK;
    };
    if ((rc = dev_alloc((void **)&d_mz, (size_t)max_peaks * 4)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&d_int, (size_t)max_peaks * 4)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&d_ps, (size_t)max_spec * 8)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&d_pe, (size_t)max_spec * 8)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&k_in, (size_t)max_peaks * 8)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&k_out, (size_t)max_peaks * 8)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&v_in, (size_t)max_peaks * 4)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&v_out, (size_t)max_peaks * 4)) != ADH_OK) return rc;
    if ((rc = dev_alloc((void **)&d_bad, 4)) != ADH_OK) return rc;
    size_t sort_bytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, k_in, k_out, v_in, v_out, (int)max_peaks, 0, 64, st));
    if ((rc = dev_alloc(&sort_tmp, std::max<size_t>(sort_bytes, 16))) != ADH_OK) return rc;
    HIP_TRY(hipMemsetAsync(d_bad, 0, 4, st));
    for (size_t sl = 0; sl + 1 < cut.size(); ++sl) {
        const int64_t g0 = cut[sl], g1 = cut[sl + 1];
        const int64_t s0 = std::min(g0 * spectra_per_group, d->n_spectra), s1 = std::min(g1 * spectra_per_group, d->n_spectra);
        int64_t p0, p1;
        group_peaks(g0, g1, p0, p1);
        const int64_t np = p1 - p0, ns = s1 - s0, n_slab_ref = grp_entry0[(size_t)g1] - grp_entry0[(size_t)g0];
        const int64_t g_first = g0 * words_per_group, g_last = g1 * words_per_group - 1;
        if (np > 0 && ns > 0) {
            HIP_TRY(hipMemcpyAsync(d_mz, d->mz_values + p0, (size_t)np * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d_int, d->intensity_values + p0, (size_t)np * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d_ps, d->peak_start_idx + s0, (size_t)ns * 8, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(d_pe, d->peak_stop_idx + s0, (size_t)ns * 8, hipMemcpyHostToDevice, st));
            // peaks no spectrum refers to keep the all-ones key and sort to the end of the slab
            HIP_TRY(hipMemsetAsync(k_in, 0xFF, (size_t)np * 8, st));
            HIP_TRY(hipMemsetAsync(v_in, 0, (size_t)np * 4, st));
            hipLaunchKernelGGL(adh_peak_key_kernel, dim3((unsigned)ns), dim3(256), 0, st, d_mz, d_ps, d_pe, s0, ns, p0,
                               (int)d->cycle_len, (int)r.block_shift, (int)r.bin0, (int)r.n_bins, k_in, v_in, d_bad);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, k_in, k_out, v_in, v_out, (int)np, 0, 64, st));
        }
        hipLaunchKernelGGL(adh_entries_kernel, dim3(8192), dim3(256), 0, st, k_out, v_out, d_int, n_slab_ref,
                           grp_entry0[(size_t)g0], entries, tab, g_first, g_last, words_per_group, d_grp_entry0, (int)r.block_shift);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));  // (the slab buffers are reused)
    }
    hipLaunchKernelGGL(adh_tab_spare_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, st, tab, n_groups,
                       words_per_group, d_grp_entry0);
    HIP_TRY(hipGetLastError());
    int bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (bad) return fail(ADH_ERR_INVALID_ARGUMENT, "m/z values are not ascending inside every spectrum");
    return ADH_OK;
}

}  // namespace

namespace {
// (round 4 marked the handle here - a staging call frees its temporaries and what was staged before; they are parked
// now: DevBlockCache above)
void note_staged(adh_handle *h, uint64_t bytes) {
    (void)h;
    (void)bytes;
}
}  // namespace

int adh_stage_alpharaw(adh_handle_t *h, const adh_alpharaw_t *d) {
    if (!h || !d) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (d->cycle_len <= 0 || d->cycle_scans <= 0 || d->n_spectra < 0 || d->n_peaks < 0)
        return fail(ADH_ERR_INVALID_ARGUMENT, "invalid run dimensions");
    if (d->cycle_scans != 1)
        return fail(ADH_ERR_UNSUPPORTED,
                    "cycle with a scan axis (ion mobility) is not an AlphaRaw run");
    if (d->n_mobility < 1) return fail(ADH_ERR_INVALID_ARGUMENT, "mobility_values is empty");
    if (d->n_spectra >= (int64_t)0x7FFFFFFFll || d->cycle_len > 65535)
        return fail(ADH_ERR_UNSUPPORTED, "too many spectra / cycle positions");
    HIP_TRY(hipSetDevice(h->device));
    {
        const int rc_m = materialise_tables(h);  // while the candidate table the last call scored is still the resident one
        if (rc_m != ADH_OK) return rc_m;
    }
    HIP_TRY(hipDeviceSynchronize());
    h->run_buf.release();
    h->plan = Plan();
    h->cands_uploaded = false;  // the resident table was planned against the previous run
    h->run_staged = false;
    h->tims_staged = false;

    // validate the CSR on the host: kernels index with it unchecked
    float mz_lo = 0.f, mz_hi = 0.f;
    bool any = false;
    int64_t n_ref = 0, prev_stop = 0;
    for (int64_t s = 0; s < d->n_spectra; ++s) {
        int64_t a = d->peak_start_idx[s], b = d->peak_stop_idx[s];
        if (a < 0 || b < a || b > d->n_peaks)
            return fail(ADH_ERR_INVALID_ARGUMENT, "peak_start/stop_idx out of range");
        if (b > a) {
            if (a < prev_stop)
                return fail(ADH_ERR_INVALID_ARGUMENT, "spectra must not share peaks (peak_start_idx < previous peak_stop_idx)");
            prev_stop = b;
            n_ref += b - a;
            float lo = d->mz_values[a], hi = d->mz_values[b - 1];
            if (!any || lo < mz_lo) mz_lo = lo;
            if (!any || hi > mz_hi) mz_hi = hi;
            any = true;
        }
    }
    if (any && !(mz_lo > 0.f && mz_hi >= mz_lo && mz_hi < INFINITY))
        return fail(ADH_ERR_INVALID_ARGUMENT, "m/z values must be positive, finite and ascending inside a spectrum");
    std::vector<int32_t> ms1;
    for (int r = 0; r < d->cycle_len * d->cycle_scans; ++r)
        if (-1.0 <= d->cycle[2 * r + 1] && -1.0 >= d->cycle[2 * r]) ms1.push_back(r);

    DevRun r{};
    r.n_spectra = d->n_spectra;
    r.n_peaks = d->n_peaks;
    r.cycle_len = d->cycle_len;
    r.cycle_scans = d->cycle_scans;
    r.n_ms1_obs = (int32_t)ms1.size();
    UP(h->run_buf, d->rt_values, d->n_spectra, &r.rt);
    UP(h->run_buf, d->mobility_values, d->n_mobility, &r.mobility);
    UP(h->run_buf, d->cycle, (int64_t)d->cycle_len * d->cycle_scans * 2, &r.cycle);
    UP(h->run_buf, ms1.data(), (int64_t)ms1.size(), &r.ms1_obs);
    h->h_cycle.assign(d->cycle, d->cycle + (size_t)d->cycle_len * d->cycle_scans * 2);
    h->h_rt.assign(d->rt_values, d->rt_values + d->n_spectra);

    // ---- transposed run (see adh_gather.hip): bins, block size, table size
    if (!any) mz_lo = mz_hi = 1.0f;
    r.mz_min = mz_lo;
    r.mz_max = mz_hi;
    r.bin0 = (int32_t)(float_bits(mz_lo) >> ADH_BIN_SHIFT);
    r.n_bins = (int32_t)(float_bits(mz_hi) >> ADH_BIN_SHIFT) - r.bin0 + 1;
    const int64_t L = d->cycle_len;
    const int64_t n_cycles = (d->n_spectra + L - 1) / L;
    // about one entry per two (block, row, bin) cells: B ~ n_bins / (2 * peaks per spectrum).
    // Measured on the bench run: gather 0.88 / 0.92 / 1.06 / 1.34 ms for B = 16 / 32 / 64 / 128.
    int64_t avg = d->n_spectra > 0 ? std::max<int64_t>(n_ref / d->n_spectra, 1) : 1;
    int64_t want = (int64_t)r.n_bins / (2 * avg);
    if (const char *env = getenv("ADH_BLOCK_CYCLES")) want = atoll(env);
    int bs = 3;
    while (bs < 20 && (1ll << (bs + 1)) <= want) ++bs;
    // the table must stay addressable with 32-bit bin ids and should not dwarf the peaks
    for (;; ++bs) {
        int64_t nblk = std::max<int64_t>((n_cycles + (1ll << bs) - 1) >> bs, 1);
        nblk = (nblk + ADH_SUB - 1) / ADH_SUB * ADH_SUB;  // whole groups of blocks (adh_device.h)
        int64_t n_tab = nblk * L * (int64_t)r.n_bins + nblk / ADH_SUB;
        if (bs >= 20 || (n_tab < (int64_t)0xFFFFFFF0ll && n_tab * 4 <= std::max<int64_t>(2 * n_ref * 8, 64ll << 20))) {
            r.n_blocks = (int32_t)nblk;
            break;
        }
    }
    r.block_shift = bs;
    // one word per (group, row, bin, block) + one spare word per group (adh_tab_row)
    const int64_t n_groups = r.n_blocks / ADH_SUB;
    const int64_t n_tab = (int64_t)r.n_blocks * L * (int64_t)r.n_bins + n_groups;
    if (n_tab >= (int64_t)0xFFFFFFF0ll)
        return fail(ADH_ERR_UNSUPPORTED, "m/z range x cycle positions too large for the bin table");
    // first entry of every group of blocks: the referenced peaks of the spectra before it
    std::vector<int64_t> grp_entry0((size_t)n_groups + 1, 0);
    {
        const int64_t spectra_per_group = L << (bs + ADH_SUB_SHIFT);
        int64_t acc = 0;
        for (int64_t g = 0; g < n_groups; ++g) {
            grp_entry0[(size_t)g] = acc;
            const int64_t s0 = std::min(g * spectra_per_group, d->n_spectra), s1 = std::min((g + 1) * spectra_per_group, d->n_spectra);
            for (int64_t sp = s0; sp < s1; ++sp) acc += d->peak_stop_idx[sp] - d->peak_start_idx[sp];
            if (acc - grp_entry0[(size_t)g] >= (int64_t)0xFFFFFFFFll)
                return fail(ADH_ERR_UNSUPPORTED, "one group of cycle blocks holds 2^32 peaks or more");
        }
        grp_entry0[(size_t)n_groups] = acc;
    }
    UP(h->run_buf, grp_entry0.data(), n_groups + 1, &r.grp_entry0);

    uint2 *entries = nullptr;
    uint32_t *tab = nullptr;
    HIP_TRY(hipMalloc((void **)&entries, (size_t)(std::max<int64_t>(n_ref, 1) + 4) * sizeof(uint2)));  // (+4: the fused kernel reads entries in pairs and table words in fours, see adh_fused.hip)
    h->run_buf.ptrs.push_back(entries);
    HIP_TRY(hipMalloc((void **)&tab, (size_t)(n_tab + 4) * sizeof(uint32_t)));
    h->run_buf.ptrs.push_back(tab);
    r.entries = entries;
    r.tab = tab;

    DeviceBuffers tmp;
    int rc = stage_transposed(h, d, r, n_ref, n_tab, grp_entry0, r.grp_entry0, entries, tab, tmp);
    tmp.release();
    if (rc != ADH_OK) return rc;
    h->run = r;
    h->run_staged = true;
    note_staged(h, (uint64_t)d->n_peaks * 16);  // (the sort's keys and values are the largest temporaries)
    return ADH_OK;
}

namespace {
// The tile-ordered copy of the events for the scoring gather (DevTims::tile_ev, adh_device.h): keys, one stable
// radix sort, the exclusive scan of the key histogram as the index.  Built when the run is below 2^31 events and
// keys, values and index fit a quarter of the free device memory; ADH_IM_TILED=0 leaves it out,
// ADH_IM_TILE_SHIFTS="c,s" fixes the tile shape (2^c cycles x 2^s scans; default: about 1.5 events per (tile,
// TOF bin), i.e. a run of ~70 bytes for the handful of bins of a 15 ppm window, s = c or c + 1: on configs[3]
// 32 cycles x 64 scans, measured against 16 x 32 ... 128 x 64).
int build_tile_layout(adh_handle *h, DevTims &t) {
    t.tile_ev = nullptr;
    t.tile_idx = nullptr;
    const char *env = getenv("ADH_IM_TILED");
    if ((env && atoi(env) == 0) || t.n_events <= 0 || t.n_events >= 0x7FFFFFFFll || t.n_cycles <= 0) return ADH_OK;
    int sbits = 1;  // (an event of the layout holds frame << sbits | scan in 32 bits)
    while ((1ll << sbits) < (int64_t)t.scan_max) ++sbits;
    if (sbits >= 31 || t.n_frames > (1ll << (32 - sbits))) return ADH_OK;
    int csh = -1, ssh = -1;
    if (const char *sh = getenv("ADH_IM_TILE_SHIFTS")) {
        if (sscanf(sh, "%d,%d", &csh, &ssh) != 2 || csh < 0 || ssh < 0 || csh > 20 || ssh > 20)
            return fail(ADH_ERR_INVALID_ARGUMENT, "ADH_IM_TILE_SHIFTS must be \"c,s\" with 0 <= c, s <= 20");
    }
    auto blocks = [](int64_t n, int sh) { return (n + (1ll << sh) - 1) >> sh; };
    // Round 6: the frame of the cycle as part of the tile (DevTims::tile_frames).  A candidate's windows live in one or two
    // of the cycle's frames, so with the frames apart it streams a cycle_len-th of the events per (cycles x scans) box;
    // the index grows by the same factor - affordable where HBM is 288 GB (ADH_IM_TILE_FRAMES=0: frames share a tile, the
    // layout of round 4; cycles of more than 32 frames keep that as well: the plan names a candidate's frames by a mask)
    const char *fenv = getenv("ADH_IM_TILE_FRAMES");
    int64_t n_fr = (t.cycle_len <= 32 && !(fenv && atoi(fenv) == 0)) ? t.cycle_len : 1;
    if (csh < 0) {
        // ~1.5 events per (tile, TOF bin) where the frames share a tile; keyed by frame the boxes stay as they are and
        // the tile count grows by the frames (capped below by the 2^31 keys of the index)
        const double want = (double)t.n_events / (1.5 * (double)t.n_tof) * (double)n_fr;  // tiles
        double best = 1e300;
        for (int c = 2; c <= 12; ++c)
            for (int s = c; s <= c + (n_fr > 1 ? 2 : 1); ++s) {
                const double tiles = (double)(blocks(t.n_cycles, c) * blocks(t.scan_max, s) * n_fr);
                if (tiles * (double)(t.n_tof + 1) >= 2.0e9) continue;
                const double miss = fabs(log(std::max(tiles, 1.0) / std::max(want, 1.0)));
                if (miss < best) best = miss, csh = c, ssh = s;
            }
        if (csh < 0) csh = 12, ssh = 14;
    }
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    int64_t ncb = blocks(t.n_cycles, csh), nsb = blocks(t.scan_max, ssh);
    // (coarser tiles until the index fits 32-bit keys and its share of the memory)
    while ((double)ncb * (double)nsb * (double)n_fr * (double)(t.n_tof + 1) >= 2.0e9 ||
           (size_t)(ncb * nsb * n_fr * (t.n_tof + 1) + 1) * 4 > free_b / 8) {
        if (ncb == 1 && nsb == 1) {
            if (n_fr == 1) return ADH_OK;
            n_fr = 1;  // (no room for the frames: share the tiles)
            continue;
        }
        if (ncb >= nsb) ++csh; else ++ssh;
        ncb = blocks(t.n_cycles, csh), nsb = blocks(t.scan_max, ssh);
    }
    const int64_t n_tiles = ncb * nsb * n_fr;
    const int64_t n_keys = n_tiles * (t.n_tof + 1), n = t.n_events;
    if (n_keys >= 0x7FFFFFFFll) return ADH_OK;  // (the scan below counts in int: such a run keeps the bin ranges)
    int key_bits = 1;
    while (key_bits < 32 && (1ll << key_bits) < n_keys) ++key_bits;
    size_t sort_bytes = 0, scan_bytes = 0;
    uint32_t *k_in = nullptr, *k_out = nullptr, *idx = nullptr;
    uint64_t *v_in = nullptr, *v_out = nullptr;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, k_in, k_out, v_in, v_out, (int)n, 0, key_bits, h->stream));
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, idx, idx, (int)(n_keys + 1), h->stream));
    const size_t tmp_bytes = std::max(sort_bytes, scan_bytes);
    if ((size_t)n * 24 + (size_t)(n_keys + 1) * 4 + tmp_bytes > free_b / 4) return ADH_OK;  // no room: bin ranges
    DeviceBuffers work;  // (released on every way out)
    void *tmp = nullptr;
    auto grab = [&](void **p, size_t bytes, DeviceBuffers &owner) {
        if (hipMalloc(p, std::max<size_t>(bytes, 16)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        owner.ptrs.push_back(*p);
        return true;
    };
    if (!grab((void **)&k_in, (size_t)n * 4, work) || !grab((void **)&k_out, (size_t)n * 4, work) ||
        !grab((void **)&v_in, (size_t)n * 8, work) || !grab(&tmp, tmp_bytes, work) ||
        !grab((void **)&v_out, (size_t)n * 8, work) || !grab((void **)&idx, (size_t)(n_keys + 1) * 4, work)) {
        work.release();  // (v_out / idx included: a layout that was not built must not keep gigabytes)
        return ADH_OK;
    }
    // built from here on: the two arrays stay with the run
    for (void *keep : {(void *)v_out, (void *)idx}) {
        work.ptrs.erase(std::find(work.ptrs.begin(), work.ptrs.end(), keep));
        h->run_buf.ptrs.push_back(keep);
    }
    hipError_t e = hipMemsetAsync(idx, 0, (size_t)(n_keys + 1) * 4, h->stream);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)std::min<int64_t>(t.n_tof, 1 << 20);
        hipLaunchKernelGGL(adh_tile_key_kernel, dim3(grid), dim3(ADH_WAVE), 0, h->stream, t.tof_indptr, t.push, t.inten, t.n_tof,
                           (uint32_t)t.scan_max, (uint32_t)t.cycle_len, (uint32_t)t.zeroth, csh, ssh, sbits, (uint32_t)ncb, (uint32_t)nsb,
                           (uint32_t)n_fr, k_in, v_in, idx);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(tmp, sort_bytes, k_in, k_out, v_in, v_out, (int)n, 0, key_bits, h->stream);
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, scan_bytes, idx, idx, (int)(n_keys + 1), h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    work.release();
    if (e != hipSuccess) return fail(ADH_ERR_HIP, hipGetErrorString(e));
    t.tile_ev = reinterpret_cast<const uint2 *>(v_out);
    t.tile_idx = idx;
    t.tile_cshift = csh;
    t.tile_sshift = ssh;
    t.tile_cblocks = (int32_t)ncb;
    t.tile_sblocks = (int32_t)nsb;
    t.tile_sbits = sbits;
    t.tile_frames = (int32_t)n_fr;
    return ADH_OK;
}
}  // namespace

int adh_stage_timstof(adh_handle_t *h, const adh_timstof_t *d) {
    if (!h || !d) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (d->cycle_len <= 0 || d->scan_max_index <= 0 || d->n_frames <= 0 || d->n_tof <= 0 || d->n_events < 0)
        return fail(ADH_ERR_INVALID_ARGUMENT, "invalid run dimensions");
    if (d->cycle_len > 65535 || d->n_tof >= 0x7FFFFFFFll ||
        d->n_frames * (int64_t)d->scan_max_index >= 0xFFFFFFFFll)
        return fail(ADH_ERR_UNSUPPORTED, "run too large for 32-bit push / TOF indices");
    HIP_TRY(hipSetDevice(h->device));
    {
        const int rc_m = materialise_tables(h);  // while the candidate table the last call scored is still the resident one
        if (rc_m != ADH_OK) return rc_m;
    }
    HIP_TRY(hipDeviceSynchronize());
    h->run_buf.release();
    h->plan = Plan();
    h->cands_uploaded = false;  // the resident table was planned against the previous run
    h->run_staged = false;
    h->tims_staged = false;
    // validate the index arrays on the host: kernels use them unchecked
    if (d->tof_indptr[0] != 0 || d->tof_indptr[d->n_tof] != d->n_events)
        return fail(ADH_ERR_INVALID_ARGUMENT, "tof_indptr does not span the event arrays");
    for (int64_t t = 0; t < d->n_tof; ++t) {
        if (d->tof_indptr[t + 1] < d->tof_indptr[t])
            return fail(ADH_ERR_INVALID_ARGUMENT, "tof_indptr is not monotone");
        if (t > 0 && !(d->mz_values[t] >= d->mz_values[t - 1]))
            return fail(ADH_ERR_INVALID_ARGUMENT, "mz_values must be ascending");
    }
    const int64_t rows = (int64_t)d->cycle_len * d->scan_max_index;
    std::vector<int32_t> dpc((size_t)rows);
    for (int64_t i = 0; i < rows; ++i) {
        if (d->dia_precursor_cycle[i] < 0 || d->dia_precursor_cycle[i] >= d->cycle_len)
            return fail(ADH_ERR_INVALID_ARGUMENT, "dia_precursor_cycle must index the cycle");
        dpc[(size_t)i] = (int32_t)d->dia_precursor_cycle[i];
    }
    DevTims t{};
    t.n_tof = d->n_tof;
    t.n_events = d->n_events;
    t.n_frames = d->n_frames;
    t.cycle_len = d->cycle_len;
    t.scan_max = d->scan_max_index;
    t.zeroth = d->zeroth_frame ? 1 : 0;
    UP(h->run_buf, d->tof_indptr, d->n_tof + 1, &t.tof_indptr);
    UP(h->run_buf, d->push_indices, d->n_events, &t.push);
    UP(h->run_buf, d->intensity_values, d->n_events, &t.inten);
    UP(h->run_buf, d->mz_values, d->n_tof, &t.mz);
    UP(h->run_buf, d->cycle, rows * 2, &t.cycle);
    UP(h->run_buf, dpc.data(), rows, &t.dpc);
    UP(h->run_buf, d->rt_values, d->n_frames, &t.rt);
    UP(h->run_buf, d->mobility_values, (int64_t)d->scan_max_index, &t.mobility);
    {
        // search indices (DevTims): the m/z table on the host, the (bin, cycle) table on the device
        t.n_cycles = (int32_t)((d->n_frames - t.zeroth + d->cycle_len - 1) / d->cycle_len);
        const char *env = getenv("ADH_IM_INDEX");
        // (event numbers are 64-bit everywhere; the index columns and the pair ranges of the kernels count from the
        // first event of a TOF bin / of a window's first bin in 32 bits: a single bin must stay below 2^32 events)
        int64_t widest = 0;
        for (int64_t b = 0; b < d->n_tof; ++b) widest = std::max(widest, d->tof_indptr[b + 1] - d->tof_indptr[b]);
        if (widest >= (int64_t)0xFFFFFFFFll)
            return fail(ADH_ERR_UNSUPPORTED, "a TOF bin with 2^32 or more events");
        const bool want = !(env && atoi(env) == 0) && d->n_tof >= 2 && t.n_cycles > 0 &&
                          d->mz_values[d->n_tof - 1] > d->mz_values[0];
        if (want) {
            int64_t nb = 1;
            while (nb < 4 * d->n_tof) nb <<= 1;
            const double lo = d->mz_values[0], hi = d->mz_values[d->n_tof - 1];
            std::vector<uint32_t> lut((size_t)nb + 1);
            int64_t pos = 0;
            const double step = (hi - lo) / (double)nb;
            for (int64_t b = 0; b <= nb; ++b) {
                const double x = lo + (double)b * step;
                while (pos < d->n_tof && d->mz_values[pos] < x) ++pos;
                lut[(size_t)b] = (uint32_t)pos;
            }
            UP(h->run_buf, lut.data(), nb + 1, &t.mz_lut);
            t.lut_min = lo;
            t.lut_inv_step = 1.0 / step;
            t.lut_n = (int32_t)nb;
            // 4 bytes per (bin, cycle block): the finest block that fits an eighth of the device memory
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            size_t budget = std::min(total_b / 8, free_b / 2);
            if (const char *mb = getenv("ADH_IM_INDEX_MB")) budget = (size_t)atoll(mb) << 20;
            int shift = 0;
            int64_t blocks = t.n_cycles;
            while (shift < 16 && (size_t)d->n_tof * (size_t)(blocks + 1) * 4 > budget) {
                ++shift;
                blocks = ((int64_t)t.n_cycles + (1ll << shift) - 1) >> shift;
            }
            if ((size_t)d->n_tof * (size_t)(blocks + 1) * 4 <= budget) {
                uint32_t *idx = nullptr;
                if (hipMalloc((void **)&idx, (size_t)d->n_tof * (size_t)(blocks + 1) * 4) != hipSuccess) {
                    (void)hipGetLastError();  // no room for the index: the kernels search instead
                    idx = nullptr;
                }
                if (idx) {
                h->run_buf.ptrs.push_back(idx);
                const unsigned grid = (unsigned)std::min<int64_t>(d->n_tof, 1 << 20);
                hipLaunchKernelGGL(adh_index_im_kernel, dim3(grid), dim3(ADH_WAVE), 0, h->stream, t.tof_indptr, t.push,
                                   t.n_tof, (uint32_t)t.scan_max, (uint32_t)t.cycle_len, (uint32_t)t.zeroth, shift,
                                   (uint32_t)blocks, idx);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(h->stream));
                t.cyc_idx = idx;
                t.cyc_shift = shift;
                t.cyc_cols = (int32_t)(blocks + 1);
                }
            }
        }
    }
    {
        const int rc_t = build_tile_layout(h, t);
        if (rc_t != ADH_OK) return rc_t;
    }
    h->h_cycle.assign(d->cycle, d->cycle + (size_t)rows * 2);
    h->h_rt_im.assign(d->rt_values, d->rt_values + d->n_frames);
    h->h_mobility_im.assign(d->mobility_values, d->mobility_values + d->scan_max_index);
    h->h_dpc = dpc;
    h->tims = t;
    h->tims_staged = true;
    note_staged(h, (uint64_t)d->n_events * 24);
    h->im_scratch_budget = 0;
    return ADH_OK;
}

int adh_stage_fragments(adh_handle_t *h, const adh_fragments_t *f) {
    if (!h || !f) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (f->n < 0) return fail(ADH_ERR_INVALID_ARGUMENT, "negative fragment count");
    if (f->n >= (int64_t)0xFFFFFFFFll) return fail(ADH_ERR_UNSUPPORTED, "too many fragments");
    HIP_TRY(hipSetDevice(h->device));
    {
        const int rc_m = materialise_tables(h);  // the last call's tables refer to the library that goes away
        if (rc_m != ADH_OK) return rc_m;
    }
    HIP_TRY(hipDeviceSynchronize());
    h->lib_buf.release();
    h->lib_staged = false;
    h->plan = Plan();
    h->cands_uploaded = false;  // fragment slices of the resident table refer to the previous library
    std::vector<LibRec> recs((size_t)f->n);
    for (int64_t i = 0; i < f->n; ++i) {
        LibRec &r = recs[(size_t)i];
        memset(&r, 0, sizeof(r));
        r.mz_library = f->mz_library[i];
        r.mz = f->mz[i];
        r.intensity = f->intensity[i];
        r.type = f->type[i];
        r.loss_type = f->loss_type[i];
        r.charge = f->charge[i];
        r.number = f->number[i];
        r.position = f->position[i];
        r.cardinality = f->cardinality[i];
    }
    UP(h->lib_buf, recs.data(), f->n, &h->d_lib);
    h->h_lib.swap(recs);
    h->n_lib = f->n;
    h->lib_staged = true;
    note_staged(h, (uint64_t)f->n * 32);
    return ADH_OK;
}

#include "adh_score_host.hip"
#include "adh_comm.hip"

namespace {

// Candidate selection on the staged ion-mobility run (see adh_select_im.hip); the caller has
// validated the arguments and zero-filled the host table.
int select_candidates_im(adh_handle *h, const adh_precursors_t *pc, const adh_selection_config_t *cfg,
                         const float *kernel, int32_t k0, int32_t k1, adh_candidate_table_t *out) {
    const int64_t n = pc->n;
    const DevTims &T = h->tims;
    const int L = T.cycle_len, SM = T.scan_max, z = T.zeroth;
    if (!pc->mobility) return fail(ADH_ERR_INVALID_ARGUMENT, "precursor mobility column is NULL");
    auto now = [] {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    const bool timing = getenv("ADH_DEBUG_TIMING") != nullptr;
    const double t_0 = now();
    // rank-1 factors of the kernel (it is an outer product up to float32 rounding of its entries)
    std::vector<double> ku((size_t)k0), kv((size_t)k1);
    {
        const int a0 = k0 / 2, b0 = k1 / 2;
        const double c = (double)kernel[a0 * k1 + b0];
        if (!(c > 0)) return fail(ADH_ERR_UNSUPPORTED, "smoothing kernel without a positive centre");
        for (int a = 0; a < k0; ++a) ku[(size_t)a] = (double)kernel[a * k1 + b0] / c;
        for (int b = 0; b < k1; ++b) kv[(size_t)b] = (double)kernel[a0 * k1 + b];
        for (int a = 0; a < k0; ++a)
            for (int b = 0; b < k1; ++b)
                if (std::fabs(ku[(size_t)a] * kv[(size_t)b] - (double)kernel[a * k1 + b]) > 1e-5 * c + 1e-30)
                    return fail(ADH_ERR_UNSUPPORTED, "smoothing kernel is not separable (not an outer product)");
    }
    const int n_iso = (int)std::min<int64_t>(cfg->top_k_precursors, pc->n_isotope_cols);
    // One grow-only slab of the handle: precursor columns, plan records, their scratch sizes, the candidate table.
    // The plan (limits, validity, empty-query test) is a kernel (adh_select_plan_im_kernel); the host only cuts
    // the batches.
    void *host_out[] = {out->precursor_idx, out->rank, out->score, out->scan_center, out->scan_start,
                        out->scan_stop, out->frame_center, out->frame_start, out->frame_stop};
    const size_t width[] = {4, 1, 4, 4, 4, 4, 4, 4, 4};
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) / 256 * 256;
        return at;
    };
    const size_t o_idx = carve((size_t)n * 4), o_fs = carve((size_t)n * 4), o_fe = carve((size_t)n * 4), o_ch = carve((size_t)n),
                 o_rt = carve((size_t)n * 4), o_mob = carve((size_t)n * 4), o_mz = carve((size_t)n * 4),
                 o_recs = carve((size_t)n * sizeof(selim::PrecRec)), o_need = carve((size_t)n * 8), o_cum = carve((size_t)n * 8),
                 o_red = carve(64), o_ku = carve((size_t)k0 * 8), o_kv = carve((size_t)k1 * 8), o_first = carve((size_t)(n + 2) * 8);
    size_t scan_bytes = 0;
    HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int)n,
                                             h->stream));
    const size_t o_tmp = carve(scan_bytes);
    size_t o_out[9];
    for (int f = 0; f < 9; ++f) o_out[f] = carve((size_t)out->n * width[f]);
    if (h->sel_slab_bytes < off) {
        HIP_TRY(hipDeviceSynchronize());
        if (h->sel_slab) (void)hipFree(h->sel_slab);
        h->sel_slab = nullptr;
        h->sel_slab_bytes = 0;
        const hipError_t e = hipMalloc(&h->sel_slab, off);
        if (e != hipSuccess) return fail(ADH_ERR_OUT_OF_MEMORY, std::string("hipMalloc(selection slab): ") + hipGetErrorString(e));
        h->sel_slab_bytes = off;
    }
    unsigned char *slab = static_cast<unsigned char *>(h->sel_slab);
    auto put = [&](size_t at, const void *src, size_t bytes) -> hipError_t {
        return bytes ? hipMemcpyAsync(slab + at, src, bytes, hipMemcpyHostToDevice, h->stream) : hipSuccess;
    };
    hipError_t e = put(o_idx, pc->precursor_idx, (size_t)n * 4);
    if (e == hipSuccess) e = put(o_fs, pc->frag_start_idx, (size_t)n * 4);
    if (e == hipSuccess) e = put(o_fe, pc->frag_stop_idx, (size_t)n * 4);
    if (e == hipSuccess) e = put(o_ch, pc->charge, (size_t)n);
    if (e == hipSuccess) e = put(o_rt, pc->rt, (size_t)n * 4);
    if (e == hipSuccess) e = put(o_mob, pc->mobility, (size_t)n * 4);
    if (e == hipSuccess) e = put(o_mz, pc->mz, (size_t)n * 4);
    if (e == hipSuccess) e = put(o_ku, ku.data(), (size_t)k0 * 8);
    if (e == hipSuccess) e = put(o_kv, kv.data(), (size_t)k1 * 8);
    if (e == hipSuccess) e = hipMemsetAsync(slab + o_red, 0, 64, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(slab + o_out[0], 0, off - o_out[0], h->stream);  // (rows without a candidate stay zero)
    if (e != hipSuccess) return fail(ADH_ERR_HIP, std::string("selection upload: ") + hipGetErrorString(e));
    SelPlanIn pin{};
    pin.precursor_idx = reinterpret_cast<const uint32_t *>(slab + o_idx);
    pin.frag_start = reinterpret_cast<const uint32_t *>(slab + o_fs);
    pin.frag_stop = reinterpret_cast<const uint32_t *>(slab + o_fe);
    pin.charge = slab + o_ch;
    pin.rt = reinterpret_cast<const float *>(slab + o_rt);
    pin.mobility = reinterpret_cast<const float *>(slab + o_mob);
    pin.mz = reinterpret_cast<const float *>(slab + o_mz);
    selim::PrecRec *d_recs = reinterpret_cast<selim::PrecRec *>(slab + o_recs);
    unsigned long long *d_need = reinterpret_cast<unsigned long long *>(slab + o_need);
    unsigned long long *d_cum = reinterpret_cast<unsigned long long *>(slab + o_cum);
    int32_t *d_red = reinterpret_cast<int32_t *>(slab + o_red);
    unsigned long long *d_biggest = reinterpret_cast<unsigned long long *>(slab + o_red + 32);
    hipLaunchKernelGGL(adh_select_plan_im_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, T, pin, n, h->n_lib,
                       n_iso, cfg->rt_tolerance, cfg->mobility_tolerance, cfg->kernel_size, (int)k0, (int)k1, d_recs, d_need, d_red,
                       d_biggest);
    e = hipGetLastError();
    size_t tb = scan_bytes;
    if (e == hipSuccess) e = hipcub::DeviceScan::InclusiveSum(slab + o_tmp, tb, d_need, d_cum, (int)n, h->stream);
    struct {
        int32_t red[8];
        unsigned long long biggest, pad;
    } meta;
    unsigned long long all = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&meta, slab + o_red, 48, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&all, d_cum + (n - 1), 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(ADH_ERR_HIP, std::string("selection plan: ") + hipGetErrorString(e));
    if (meta.red[0] & 1) return fail(ADH_ERR_INVALID_ARGUMENT, "fragment slice outside the staged library");
    if (meta.red[0] & 2) return fail(ADH_ERR_INVALID_ARGUMENT, "precursor charge must be > 0");
    if (meta.red[0] & 4) return fail(ADH_ERR_UNSUPPORTED, "more than 64 m/z windows per precursor");
    int32_t cap_cells = std::max(meta.red[1], 1), cap_s = std::max(meta.red[2], 1), cap_f = std::max(meta.red[3], 1);
    cap_cells = (cap_cells + 1) & ~1;  // the float64 kernel factors follow the float tiles in LDS
    const int tap_budget = adh_select_tap_budget(cap_cells, cap_s, k0, k1);
    const size_t lds_smooth = adh_select_smooth_im_lds_bytes(cap_cells, cap_s, k0, k1, tap_budget);
    const size_t lds = std::max(lds_smooth, adh_select_score_im_lds_bytes(cap_cells, cap_s, cap_f));
    if (lds > 150 * 1024 || cap_f > selim::SCORE_THREADS) {
        char buf[200];
        snprintf(buf, sizeof(buf), "selection tile of %d cells (scans x cycles, %d cycles) needs %zu bytes of LDS: exceeds 150 KiB",
                 cap_cells, cap_f, lds);
        return fail(ADH_ERR_UNSUPPORTED, buf);
    }
    // batches of precursors whose tiles fit a bounded scratch slab: the scratch slab of the handle is used (kept
    // between calls; only reserved memory: a precursor's tiles are normally kept in sparse form and touch a few KB of
    // their block); batches follow each other on the stream, so a batch may reuse the slab of the one before without
    // the host waiting
    uint64_t budget = 8ull << 30;
    const char *budget_env = getenv("ADH_SELECT_SCRATCH_MB");
    if (budget_env) budget = (uint64_t)atoll(budget_env) << 20;
    budget = std::max<uint64_t>(std::min<uint64_t>(budget, all), meta.biggest);
    // (a bigger slab is there already: fewer batches - unless the budget was set by hand, which is how the tests
    // reach the cutting of batches)
    if (!budget_env) budget = std::max<uint64_t>(budget, h->scratch_slab_bytes);
    const double t_plan = now();
    int rc = ADH_OK;
    std::vector<int64_t> first{0};  // first precursor of every batch, then n
    if (all > budget) {
        // the longest prefixes that fit: the inclusive sums come back and are cut on the host
        std::vector<unsigned long long> cum((size_t)n);
        HIP_TRY(hipMemcpy(cum.data(), d_cum, (size_t)n * 8, hipMemcpyDeviceToHost));
        while (first.back() < n) {
            const int64_t f0 = first.back();
            const unsigned long long base = f0 > 0 ? cum[(size_t)f0 - 1] : 0ull;
            const int64_t last = std::upper_bound(cum.begin() + f0, cum.end(), base + budget) - cum.begin();
            if (last == f0) return fail(ADH_ERR_UNSUPPORTED, "one precursor's tiles exceed the selection scratch budget");
            first.push_back(last);
        }
    } else {
        first.push_back(n);
    }
    const int n_batches = (int)first.size() - 1;
    int64_t *d_first = reinterpret_cast<int64_t *>(slab + o_first);
    HIP_TRY(hipMemcpyAsync(d_first, first.data(), first.size() * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(adh_select_offsets_im_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, d_recs, d_cum, d_need,
                       n, d_first, n_batches);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));  // (first lives on this stack frame)
    const double *d_ku = reinterpret_cast<const double *>(slab + o_ku), *d_kv = reinterpret_cast<const double *>(slab + o_kv);
    DevCandTable dt{};
    dt.precursor_idx = reinterpret_cast<uint32_t *>(slab + o_out[0]);
    dt.rank = slab + o_out[1];
    dt.score = reinterpret_cast<float *>(slab + o_out[2]);
    dt.scan_center = reinterpret_cast<uint32_t *>(slab + o_out[3]);
    dt.scan_start = reinterpret_cast<uint32_t *>(slab + o_out[4]);
    dt.scan_stop = reinterpret_cast<uint32_t *>(slab + o_out[5]);
    dt.frame_center = reinterpret_cast<uint32_t *>(slab + o_out[6]);
    dt.frame_start = reinterpret_cast<uint32_t *>(slab + o_out[7]);
    dt.frame_stop = reinterpret_cast<uint32_t *>(slab + o_out[8]);
    unsigned char *d_scratch = nullptr;
    if (h->scratch_slab_bytes < budget) {
        // (exactly the budget: ensure_scratch's head-room is for scoring batches that grow from call to call)
        e = hipDeviceSynchronize();
        if (h->scratch_slab) (void)hipFree(h->scratch_slab);
        h->scratch_slab = nullptr;
        h->scratch_slab_bytes = 0;
        if (e == hipSuccess) e = hipMalloc(&h->scratch_slab, budget);
        if (e != hipSuccess) return fail(ADH_ERR_OUT_OF_MEMORY, std::string("hipMalloc(selection scratch): ") + hipGetErrorString(e));
        h->scratch_slab_bytes = budget;
    }
    d_scratch = static_cast<unsigned char *>(h->scratch_slab);
    double total_ms = 0.0;
    const double t_alloc = now();
    {
        (void)hipFuncSetAttribute((const void *)adh_select_score_im_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024);
        (void)hipFuncSetAttribute((const void *)adh_select_smooth_im_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024);
        (void)hipFuncSetAttribute((const void *)adh_select_smooth_im_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024);
        (void)hipGetLastError();
        hipEvent_t e0 = nullptr, e1 = nullptr;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        int32_t debug_dense = 0;
        if (const char *dbg = getenv("ADH_DEBUG_SELECT_IM_DENSE")) debug_dense = atoi(dbg);
        int32_t debug_abl = 0;
        if (const char *dbg = getenv("ADH_DEBUG_SELECT_IM_ABL")) debug_abl = atoi(dbg);
        e = hipEventRecord(e0, h->stream);
        for (int b = 0; b < n_batches && e == hipSuccess; ++b) {
            const int64_t b0 = first[(size_t)b];
            const int32_t cnt = (int32_t)(first[(size_t)b + 1] - b0);
            hipLaunchKernelGGL(adh_select_gather_im_kernel, dim3((unsigned)cnt), dim3(ADH_WAVE), 0, h->stream, T,
                               h->d_lib, d_recs + b0, cnt, *cfg, (int32_t)n_iso, d_scratch, debug_dense);
            if (debug_abl != 0)
                hipLaunchKernelGGL(adh_select_smooth_im_kernel<true>, dim3((unsigned)cnt), dim3(selim::SMOOTH_THREADS), lds_smooth,
                                   h->stream, d_recs + b0, cnt, d_ku, d_kv, k0, k1, cap_cells, cap_s, d_scratch, debug_abl,
                                   (int32_t)tap_budget);
            else
                hipLaunchKernelGGL(adh_select_smooth_im_kernel<false>, dim3((unsigned)cnt), dim3(selim::SMOOTH_THREADS), lds_smooth,
                                   h->stream, d_recs + b0, cnt, d_ku, d_kv, k0, k1, cap_cells, cap_s, d_scratch, debug_abl,
                                   (int32_t)tap_budget);
            hipLaunchKernelGGL(adh_select_score_im_kernel, dim3((unsigned)cnt), dim3(selim::SCORE_THREADS),
                               adh_select_score_im_lds_bytes(cap_cells, cap_s, cap_f), h->stream, T, d_recs + b0, cnt, b0, *cfg,
                               cap_cells, cap_s, cap_f, d_scratch, dt);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
        for (int f = 0; f < 9 && e == hipSuccess; ++f)
            e = hipMemcpyAsync(host_out[f], slab + o_out[f], (size_t)out->n * width[f], hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(ADH_ERR_HIP, std::string("ion-mobility selection kernels: ") + hipGetErrorString(e));
        float ms = 0.0f;
        if (rc == ADH_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) total_ms += ms;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    const double t_kernels = now();
    if (rc == ADH_OK) h->last_select_ms = total_ms;
    const double t_copy = now();
    if (timing)
        fprintf(stderr, "[adh] select_candidates_im n=%lld: upload + plan %.2f ms, batches %.2f, kernels + copy-out %.2f (kernels %.2f)\n",
                (long long)n, t_plan - t_0, t_alloc - t_plan, t_kernels - t_alloc, total_ms);
    (void)t_copy;
    return rc;
}

}  // namespace

int adh_select_candidates(adh_handle_t *h, const adh_precursors_t *pc, const adh_selection_config_t *cfg,
                          const float *kernel, int32_t k_rows, int32_t k_cols, adh_candidate_table_t *out) {
    if (!h || !pc || !cfg || !kernel || !out) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!h->run_staged && !h->tims_staged) return fail(ADH_ERR_NOT_STAGED, "no run staged");
    if (!h->d_lib) return fail(ADH_ERR_NOT_STAGED, "no fragment library staged");
    if (pc->n < 0 || cfg->candidate_count <= 0 || cfg->candidate_count > sel::MAX_CAND)
        return fail(ADH_ERR_INVALID_ARGUMENT, "candidate_count must be in 1..16");
    if (out->n != pc->n * cfg->candidate_count)
        return fail(ADH_ERR_INVALID_ARGUMENT, "candidate table rows != precursors x candidate_count");
    if (k_rows <= 0 || k_cols <= 0 || cfg->top_k_precursors <= 0 || pc->n_isotope_cols <= 0)
        return fail(ADH_ERR_INVALID_ARGUMENT, "kernel / isotope dimensions must be positive");
    HIP_TRY(hipSetDevice(h->device));
    const bool tdbg = getenv("ADH_DEBUG_SELECT_TIMING") != nullptr;
    auto t_now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = t_now();
    auto lap = [&](const char *what) {
        if (!tdbg) return;
        const auto t = t_now();
        fprintf(stderr, "[select] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    const int64_t n = pc->n;
    void *host_out[] = {out->precursor_idx, out->rank, out->score, out->scan_center, out->scan_start,
                        out->scan_stop, out->frame_center, out->frame_start, out->frame_stop};
    const size_t width[] = {4, 1, 4, 4, 4, 4, 4, 4, 4};
    for (int f = 0; f < 9; ++f) {
        if (!host_out[f]) return fail(ADH_ERR_INVALID_ARGUMENT, "candidate table buffer is NULL");
        memset(host_out[f], 0, (size_t)out->n * width[f]);
    }
    lap("memset");
    if (n == 0) return ADH_OK;
    if (h->tims_staged) return select_candidates_im(h, pc, cfg, kernel, k_rows, k_cols, out);
    // One grow-only slab of the handle holds the precursor columns, the tile limits and the candidate table of a
    // call (19 allocations and as many frees cost more than the kernel).  The per-precursor limits - two searches
    // over the retention times of 3e5 spectra each: 17 ms per 100 000 precursors on one host core - and the checks
    // of the fragment slices are a kernel of their own; three words come back to size the LDS.
    sel::SelCaps caps{};
    caps.n_iso = (int32_t)std::min<int64_t>(cfg->top_k_precursors, pc->n_isotope_cols);
    caps.k_rows = k_rows;
    caps.k_cols = k_cols;
    const int L = h->run.cycle_len;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) / 256 * 256;
        return at;
    };
    const size_t o_idx = carve((size_t)n * 4), o_fs = carve((size_t)n * 4), o_fe = carve((size_t)n * 4), o_ch = carve((size_t)n),
                 o_rt = carve((size_t)n * 4), o_mz = carve((size_t)n * 4), o_iso = carve((size_t)n * pc->n_isotope_cols * 4),
                 o_cs = carve((size_t)n * 4), o_cc = carve((size_t)n * 4), o_red = carve(16),
                 o_kern = carve((size_t)k_rows * k_cols * 4);
    size_t o_out[9];
    for (int f = 0; f < 9; ++f) o_out[f] = carve((size_t)out->n * width[f]);
    if (h->sel_slab_bytes < off) {
        HIP_TRY(hipDeviceSynchronize());
        if (h->sel_slab) (void)hipFree(h->sel_slab);
        h->sel_slab = nullptr;
        h->sel_slab_bytes = 0;
        const hipError_t e = hipMalloc(&h->sel_slab, off);
        if (e != hipSuccess) return fail(ADH_ERR_OUT_OF_MEMORY, std::string("hipMalloc(selection slab): ") + hipGetErrorString(e));
        h->sel_slab_bytes = off;
    }
    unsigned char *slab = static_cast<unsigned char *>(h->sel_slab);
    auto put = [&](size_t at, const void *src, size_t bytes) -> hipError_t {
        return bytes ? hipMemcpyAsync(slab + at, src, bytes, hipMemcpyHostToDevice, h->stream) : hipSuccess;
    };
    hipError_t e = put(o_idx, pc->precursor_idx, (size_t)n * 4);
    if (e == hipSuccess) e = put(o_fs, pc->frag_start_idx, (size_t)n * 4);
    if (e == hipSuccess) e = put(o_fe, pc->frag_stop_idx, (size_t)n * 4);
    if (e == hipSuccess) e = put(o_ch, pc->charge, (size_t)n);
    if (e == hipSuccess) e = put(o_rt, pc->rt, (size_t)n * 4);
    if (e == hipSuccess) e = put(o_mz, pc->mz, (size_t)n * 4);
    if (e == hipSuccess) e = put(o_iso, pc->isotope_intensity, (size_t)n * pc->n_isotope_cols * 4);
    if (e == hipSuccess) e = put(o_kern, kernel, (size_t)k_rows * k_cols * 4);
    if (e == hipSuccess) e = hipMemsetAsync(slab + o_red, 0, 16, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(slab + o_out[0], 0, off - o_out[0], h->stream);  // (the candidate table: rows without a candidate stay zero)
    if (e != hipSuccess) return fail(ADH_ERR_HIP, std::string("selection upload: ") + hipGetErrorString(e));
    DevPrecursors dp{};
    dp.n_iso_cols = pc->n_isotope_cols;
    dp.precursor_idx = reinterpret_cast<const uint32_t *>(slab + o_idx);
    dp.frag_start = reinterpret_cast<const uint32_t *>(slab + o_fs);
    dp.frag_stop = reinterpret_cast<const uint32_t *>(slab + o_fe);
    dp.charge = slab + o_ch;
    dp.rt = reinterpret_cast<const float *>(slab + o_rt);
    dp.mz = reinterpret_cast<const float *>(slab + o_mz);
    dp.iso = reinterpret_cast<const float *>(slab + o_iso);
    dp.cycle_start = reinterpret_cast<const int32_t *>(slab + o_cs);
    dp.cycle_count = reinterpret_cast<const int32_t *>(slab + o_cc);
    int32_t *d_red = reinterpret_cast<int32_t *>(slab + o_red);
    hipLaunchKernelGGL(adh_select_limits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->run.rt,
                       h->run.n_spectra, L, dp, n, h->n_lib, cfg->rt_tolerance, cfg->kernel_size,
                       reinterpret_cast<int32_t *>(slab + o_cs), reinterpret_cast<int32_t *>(slab + o_cc), d_red);
    e = hipGetLastError();
    int32_t red[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(red, d_red, sizeof(red), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(ADH_ERR_HIP, std::string("selection limits: ") + hipGetErrorString(e));
    lap("uploads + limits");
    if (red[2] & 1) return fail(ADH_ERR_INVALID_ARGUMENT, "fragment slice outside the staged library");
    if (red[2] & 2) return fail(ADH_ERR_INVALID_ARGUMENT, "precursor charge must be > 0");
    caps.n_lib = std::max(red[0], 1);
    caps.f = std::max(red[1], 1);
    const size_t lds = sel::lds_bytes(caps);
    if (lds > 150 * 1024) {
        char buf[200];
        snprintf(buf, sizeof(buf), "selection tile needs %zu bytes of LDS (%d cycles, %d fragments): exceeds 150 KiB",
                 lds, caps.f, caps.n_lib);
        return fail(ADH_ERR_UNSUPPORTED, buf);
    }
    DevCandTable dt{};
    dt.precursor_idx = reinterpret_cast<uint32_t *>(slab + o_out[0]);
    dt.rank = slab + o_out[1];
    dt.score = reinterpret_cast<float *>(slab + o_out[2]);
    dt.scan_center = reinterpret_cast<uint32_t *>(slab + o_out[3]);
    dt.scan_start = reinterpret_cast<uint32_t *>(slab + o_out[4]);
    dt.scan_stop = reinterpret_cast<uint32_t *>(slab + o_out[5]);
    dt.frame_center = reinterpret_cast<uint32_t *>(slab + o_out[6]);
    dt.frame_start = reinterpret_cast<uint32_t *>(slab + o_out[7]);
    dt.frame_stop = reinterpret_cast<uint32_t *>(slab + o_out[8]);
    int rc = ADH_OK;
    {
        (void)hipFuncSetAttribute((const void *)adh_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024);
        (void)hipGetLastError();
        hipEvent_t e0 = nullptr, e1 = nullptr;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, h->stream);
        sel::Taps taps{};  // (the non-zero columns of a two-row smoothing kernel as float64 scalar operands)
        if (k_rows == 2) {
            int lo = k_cols, hi = -1;
            for (int b = 0; b < k_cols; ++b)
                if (kernel[b] != 0.0f || kernel[k_cols + b] != 0.0f) lo = std::min(lo, b), hi = std::max(hi, b);
            if (hi < lo) lo = hi = 0;
            const int nb = hi - lo + 1;
            taps.cols = nb <= sel::TAP_COLS ? sel::TAP_COLS : 0;
            taps.col0 = lo;
            for (int a = 0; a < 2 && taps.cols; ++a)
                for (int b = 0; b < nb; ++b) taps.v[a * taps.cols + b] = (double)kernel[a * k_cols + lo + b];
        }
        if (const char *env = getenv("ADH_DEBUG_SELECT_STOP")) caps.stop = atoi(env);
        if (getenv("ADH_DEBUG_SELECT_LDS_TAPS")) taps.cols = 0;  // A/B: the generic smoothing loop
        hipLaunchKernelGGL(adh_select_kernel, dim3((unsigned)n), dim3(ADH_WAVE), lds, h->stream,
                           (SelectArgs{h->run, h->d_lib, dp, (int64_t)n, *cfg, reinterpret_cast<const float *>(slab + o_kern), taps, caps, dt}));
        e = hipGetLastError();
        (void)hipEventRecord(e1, h->stream);
        for (int f = 0; f < 9 && e == hipSuccess; ++f)
            e = hipMemcpyAsync(host_out[f], slab + o_out[f], (size_t)out->n * width[f], hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(ADH_ERR_HIP, std::string("selection kernel: ") + hipGetErrorString(e));
        float ms = 0.0f;
        if (rc == ADH_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) h->last_select_ms = ms;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    lap("kernel + copy out");
    return rc;
}

namespace {

// adh_transpose_timstof for runs of more events than one sort takes (see adh_transpose.hip): two passes over
// slabs of whole pushes; the outputs are assembled in HBM (6 bytes per event) and copied out at the end.
int transpose_in_slabs(adh_handle_t *h, const uint32_t *tof_indices, const int64_t *push_indptr, int64_t n_push, int64_t n_tof,
                       const uint16_t *values, int64_t n, uint32_t *push_out, int64_t *tof_indptr_out, uint16_t *values_out,
                       int64_t slab_events) {
    // slabs of whole pushes, each within the event budget
    std::vector<int64_t> cut{0};
    while (cut.back() < n_push) {
        const int64_t p0 = cut.back(), e0 = push_indptr[p0];
        // last push whose end is still within e0 + slab_events
        int64_t p1 = std::upper_bound(push_indptr + p0, push_indptr + n_push + 1, e0 + slab_events) - push_indptr - 1;
        if (p1 <= p0) return fail(ADH_ERR_UNSUPPORTED, "a single push holds more events than one slab takes");
        cut.push_back(std::min(p1, n_push));
    }
    const int64_t n_slabs = (int64_t)cut.size() - 1;
    int64_t max_slab = 0;
    for (int64_t s = 0; s < n_slabs; ++s) max_slab = std::max(max_slab, push_indptr[cut[(size_t)s + 1]] - push_indptr[cut[(size_t)s]]);
    hipStream_t st = h->stream;
    DeviceBuffers tmp;
    int rc = ADH_OK;
    auto dev_alloc = [&](void **p, size_t bytes) -> int {
        hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(ADH_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
        }
        tmp.ptrs.push_back(*p);
        return ADH_OK;
    };
    uint32_t *d_tof = nullptr, *d_push_of = nullptr, *d_ev_in = nullptr, *d_ev_out = nullptr, *d_tof_out = nullptr, *d_push_out = nullptr;
    uint16_t *d_val = nullptr, *d_val_out = nullptr;
    int64_t *d_ptr = nullptr, *d_indptr = nullptr, *d_slab_indptr = nullptr;
    unsigned long long *d_count = nullptr, *d_running = nullptr, *d_prior = nullptr;
    int *d_bad = nullptr;
    void *sort_tmp = nullptr, *scan_tmp = nullptr;
    const int64_t *d_ptr_c = nullptr;
    rc = upload(tmp, push_indptr, n_push + 1, &d_ptr_c, st);
    d_ptr = const_cast<int64_t *>(d_ptr_c);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_tof, (size_t)max_slab * 4);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_val, (size_t)max_slab * 2);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_push_of, (size_t)max_slab * 4);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_ev_in, (size_t)max_slab * 4);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_ev_out, (size_t)max_slab * 4);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_tof_out, (size_t)max_slab * 4);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_push_out, (size_t)n * 4);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_val_out, (size_t)n * 2);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_indptr, (size_t)(n_tof + 1) * 8);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_slab_indptr, (size_t)(n_tof + 1) * 8);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_count, (size_t)(n_tof + 1) * 8);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_running, (size_t)(n_tof + 1) * 8);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_prior, (size_t)n_slabs * (size_t)(n_tof + 1) * 8);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_bad, sizeof(int));
    if (rc != ADH_OK) {
        tmp.release();
        return rc;
    }
    hipError_t e = hipMemsetAsync(d_running, 0, (size_t)(n_tof + 1) * 8, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(int), st);
    const unsigned tof_blocks = (unsigned)((n_tof + 255) / 256);
    // ---- pass 1: events per TOF bin of every slab -> where the slab's run of a bin starts inside the bin
    for (int64_t s = 0; s < n_slabs && e == hipSuccess; ++s) {
        const int64_t e0 = push_indptr[cut[(size_t)s]], ns = push_indptr[cut[(size_t)s + 1]] - e0;
        e = hipMemcpyAsync(d_tof, tof_indices + e0, (size_t)ns * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemsetAsync(d_count, 0, (size_t)(n_tof + 1) * 8, st);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(adh_tof_count_kernel, dim3(8192), dim3(256), 0, st, d_tof, ns, n_tof, d_count, d_bad);
        hipLaunchKernelGGL(adh_tof_prior_kernel, dim3(tof_blocks), dim3(256), 0, st, d_count, n_tof, d_running,
                           d_prior + (size_t)s * (size_t)(n_tof + 1));
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // (the slab buffer is reused)
    }
    int bad = 0;
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess && bad) {
        tmp.release();
        return fail(ADH_ERR_INVALID_ARGUMENT, "tof_indices holds a value >= n_tof");
    }
    if (e == hipSuccess) {  // tof_indptr = exclusive sum of the totals (entry n_tof: all events)
        size_t scan_bytes = 0;
        e = hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_running, (unsigned long long *)d_indptr, (int)(n_tof + 1), st);
        if (e == hipSuccess) rc = dev_alloc(&scan_tmp, scan_bytes);
        if (e == hipSuccess && rc == ADH_OK)
            e = hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, d_running, (unsigned long long *)d_indptr, (int)(n_tof + 1), st);
    }
    // ---- pass 2: sort every slab by TOF (stable: pushes stay ascending), scatter its runs
    int end_bit = 1;
    while (end_bit < 32 && ((int64_t)1 << end_bit) < std::max<int64_t>(n_tof, 2)) ++end_bit;
    size_t sort_bytes = 0;
    if (e == hipSuccess && rc == ADH_OK)
        e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_tof, d_tof_out, d_ev_in, d_ev_out, (int)max_slab, 0, end_bit, st);
    if (e == hipSuccess && rc == ADH_OK) rc = dev_alloc(&sort_tmp, sort_bytes);
    for (int64_t s = 0; s < n_slabs && e == hipSuccess && rc == ADH_OK; ++s) {
        const int64_t p0 = cut[(size_t)s], p1 = cut[(size_t)s + 1];
        const int64_t e0 = push_indptr[p0], ns = push_indptr[p1] - e0;
        if (ns == 0) continue;
        e = hipMemcpyAsync(d_tof, tof_indices + e0, (size_t)ns * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_val, values + e0, (size_t)ns * 2, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(adh_expand_push_slab_kernel, dim3(4096), dim3(256), 0, st, d_ptr, p0, p1, e0, d_push_of);
        hipLaunchKernelGGL(adh_iota_kernel, dim3(4096), dim3(256), 0, st, d_ev_in, ns);
        e = hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, d_tof, d_tof_out, d_ev_in, d_ev_out, (int)ns, 0, end_bit, st);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(adh_tof_indptr_kernel, dim3(1024), dim3(256), 0, st, d_tof_out, ns, n_tof, d_slab_indptr);
        hipLaunchKernelGGL(adh_transpose_scatter_kernel, dim3(8192), dim3(256), 0, st, d_tof_out, d_ev_out, d_push_of, d_val, ns,
                           d_slab_indptr, d_indptr, d_prior + (size_t)s * (size_t)(n_tof + 1), d_push_out, d_val_out);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e == hipSuccess && rc == ADH_OK) e = hipMemcpyAsync(push_out, d_push_out, (size_t)n * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && rc == ADH_OK) e = hipMemcpyAsync(values_out, d_val_out, (size_t)n * 2, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && rc == ADH_OK) e = hipMemcpyAsync(tof_indptr_out, d_indptr, (size_t)(n_tof + 1) * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && rc == ADH_OK) e = hipStreamSynchronize(st);
    if (e != hipSuccess && rc == ADH_OK) {
        (void)hipGetLastError();
        rc = fail(e == hipErrorOutOfMemory ? ADH_ERR_OUT_OF_MEMORY : ADH_ERR_HIP, std::string("transpose (slabs): ") + hipGetErrorString(e));
    }
    tmp.release();
    return rc;
}

}  // namespace

int adh_transpose_timstof(adh_handle_t *h, const uint32_t *tof_indices, const int64_t *push_indptr,
                          int64_t n_push, int64_t n_tof, const uint16_t *values, int64_t n,
                          uint32_t *push_out, int64_t *tof_indptr_out, uint16_t *values_out) {
    if (!h || !push_indptr || !tof_indptr_out || (n > 0 && (!tof_indices || !values || !push_out || !values_out)))
        return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_push < 0 || n_tof < 0 || n < 0) return fail(ADH_ERR_INVALID_ARGUMENT, "negative size");
    if (n_push >= (int64_t)0xFFFFFFFFll || n_tof >= (int64_t)0xFFFFFFFFll)
        return fail(ADH_ERR_UNSUPPORTED, "2^32 - 1 pushes / TOF bins or more are not supported (32-bit push_indices, as in the reference)");
    if (push_indptr[0] != 0 || push_indptr[n_push] != n)
        return fail(ADH_ERR_INVALID_ARGUMENT, "push_indptr does not span the event arrays");
    for (int64_t p = 0; p < n_push; ++p)
        if (push_indptr[p + 1] < push_indptr[p]) return fail(ADH_ERR_INVALID_ARGUMENT, "push_indptr is not monotone");
    HIP_TRY(hipSetDevice(h->device));
    if (n == 0) {
        for (int64_t t = 0; t <= n_tof; ++t) tof_indptr_out[t] = 0;
        return ADH_OK;
    }
    {
        // one sort handles < 2^31 events (32-bit event numbers, hipCUB item counts): longer runs go slab by slab
        int64_t slab_events = (int64_t)0x7FFFFFFFll - 1;
        if (const char *env = getenv("ADH_TRANSPOSE_SLAB_EVENTS")) slab_events = std::max<int64_t>(atoll(env), 1);  // (tests)
        if (n > slab_events)
            return transpose_in_slabs(h, tof_indices, push_indptr, n_push, n_tof, values, n, push_out, tof_indptr_out, values_out,
                                      slab_events);
    }
    hipStream_t st = h->stream;
    DeviceBuffers tmp;
    const uint32_t *d_tof = nullptr;
    const int64_t *d_ptr = nullptr;
    const uint16_t *d_val = nullptr;
    int rc = upload(tmp, tof_indices, n, &d_tof, st);
    if (rc == ADH_OK) rc = upload(tmp, push_indptr, n_push + 1, &d_ptr, st);
    if (rc == ADH_OK) rc = upload(tmp, values, n, &d_val, st);
    uint32_t *d_push_of = nullptr, *d_ev_in = nullptr, *d_ev_out = nullptr, *d_tof_out = nullptr, *d_push_out = nullptr;
    uint16_t *d_val_out = nullptr;
    int64_t *d_indptr = nullptr;
    int *d_bad = nullptr;
    void *sort_tmp = nullptr;
    auto dev_alloc = [&](void **p, size_t bytes) -> int {
        hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return fail(ADH_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
        tmp.ptrs.push_back(*p);
        return ADH_OK;
    };
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_push_of, (size_t)n * 4);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_ev_in, (size_t)n * 4);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_ev_out, (size_t)n * 4);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_tof_out, (size_t)n * 4);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_push_out, (size_t)n * 4);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_val_out, (size_t)n * 2);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_indptr, (size_t)(n_tof + 1) * 8);
    if (rc == ADH_OK) rc = dev_alloc((void **)&d_bad, sizeof(int));
    hipError_t e = hipSuccess;
    if (rc == ADH_OK) {
        e = hipMemsetAsync(d_bad, 0, sizeof(int), st);
        hipLaunchKernelGGL(adh_tof_range_kernel, dim3(4096), dim3(256), 0, st, d_tof, n, n_tof, d_bad);
        int bad = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess && bad) {
            tmp.release();
            return fail(ADH_ERR_INVALID_ARGUMENT, "tof_indices holds a value >= n_tof");
        }
    }
    if (rc == ADH_OK && e == hipSuccess) {
        hipLaunchKernelGGL(adh_expand_push_kernel, dim3(4096), dim3(256), 0, st, d_ptr, n_push, d_push_of);
        hipLaunchKernelGGL(adh_iota_kernel, dim3(4096), dim3(256), 0, st, d_ev_in, n);
        int end_bit = 1;
        while (end_bit < 32 && ((int64_t)1 << end_bit) < std::max<int64_t>(n_tof, 2)) ++end_bit;
        size_t sort_bytes = 0;
        e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_tof, d_tof_out, d_ev_in, d_ev_out, (int)n, 0,
                                               end_bit, st);
        if (e == hipSuccess) rc = dev_alloc(&sort_tmp, sort_bytes);
        if (e == hipSuccess && rc == ADH_OK)
            e = hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, d_tof, d_tof_out, d_ev_in, d_ev_out, (int)n, 0,
                                                   end_bit, st);
        if (e == hipSuccess && rc == ADH_OK) {
            hipLaunchKernelGGL(adh_transpose_gather_kernel, dim3(8192), dim3(256), 0, st, d_ev_out, d_push_of, d_val, n,
                               d_push_out, d_val_out);
            hipLaunchKernelGGL(adh_tof_indptr_kernel, dim3(1024), dim3(256), 0, st, d_tof_out, n, n_tof, d_indptr);
            e = hipGetLastError();
        }
        if (e == hipSuccess && rc == ADH_OK) e = hipMemcpyAsync(push_out, d_push_out, (size_t)n * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && rc == ADH_OK) e = hipMemcpyAsync(values_out, d_val_out, (size_t)n * 2, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && rc == ADH_OK)
            e = hipMemcpyAsync(tof_indptr_out, d_indptr, (size_t)(n_tof + 1) * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && rc == ADH_OK) e = hipStreamSynchronize(st);
        if (e != hipSuccess && rc == ADH_OK) rc = fail(ADH_ERR_HIP, std::string("transpose: ") + hipGetErrorString(e));
    }
    tmp.release();
    return rc;
}

int adh_select_time_ms(adh_handle_t *h, double *kernel_ms) {
    if (!h || !kernel_ms) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    *kernel_ms = h->last_select_ms;
    return ADH_OK;
}

int adh_fragcomp(adh_handle_t *h, int64_t n_windows, const int64_t *window_start,
                 const int64_t *window_stop, int64_t n_psm, const float *rt,
                 const int64_t *frag_start_idx, const int64_t *frag_stop_idx, int64_t n_frag,
                 const float *fragment_mz, double rt_tol_seconds, double mass_tol_ppm,
                 uint8_t *valid) {
    if (!h || (n_windows > 0 && (!window_start || !window_stop)) ||
        (n_psm > 0 && (!rt || !frag_start_idx || !frag_stop_idx || !valid)) ||
        (n_frag > 0 && !fragment_mz))
        return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_windows < 0 || n_psm < 0 || n_frag < 0)
        return fail(ADH_ERR_INVALID_ARGUMENT, "negative size");
    for (int64_t w = 0; w < n_windows; ++w)
        if (window_start[w] < 0 || window_stop[w] < window_start[w] || window_stop[w] > n_psm)
            return fail(ADH_ERR_INVALID_ARGUMENT, "window range outside the PSM table");
    {  // a PSM belongs to one window (the reference hands every window to its own thread)
        std::vector<std::pair<int64_t, int64_t>> ranges;
        for (int64_t w = 0; w < n_windows; ++w)
            if (window_stop[w] > window_start[w]) ranges.emplace_back(window_start[w], window_stop[w]);
        std::sort(ranges.begin(), ranges.end());
        for (size_t r = 1; r < ranges.size(); ++r)
            if (ranges[r].first < ranges[r - 1].second)
                return fail(ADH_ERR_INVALID_ARGUMENT, "window ranges overlap");
    }
    for (int64_t i = 0; i < n_psm; ++i)
        if (frag_start_idx[i] < 0 || frag_stop_idx[i] < frag_start_idx[i] || frag_stop_idx[i] > n_frag)
            return fail(ADH_ERR_INVALID_ARGUMENT, "fragment range outside the fragment table");
    if (n_windows == 0 || n_psm == 0) return ADH_OK;
    HIP_TRY(hipSetDevice(h->device));
    DeviceBuffers tmp;
    const int64_t *d_ws, *d_we, *d_fs, *d_fe;
    const float *d_rt, *d_mz;
    const uint8_t *d_valid_c;
    int rc;
    // (device copies first, then every column in one staged upload: 69 MB of pageable arrays for 1e6 PSMs)
    std::vector<UpJob> jobs;
    auto dev_copy = [&](auto *host, int64_t n, auto **dev) -> int {
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(host)>>;
        void *p = nullptr;
        HIP_TRY(hipMalloc(&p, (size_t)std::max<int64_t>(n, 1) * sizeof(T)));
        tmp.ptrs.push_back(p);
        *dev = static_cast<const T *>(p);
        if (n > 0) jobs.push_back(UpJob{p, host, (size_t)n * sizeof(T)});
        return ADH_OK;
    };
#define FC_UP(host, n, dev)                                   \
    rc = dev_copy(host, n, dev);                              \
    if (rc != ADH_OK) {                                       \
        tmp.release();                                        \
        return rc;                                            \
    }
    FC_UP(window_start, n_windows, &d_ws);
    FC_UP(window_stop, n_windows, &d_we);
    FC_UP(rt, n_psm, &d_rt);
    FC_UP(frag_start_idx, n_psm, &d_fs);
    FC_UP(frag_stop_idx, n_psm, &d_fe);
    FC_UP(fragment_mz, n_frag, &d_mz);
    FC_UP(valid, n_psm, &d_valid_c);
#undef FC_UP
    rc = upload_staged(h, jobs);
    if (rc != ADH_OK) {
        tmp.release();
        return rc;
    }
    uint8_t *d_valid = const_cast<uint8_t *>(d_valid_c);
    fragcomp::Stats stats;
    hipError_t e = fragcomp::compete(h->stream, n_windows, d_ws, d_we, n_psm, d_rt, d_fs, d_fe, d_mz, rt_tol_seconds,
                                     mass_tol_ppm, d_valid, &stats);
    if (e == hipSuccess) e = hipMemcpy(valid, d_valid, (size_t)n_psm, hipMemcpyDeviceToHost);
    tmp.release();
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? ADH_ERR_OUT_OF_MEMORY : ADH_ERR_HIP,
                    std::string("fragcomp: ") + hipGetErrorString(e));
    }
    h->last_fragcomp = stats;
    return ADH_OK;
}

int adh_fragcomp_stats(adh_handle_t *h, double *kernel_ms, int64_t *pairs, int64_t *waiting, int32_t *rounds,
                       int32_t *serial) {
    if (!h) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL handle");
    if (kernel_ms) *kernel_ms = h->last_fragcomp.kernel_ms;
    if (pairs) *pairs = h->last_fragcomp.pairs;
    if (waiting) *waiting = h->last_fragcomp.unknown;
    if (rounds) *rounds = h->last_fragcomp.rounds;
    if (serial) *serial = h->last_fragcomp.serial;
    return ADH_OK;
}

}  // extern "C"

#include "adh_fragcomp_plan.hip"
#include "adh_fdr.hip"
#include "adh_mlp.hip"
#include "adh_fdr_device.hip"
