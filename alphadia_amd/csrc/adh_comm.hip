// adh_comm.hip - the one collective of the path: an RCCL all-gather of the packed score/feature
// tables over xGMI, behind the C ABI (included by adh_api.hip inside its extern "C" block).
//
// Score groups are independent (alphadia/search/scoring/containers/score_group.py:66-75: disjoint
// output rows), so every rank scores a contiguous shard of the candidate table and ONE all-gather
// reassembles the computed tables on every GPU.  librccl is opened with dlopen() when a
// communicator is created: a single-GPU process never loads it.  One communicator per handle
// (= per GPU = per process); the collective runs on its own stream so that it overlaps the D2H
// copies of the same call and the kernels of the next one (two table slots).
struct adh_comm_state {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;           // the collective's stream
    hipEvent_t ready = nullptr;             // kernels of the call that feeds the gather
    hipEvent_t done[2] = {nullptr, nullptr};  // gather of table slot s finished
    bool pending[2] = {false, false};
    void *gathered[2] = {nullptr, nullptr};   // [world][wire_bytes] per slot
    size_t gathered_bytes[2] = {0, 0};
    size_t wire_bytes[2] = {0, 0};
    double *d_scalar = nullptr;             // all-reduce scratch
};

namespace {

// RCCL prints a version banner to stdout when the first communicator is created; callers may
// reserve stdout for their own output (bench.py: one JSON line), so it is sent to stderr
struct StdoutToStderr {
    int saved = -1;
    StdoutToStderr() {
        fflush(stdout);
        saved = dup(1);
        if (saved >= 0) (void)dup2(2, 1);
    }
    ~StdoutToStderr() {
        fflush(stdout);
        if (saved >= 0) {
            (void)dup2(saved, 1);
            close(saved);
        }
    }
};

int rccl_open(adh_comm_state &c) {
    if (c.lib) return ADH_OK;
    // The RCCL that belongs to the HIP runtime THIS library runs on: the one in the same directory, by absolute
    // path.  A bare soname is not enough: a process that also imports torch holds the wheel's private ROCm
    // copies (its own libamdhip64 / libhsa-runtime64 / librccl), dlopen("librccl.so.1") then returns the
    // wheel's RCCL, which talks to the wheel's second, device-less HSA runtime ("no ROCm-capable device is
    // detected" from ncclCommInitRank; tools/probes/rccl_with_torch_probe.py).
    std::string beside[2];
    {
        Dl_info info;
        if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                beside[0] = dir + "librccl.so.1";
                beside[1] = dir + "librccl.so";
            }
        }
    }
    const char *names[] = {getenv("ADH_RCCL_LIBRARY"), beside[0].c_str(), beside[1].c_str(), "librccl.so.1", "librccl.so",
                           "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        if (!nm || !nm[0]) continue;
        c.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (c.lib) break;
    }
    if (!c.lib) return fail(ADH_ERR_UNSUPPORTED, std::string("librccl not found: ") + (dlerror() ? dlerror() : ""));
#define ADH_RCCL_SYM(field, name)                                                         \
    c.field = reinterpret_cast<decltype(c.field)>(dlsym(c.lib, name));                    \
    if (!c.field) return fail(ADH_ERR_UNSUPPORTED, std::string("librccl lacks ") + name)
    ADH_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    ADH_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    ADH_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    ADH_RCCL_SYM(AllGather, "ncclAllGather");
    ADH_RCCL_SYM(AllReduce, "ncclAllReduce");
    ADH_RCCL_SYM(GetErrorString, "ncclGetErrorString");
    ADH_RCCL_SYM(CommCount, "ncclCommCount");
    ADH_RCCL_SYM(CommUserRank, "ncclCommUserRank");
#undef ADH_RCCL_SYM
    return ADH_OK;
}

#define RCCL_TRY(c, expr)                                                                              \
    do {                                                                                               \
        ncclResult_t _r = (expr);                                                                      \
        if (_r != ncclSuccess)                                                                         \
            return fail(ADH_ERR_HIP, std::string(#expr) + ": " + ((c).GetErrorString ? (c).GetErrorString(_r) : "?")); \
    } while (0)

// before table slot `slot` is overwritten: its previous gather must have read it
int comm_wait_slot(adh_handle *h, int slot) {
    adh_comm_state *c = h->comm;
    if (!c || !c->pending[slot]) return ADH_OK;
    HIP_TRY(hipEventSynchronize(c->done[slot]));
    c->pending[slot] = false;
    return ADH_OK;
}

// enqueue the all-gather of the wire prefix of table slot `slot` (after everything on the compute stream)
int comm_gather_slot(adh_handle *h, int slot) {
    adh_comm_state *c = h->comm;
    if (!c) return ADH_OK;
    DevTables &t = h->tables[slot];
    const size_t need = t.wire_bytes * (size_t)c->world;
    if (c->gathered_bytes[slot] < need) {
        if (c->gathered[slot]) (void)hipFree(c->gathered[slot]);
        c->gathered[slot] = nullptr;
        c->gathered_bytes[slot] = 0;
        HIP_TRY(hipMalloc(&c->gathered[slot], std::max<size_t>(need, 256)));
        c->gathered_bytes[slot] = std::max<size_t>(need, 256);
    }
    c->wire_bytes[slot] = t.wire_bytes;
    HIP_TRY(hipEventRecord(c->ready, h->stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0));
    RCCL_TRY(*c, c->AllGather(t.base, c->gathered[slot], t.wire_bytes, ncclUint8, c->comm, c->stream));
    HIP_TRY(hipEventRecord(c->done[slot], c->stream));
    c->pending[slot] = true;
    return ADH_OK;
}

}  // namespace

int adh_comm_unique_id(void *id128) {
    if (!id128) return fail(ADH_ERR_INVALID_ARGUMENT, "id buffer is NULL");
    adh_comm_state tmp;
    int rc = rccl_open(tmp);
    if (rc != ADH_OK) return rc;
    ncclUniqueId id;
    {
        StdoutToStderr quiet;
        RCCL_TRY(tmp, tmp.GetUniqueId(&id));
    }
    memcpy(id128, &id, sizeof(id));
    return ADH_OK;  // the library stays loaded (a later adh_comm_init reuses the mapping)
}

int adh_comm_init(adh_handle_t *h, int rank, int world, const void *id128, int64_t max_rows_per_rank) {
    if (!h || !id128) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world || max_rows_per_rank < 0)
        return fail(ADH_ERR_INVALID_ARGUMENT, "invalid rank / world size / row count");
    if (h->comm) return fail(ADH_ERR_INVALID_ARGUMENT, "the handle already has a communicator");
    HIP_TRY(hipSetDevice(h->device));
    adh_comm_state *c = new adh_comm_state();
    int rc = rccl_open(*c);
    if (rc != ADH_OK) {
        delete c;
        return rc;
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r;
    {
        StdoutToStderr quiet;
        r = c->CommInitRank(&c->comm, world, id, rank);
    }
    if (r != ncclSuccess) {
        std::string msg = std::string("ncclCommInitRank: ") + c->GetErrorString(r);
        delete c;
        return fail(ADH_ERR_HIP, msg);
    }
    c->rank = rank;
    c->world = world;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[1], hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_scalar, 2 * sizeof(double));
    h->comm = c;
    h->comm_rows = max_rows_per_rank;
    if (e != hipSuccess) {
        adh_comm_destroy(h);
        return fail(ADH_ERR_HIP, std::string("communicator resources: ") + hipGetErrorString(e));
    }
    return ADH_OK;
}

int adh_comm_destroy(adh_handle_t *h) {
    if (!h || !h->comm) return ADH_OK;
    adh_comm_state *c = h->comm;
    (void)hipSetDevice(h->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)c->CommDestroy(c->comm);
    for (int s = 0; s < 2; ++s) {
        if (c->gathered[s]) (void)hipFree(c->gathered[s]);
        if (c->done[s]) (void)hipEventDestroy(c->done[s]);
    }
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->d_scalar) (void)hipFree(c->d_scalar);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    h->comm = nullptr;
    h->comm_rows = 0;
    return ADH_OK;
}

int adh_comm_wait(adh_handle_t *h) {
    if (!h) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL handle");
    if (!h->comm) return ADH_OK;
    HIP_TRY(hipSetDevice(h->device));
    for (int s = 0; s < 2; ++s) {
        int rc = comm_wait_slot(h, s);
        if (rc != ADH_OK) return rc;
    }
    return ADH_OK;
}

int adh_comm_gathered(adh_handle_t *h, int rank, adh_output_t *device_view, int64_t *rows) {
    if (!h || !device_view) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    adh_comm_state *c = h->comm;
    if (!c || h->last_tables < 0) return fail(ADH_ERR_NOT_STAGED, "no gathered tables");
    if (rank < 0 || rank >= c->world) return fail(ADH_ERR_INVALID_ARGUMENT, "rank out of range");
    const int slot = h->last_tables;
    int rc = comm_wait_slot(h, slot);
    if (rc != ADH_OK) return rc;
    const DevTables &t = h->tables[slot];
    adh_output_t view;
    memset(&view, 0, sizeof(view));
    unsigned char *base = static_cast<unsigned char *>(c->gathered[slot]) + (size_t)rank * c->wire_bytes[slot];
    adh_output_t full;
    memset(&full, 0, sizeof(full));
    layout_tables(base, t.rows, t.top_k, &full, nullptr);
    for (int i = 0; i < kNumOutFields; ++i)
        if (kOutFields[i].wire) *out_member(&view, kOutFields[i]) = *out_member(&full, kOutFields[i]);
    view.n = t.rows;
    view.top_k = t.top_k;
    *device_view = view;
    if (rows) *rows = t.rows;
    return ADH_OK;
}

int adh_comm_all_reduce_max(adh_handle_t *h, double *value) {
    if (!h || !value) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    adh_comm_state *c = h->comm;
    if (!c) return ADH_OK;  // a single rank: the value is its own maximum
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(c->d_scalar, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(*c, c->AllReduce(c->d_scalar, c->d_scalar + 1, 1, ncclFloat64, ncclMax, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(value, c->d_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ADH_OK;
}

// Every rank contributes `bytes` bytes of a host buffer and receives world x bytes, rank after rank: the
// exchange of the two other stages that shard (candidate selection by precursor range, fragment competition by
// DIA window) - one gather of small per-rank results, staged through HBM so that it travels over xGMI like the
// tables do.  Without a communicator the buffer is its own gather.
int adh_comm_all_gather_host(adh_handle_t *h, const void *send, uint64_t bytes, void *recv) {
    if (!h || (bytes > 0 && (!send || !recv))) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    adh_comm_state *c = h->comm;
    if (!c) {
        if (bytes > 0 && recv != send) memmove(recv, send, (size_t)bytes);
        return ADH_OK;
    }
    if (bytes == 0) return ADH_OK;  // (every rank passes the same count: nothing to wait for)
    HIP_TRY(hipSetDevice(h->device));
    void *d_send = nullptr, *d_recv = nullptr;
    HIP_TRY(hipMalloc(&d_send, (size_t)bytes));
    hipError_t e = hipMalloc(&d_recv, (size_t)bytes * (size_t)c->world);
    if (e != hipSuccess) {
        (void)hipFree(d_send);
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? ADH_ERR_OUT_OF_MEMORY : ADH_ERR_HIP, std::string("all-gather staging: ") + hipGetErrorString(e));
    }
    int rc = ADH_OK;
    e = hipMemcpyAsync(d_send, send, (size_t)bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const ncclResult_t r = c->AllGather(d_send, d_recv, (size_t)bytes, ncclUint8, c->comm, c->stream);
        if (r != ncclSuccess) rc = fail(ADH_ERR_HIP, std::string("ncclAllGather: ") + c->GetErrorString(r));
    }
    if (e == hipSuccess && rc == ADH_OK)
        e = hipMemcpyAsync(recv, d_recv, (size_t)bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && rc == ADH_OK) e = hipStreamSynchronize(c->stream);
    else (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_send);
    (void)hipFree(d_recv);
    if (rc == ADH_OK && e != hipSuccess) {
        (void)hipGetLastError();
        rc = fail(ADH_ERR_HIP, std::string("all-gather: ") + hipGetErrorString(e));
    }
    return rc;
}

int adh_comm_barrier(adh_handle_t *h) {
    double v = 0.0;
    return adh_comm_all_reduce_max(h, &v);
}

int adh_comm_info(adh_handle_t *h, int *rank, int *world) {
    if (!h || !rank || !world) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL argument");
    *rank = 0;
    *world = 1;
    if (!h->comm) return ADH_OK;
    adh_comm_state &c = *h->comm;
    int n = 0, r = 0;
    RCCL_TRY(c, c.CommCount(c.comm, &n));
    RCCL_TRY(c, c.CommUserRank(c.comm, &r));
    *rank = r;
    *world = n;
    return ADH_OK;
}

int adh_device_synchronize(adh_handle_t *h) {
    if (!h) return fail(ADH_ERR_INVALID_ARGUMENT, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    return ADH_OK;
}
