// surface_comm.inc.h — part of pisces_hip.hip (included there, inside its extern "C" block; not a translation unit of its own).
// The per-chromosome summary reduce: RCCL bound at run time with dlopen, one ncclAllReduce of int64[4].

// ---- RCCL, bound at run time: the library itself does not link librccl (a single-GPU host never loads it) ----
struct RcclId { char internal[PISCES_COMM_ID_BYTES]; };   // ncclUniqueId
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /* ncclUniqueId by value */ RcclId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string path;     // what was opened
    std::string source;   // "PISCES_HIP_RCCL_PATH" | "mapped" (a copy the process had loaded already) | "default"
};
// an RCCL that is mapped into the process already (a PyTorch host brings its own, torch/lib/librccl.so): its path, or ""
static int rccl_find_mapped(struct dl_phdr_info* info, size_t, void* out)
{
    const char* name = info->dlpi_name;
    if (!name || !*name) return 0;
    const char* base = std::strrchr(name, '/');
    base = base ? base + 1 : name;
    if (std::strncmp(base, "librccl.so", 10) != 0) return 0;
    *static_cast<std::string*>(out) = name;
    return 1;
}
}  // namespace
// Lookup order (the first that applies decides; nothing after it is tried):
//   1. PISCES_HIP_RCCL_PATH: that file, and an error naming it if it cannot be bound;
//   2. a librccl.so the process has mapped already (dl_iterate_phdr): bound with RTLD_NOLOAD, so that a host that came with its own RCCL —
//      PyTorch's wheel does — never ends up with two copies (two sets of communicator state, two IPC set-ups) in one process;
//   3. librccl.so.1 / librccl.so through the loader's search path, then /opt/rocm/lib.
static std::string g_rccl_reason;   // why the last rccl() call bound nothing (written under rccl()'s lock)
static Rccl* rccl()
{
    static std::mutex mu;
    static Rccl r;
    std::lock_guard<std::mutex> lock(mu);
    if (r.lib) return &r;
    g_rccl_reason.clear();
    const char* env = std::getenv("PISCES_HIP_RCCL_PATH");
    if (env && *env) {
        r.lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) { const char* e = dlerror(); g_rccl_reason = std::string("PISCES_HIP_RCCL_PATH=") + env + ": " + (e ? e : "cannot be loaded"); return nullptr; }
        r.path = env;
        r.source = "PISCES_HIP_RCCL_PATH";
    }
    if (!r.lib) {
        std::string mapped;
        dl_iterate_phdr(rccl_find_mapped, &mapped);
        if (!mapped.empty()) {
            r.lib = dlopen(mapped.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (r.lib) { r.path = mapped; r.source = "mapped"; }
        }
    }
    if (!r.lib)
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) { r.path = name; r.source = "default"; break; }
        }
    if (!r.lib) { g_rccl_reason = "no librccl.so.1 / librccl.so on the loader's path or under /opt/rocm/lib (PISCES_HIP_RCCL_PATH names one explicitly)"; return nullptr; }
    r.GetUniqueId = (int (*)(void*))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(r.lib, "ncclCommInitRank");
    r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.lib, "ncclAllReduce");
    r.CommDestroy = (int (*)(void*))dlsym(r.lib, "ncclCommDestroy");
    r.CommCount = (int (*)(void*, int*))dlsym(r.lib, "ncclCommCount");
    r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) {
        g_rccl_reason = r.path + " does not export the ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy this library binds";
        dlclose(r.lib);
        r.lib = nullptr;
        return nullptr;
    }
    return &r;
}
static std::string rccl_reason() { return g_rccl_reason; }
static std::string rccl_error(Rccl* r, int code)
{
    return std::string("RCCL: ") + ((r && r->GetErrorString) ? r->GetErrorString(code) : "error") + " (" + std::to_string(code) + ")";
}

int32_t pisces_hip_comm_unique_id(uint8_t* id_out, int32_t capacity)
{
    return abi_guard<int32_t>((PiscesHip*)nullptr, [&]() -> int32_t {
    if (!id_out || capacity < PISCES_COMM_ID_BYTES) return fail(nullptr, PISCES_E_INVALID_ARG, "comm_unique_id: the id needs 128 bytes");
    Rccl* r = rccl();
    if (!r) return fail(nullptr, PISCES_E_DEVICE, "comm_unique_id: librccl could not be loaded: " + rccl_reason());
    RcclId id;
    std::memset(&id, 0, sizeof(id));
    const int rc = r->GetUniqueId(&id);
    if (rc != 0) return fail(nullptr, PISCES_E_DEVICE, rccl_error(r, rc));
    std::memcpy(id_out, id.internal, PISCES_COMM_ID_BYTES);
    return PISCES_OK;
    });
}

int32_t pisces_hip_comm_init(PiscesHip* h, const uint8_t* id, int32_t rank, int32_t world)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(h, PISCES_E_INVALID_ARG, "comm_init: rank / world out of range");
    if (h->comm) return fail(h, PISCES_E_STATE, "comm_init: the handle already has a communicator");
    Rccl* r = rccl();
    if (!r) return fail(h, PISCES_E_DEVICE, "comm_init: librccl could not be loaded: " + rccl_reason());
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    PISCES_HIP_CHECK(h, h->d_summary.reserve(4));
    RcclId uid;
    std::memcpy(uid.internal, id, PISCES_COMM_ID_BYTES);
    void* comm = nullptr;
    const int rc = r->CommInitRank(&comm, world, uid, rank);
    if (rc != 0) return fail(h, PISCES_E_DEVICE, rccl_error(r, rc));
    h->comm = comm;
    h->comm_world = world;
    return PISCES_OK;
    });
}

int32_t pisces_hip_reduce_summary(PiscesHip* h, int64_t inout[4])
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !inout) return PISCES_E_INVALID_ARG;
    if (!h->comm) return PISCES_OK;   // one shard: the sum is the value
    Rccl* r = rccl();
    if (!r) return fail(h, PISCES_E_DEVICE, "reduce_summary: librccl could not be loaded: " + rccl_reason());
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    long long v[4] = {inout[0], inout[1], inout[2], inout[3]};
    PISCES_HIP_CHECK(h, hipMemcpyAsync(h->d_summary.p, v, sizeof(v), hipMemcpyHostToDevice, h->stream));
    const int rc = r->AllReduce(h->d_summary.p, h->d_summary.p, 4, /* ncclInt64 */ 4, /* ncclSum */ 0, h->comm, h->stream);
    if (rc != 0) return fail(h, PISCES_E_DEVICE, rccl_error(r, rc));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(v, h->d_summary.p, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 4; i++) inout[i] = v[i];
    return PISCES_OK;
    });
}

int32_t pisces_hip_comm_destroy(PiscesHip* h)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!h->comm) return PISCES_OK;
    Rccl* r = rccl();
    if (r) (void)r->CommDestroy(h->comm);
    h->comm = nullptr;
    h->comm_world = 1;
    return PISCES_OK;
    });
}


// Which RCCL the library bound and how it found it (binds it if that has not happened yet): "<source>: <path>", source being
// PISCES_HIP_RCCL_PATH, mapped (a copy the process had loaded already) or default.  Returns the length needed (without the NUL), or
// PISCES_E_DEVICE when none can be bound (pisces_hip_last_error(NULL) says why).
int32_t pisces_hip_comm_library(char* out, int32_t capacity)
{
    return abi_guard<int32_t>((PiscesHip*)nullptr, [&]() -> int32_t {
    Rccl* r = rccl();
    if (!r) return fail(nullptr, PISCES_E_DEVICE, "comm_library: librccl could not be loaded: " + rccl_reason());
    const std::string text = r->source + ": " + r->path;
    if (out && capacity > 0) {
        const size_t n = std::min((size_t)capacity - 1, text.size());
        std::memcpy(out, text.data(), n);
        out[n] = 0;
    }
    return (int32_t)text.size();
    });
}

// ncclCommCount of the handle's communicator (1 without one): what a launcher checks to see that every rank joined
int32_t pisces_hip_comm_ranks(PiscesHip* h, int32_t* ranks)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !ranks) return PISCES_E_INVALID_ARG;
    *ranks = 1;
    if (!h->comm) return PISCES_OK;
    Rccl* r = rccl();
    if (!r) return fail(h, PISCES_E_DEVICE, "comm_ranks: librccl could not be loaded: " + rccl_reason());
    if (!r->CommCount) { *ranks = h->comm_world; return PISCES_OK; }
    int n = 0;
    const int rc = r->CommCount(h->comm, &n);
    if (rc != 0) return fail(h, PISCES_E_DEVICE, rccl_error(r, rc));
    *ranks = n;
    return PISCES_OK;
    });
}
