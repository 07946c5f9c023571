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
    const char* (*GetErrorString)(int) = nullptr;
};
}  // namespace
static Rccl* rccl()
{
    static std::mutex mu;
    static Rccl r;
    std::lock_guard<std::mutex> lock(mu);
    if (r.lib) return &r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) return nullptr;
    r.GetUniqueId = (int (*)(void*))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(r.lib, "ncclCommInitRank");
    r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.lib, "ncclAllReduce");
    r.CommDestroy = (int (*)(void*))dlsym(r.lib, "ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) { dlclose(r.lib); r.lib = nullptr; return nullptr; }
    return &r;
}
static std::string rccl_error(Rccl* r, int code)
{
    return std::string("RCCL: ") + ((r && r->GetErrorString) ? r->GetErrorString(code) : "error") + " (" + std::to_string(code) + ")";
}

int32_t pisces_hip_comm_unique_id(uint8_t* id_out, int32_t capacity)
{
    return abi_guard<int32_t>((PiscesHip*)nullptr, [&]() -> int32_t {
    if (!id_out || capacity < PISCES_COMM_ID_BYTES) return fail(nullptr, PISCES_E_INVALID_ARG, "comm_unique_id: the id needs 128 bytes");
    Rccl* r = rccl();
    if (!r) return fail(nullptr, PISCES_E_DEVICE, "comm_unique_id: librccl could not be loaded");
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
    if (!r) return fail(h, PISCES_E_DEVICE, "comm_init: librccl could not be loaded");
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
    if (!r) return fail(h, PISCES_E_DEVICE, "reduce_summary: librccl could not be loaded");
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

