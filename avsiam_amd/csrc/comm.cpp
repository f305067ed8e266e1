// Collectives of the data-parallel path as thin C-ABI wrappers over RCCL (avs_comm_init / avs_allreduce /
// avs_allgather / avs_reducescatter "on a dedicated stream with event hand-off").  They replace, for the data path, what the
// reference gets from torch.distributed: GatherLayer's all_gather + all_reduce (/root/reference/src/models/gather_layer.py:21-37)
// and DistributedDataParallel's bucketed gradient all-reduce (/root/reference/src/traintest_cavmae_base.py:58-59).
//
// One communicator = one RCCL communicator + one HIP stream of its own + two events:
//   * every collective is ordered BEHIND the work already queued on the caller's stream (`after`: an event recorded there, waited for
//     by the communicator's stream) and runs on the communicator's stream, so it overlaps whatever the caller queues next;
//   * avs_comm_wait(comm, stream) orders `stream` behind every collective issued so far.
// Nothing synchronises the host.  RCCL is resolved at run time from the copy already in the process (PyTorch loads one) or from
// the system, so libavsiam_hip.so neither links against it nor needs its headers to build: the handful of NCCL-API types and
// prototypes used here (stable since NCCL 2.x, which RCCL mirrors) are declared below; on a machine without RCCL the library still
// builds and loads, and avs_comm_* fail loudly.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

// ---- the part of the NCCL / RCCL C API this file calls (rccl.h: ncclUniqueId :43, ncclResult_t :50-, ncclRedOp_t :448, ncclDataType_t :466-468)
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclFloat32 = 7, ncclBfloat16 = 9 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclReduceScatter(const void* sendbuff, void* recvbuff, size_t recvcount, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                               hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
}

extern "C" void avs_set_error(const char* fmt, ...);

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl g_rccl;

bool load_rccl() {
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)                       // the copy the process already holds, if any (one RCCL per process)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        avs_set_error("avs_comm: cannot load librccl.so.1 (%s)", dlerror());
        return false;
    }
    Rccl r;
    r.lib = h;
#define AVS_SYM(field, sym)                                                   \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, #sym));            \
    if (!r.field) {                                                           \
        avs_set_error("avs_comm: librccl has no symbol %s", #sym);            \
        return false;                                                         \
    }
    AVS_SYM(GetUniqueId, ncclGetUniqueId)
    AVS_SYM(CommInitRank, ncclCommInitRank)
    AVS_SYM(CommDestroy, ncclCommDestroy)
    AVS_SYM(AllReduce, ncclAllReduce)
    AVS_SYM(AllGather, ncclAllGather)
    AVS_SYM(ReduceScatter, ncclReduceScatter)
    AVS_SYM(GetErrorString, ncclGetErrorString)
#undef AVS_SYM
    g_rccl = r;
    return true;
}

struct Comm {
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t issue = nullptr, done = nullptr;
    int rank = 0, world = 1, device = 0;
};

void release(Comm* c) {
    if (c->issue) (void)hipEventDestroy(c->issue);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

bool dtype_of(int dtype, ncclDataType_t* out) {
    if (dtype == 0) { *out = ncclFloat32; return true; }
    if (dtype == 1) { *out = ncclBfloat16; return true; }
    return false;
}

#define AVS_HIP(call, what)                                                              \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            avs_set_error("%s: %s failed: %s", what, #call, hipGetErrorString(e_));      \
            return -1;                                                                   \
        }                                                                                \
    } while (0)

#define AVS_NCCL(call, what)                                                             \
    do {                                                                                 \
        ncclResult_t r_ = (call);                                                        \
        if (r_ != ncclSuccess) {                                                         \
            avs_set_error("%s: RCCL error: %s", what, g_rccl.GetErrorString(r_));        \
            return -1;                                                                   \
        }                                                                                \
    } while (0)

// order the communicator's stream behind `after`, run `fn` on it, mark completion
template <class F>
int issue_on_comm_stream(Comm* c, hipStream_t after, const char* what, F fn) {
    AVS_HIP(hipEventRecord(c->issue, after), what);
    AVS_HIP(hipStreamWaitEvent(c->stream, c->issue, 0), what);
    AVS_NCCL(fn(), what);
    AVS_HIP(hipEventRecord(c->done, c->stream), what);
    return 0;
}

}  // namespace

extern "C" {

// rank 0 makes the 128-byte rendezvous id and hands it to the other ranks by any host channel (file, store, MPI ...)
int avs_comm_unique_id(void* id128) {
    if (!id128) { avs_set_error("avs_comm_unique_id: null output"); return -2; }
    if (!load_rccl()) return -1;
    ncclUniqueId id;
    AVS_NCCL(g_rccl.GetUniqueId(&id), "avs_comm_unique_id");
    static_assert(sizeof(id) == 128, "RCCL unique id is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return 0;
}

// collective over all `world` ranks (blocks the host until every rank has called it, like ncclCommInitRank); uses the CURRENT device
int avs_comm_init(const void* id128, int rank, int world, void** comm) {
    if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) {
        avs_set_error("avs_comm_init: bad arguments (rank %d of %d)", rank, world);
        return -2;
    }
    if (!load_rccl()) return -1;
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    if (hipGetDevice(&c->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->issue, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        avs_set_error("avs_comm_init: cannot create the communicator's stream / events");
        release(c);
        return -1;
    }
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
        avs_set_error("avs_comm_init: ncclCommInitRank: %s", g_rccl.GetErrorString(r));
        release(c);
        return -1;
    }
    (void)hipEventRecord(c->done, c->stream);                 // a wait before the first collective finds a completed event
    *comm = c;
    return 0;
}

int avs_comm_destroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return 0;
    (void)hipStreamSynchronize(c->stream);
    if (c->nccl) g_rccl.CommDestroy(c->nccl);
    release(c);
    return 0;
}

int avs_comm_rank(void* comm) { return comm ? static_cast<Comm*>(comm)->rank : -2; }
int avs_comm_world(void* comm) { return comm ? static_cast<Comm*>(comm)->world : -2; }

// buf[count] <- SUM over ranks, in place.  dtype 0 = fp32, 1 = bf16.  Ordered behind `after`, runs on the communicator's stream.
int avs_allreduce(void* comm, void* buf, unsigned long long count, int dtype, hipStream_t after) {
    Comm* c = static_cast<Comm*>(comm);
    ncclDataType_t dt;
    if (!c || !buf || !dtype_of(dtype, &dt)) { avs_set_error("avs_allreduce: bad arguments"); return -2; }
    return issue_on_comm_stream(c, after, "avs_allreduce", [&] { return g_rccl.AllReduce(buf, buf, count, dt, ncclSum, c->nccl, c->stream); });
}

// out[world * count] <- every rank's in[count], rank-major
int avs_allgather(void* comm, const void* in, void* out, unsigned long long count, int dtype, hipStream_t after) {
    Comm* c = static_cast<Comm*>(comm);
    ncclDataType_t dt;
    if (!c || !in || !out || !dtype_of(dtype, &dt)) { avs_set_error("avs_allgather: bad arguments"); return -2; }
    return issue_on_comm_stream(c, after, "avs_allgather", [&] { return g_rccl.AllGather(in, out, count, dt, c->nccl, c->stream); });
}

// out[count] <- SUM over ranks of in[rank * count ...]   (in holds world * count elements)
int avs_reducescatter(void* comm, const void* in, void* out, unsigned long long count, int dtype, hipStream_t after) {
    Comm* c = static_cast<Comm*>(comm);
    ncclDataType_t dt;
    if (!c || !in || !out || !dtype_of(dtype, &dt)) { avs_set_error("avs_reducescatter: bad arguments"); return -2; }
    return issue_on_comm_stream(c, after, "avs_reducescatter", [&] { return g_rccl.ReduceScatter(in, out, count, dt, ncclSum, c->nccl, c->stream); });
}

// `stream` waits (on the device) for every collective issued on this communicator so far
int avs_comm_wait(void* comm, hipStream_t stream) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) { avs_set_error("avs_comm_wait: null communicator"); return -2; }
    AVS_HIP(hipStreamWaitEvent(stream, c->done, 0), "avs_comm_wait");
    return 0;
}

}  // extern "C"
