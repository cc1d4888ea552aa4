// tma_comm.hip -- RCCL communicator behind the C ABI: the gradient all-reduce of the data-parallel epoch loop without a host-language hop.
//
// The reference trains in one process (/root/reference/backend/mlagents/training.py:71-89,150: one DummyVecEnv, one PPO); north_star shards
// the envs over the GPUs of a node with ONE collective on the path, the per-minibatch SUM of the flat f32 policy gradient (SURVEY.md 8e,
// 5.8).  Rounds 1-3 issued that collective through a ctypes -> Python -> torch.distributed callback from inside tma_ppo_train_epoch_dp; here
// the library owns an RCCL communicator (ncclCommInitRank from a unique id the caller broadcasts over whatever process group it already
// has) and calls ncclAllReduce itself, in place, on the compute stream -- the optimizer launch queues right behind it, no event hand-off to a
// second stream, no interpreter between the gradient kernel and the collective.  tma_comm_allreduce_cb has the tma_allreduce_fn signature,
// so the epoch entry point is unchanged: pass it with the communicator as ctx.
//
// RCCL is bound at RUN time (dlopen of librccl.so.1): the library still loads on a machine without RCCL or without a GPU (the `not gpu`
// tests), and inside a PyTorch process the already-loaded librccl of that process is reused (same SONAME) instead of a second copy.
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "tma_common.h"

namespace {

typedef struct ncclComm *ncclComm_t;
struct ncclUniqueId_ {
    char internal[128];
};
typedef int ncclResult_t;
constexpr int kNcclSum = 0, kNcclFloat32 = 7, kNcclFloat64 = 8;  // rccl.h: ncclRedOp_t / ncclDataType_t

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId_ *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId_, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = {0};
};

Rccl g_rccl;

Rccl *rccl() {
    static std::once_flag once;
    std::call_once(once, [] {
        Rccl &r = g_rccl;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            snprintf(r.why, sizeof(r.why), "librccl.so.1 could not be loaded: %s", dlerror());
            return;
        }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
            snprintf(r.why, sizeof(r.why), "librccl.so.1 lacks an expected nccl* symbol");
            r.handle = nullptr;
        }
    });
    return g_rccl.handle ? &g_rccl : nullptr;
}

const char *rccl_why() {
    rccl();
    return g_rccl.why[0] ? g_rccl.why : "RCCL is not available";
}

}  // namespace

struct tma_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = -1;
    hipStream_t stream = nullptr;  // tma_comm_bind_stream: the stream tma_comm_allreduce_cb enqueues on
    // timing (bench.py dp_timing): HIP events around the next `want` all-reduces issued through this communicator
    int want = 0;
    std::vector<hipEvent_t> ev;
    int64_t calls = 0;
};

#define TMA_NCCL(expr)                                                                                                       \
    do {                                                                                                                     \
        ncclResult_t _r = (expr);                                                                                            \
        if (_r != 0) return ::tma::fail(TMA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, R->GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

extern "C" {

int tma_comm_available(void) { return rccl() != nullptr ? 1 : 0; }

int tma_comm_unique_id(unsigned char *id_out128) {
    if (!id_out128) return tma::fail(TMA_ERR_INVALID, "tma_comm_unique_id: null argument");
    Rccl *R = rccl();
    if (!R) return tma::fail(TMA_ERR_HIP, "tma_comm_unique_id: %s", rccl_why());
    ncclUniqueId_ id;
    TMA_NCCL(R->GetUniqueId(&id));
    memcpy(id_out128, id.internal, 128);
    return TMA_OK;
}

int tma_comm_create(const unsigned char *id128, int world, int rank, int device, tma_comm **out) {
    if (!id128 || !out) return tma::fail(TMA_ERR_INVALID, "tma_comm_create: null argument");
    if (world < 1 || rank < 0 || rank >= world) return tma::fail(TMA_ERR_INVALID, "tma_comm_create: rank %d of world %d", rank, world);
    Rccl *R = rccl();
    if (!R) return tma::fail(TMA_ERR_HIP, "tma_comm_create: %s", rccl_why());
    if (device >= 0) TMA_HIP(hipSetDevice(device));
    ncclUniqueId_ id;
    memcpy(id.internal, id128, 128);
    tma_comm *c = new tma_comm();
    c->world = world, c->rank = rank, c->device = device;
    ncclResult_t rc = R->CommInitRank(&c->comm, world, id, rank);
    if (rc != 0) {
        delete c;
        return tma::fail(TMA_ERR_HIP, "ncclCommInitRank(world %d, rank %d) failed: %s", world, rank, R->GetErrorString(rc));
    }
    *out = c;
    return TMA_OK;
}

int tma_comm_destroy(tma_comm *c) {
    if (!c) return TMA_OK;
    Rccl *R = rccl();
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    delete c;
    return TMA_OK;
}

int tma_comm_bind_stream(tma_comm *c, void *stream) {
    if (!c) return tma::fail(TMA_ERR_INVALID, "tma_comm_bind_stream: null communicator");
    c->stream = (hipStream_t)stream;
    return TMA_OK;
}

int tma_comm_allreduce(tma_comm *c, void *buffer, int64_t count, int dtype, void *stream) {
    if (!c || !buffer || count < 1) return tma::fail(TMA_ERR_INVALID, "tma_comm_allreduce: null argument or empty buffer");
    if (dtype != 0 && dtype != 1) return tma::fail(TMA_ERR_INVALID, "tma_comm_allreduce: dtype must be 0 (f32) or 1 (f64)");
    Rccl *R = rccl();
    if (!R) return tma::fail(TMA_ERR_HIP, "tma_comm_allreduce: %s", rccl_why());
    hipStream_t s = (hipStream_t)stream;
    const bool timed = (int)(c->ev.size() / 2) < c->want;
    if (timed) {
        hipEvent_t a, b;
        TMA_HIP(hipEventCreate(&a));
        TMA_HIP(hipEventCreate(&b));
        c->ev.push_back(a), c->ev.push_back(b);
        TMA_HIP(hipEventRecord(a, s));
    }
    TMA_NCCL(R->AllReduce(buffer, buffer, (size_t)count, dtype == 0 ? kNcclFloat32 : kNcclFloat64, kNcclSum, c->comm, s));
    if (timed) TMA_HIP(hipEventRecord(c->ev.back(), s));
    c->calls++;
    return TMA_OK;
}

int tma_comm_allreduce_cb(void *ctx, float *buffer, int64_t count) {
    tma_comm *c = static_cast<tma_comm *>(ctx);
    if (!c) return 1;
    return tma_comm_allreduce(c, buffer, count, 0, c->stream) == TMA_OK ? 0 : 1;
}

int tma_comm_timing(tma_comm *c, int samples) {
    if (!c || samples < 0) return tma::fail(TMA_ERR_INVALID, "tma_comm_timing: bad argument");
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    c->ev.clear();
    c->want = samples;
    return TMA_OK;
}

int tma_comm_pop_timing(tma_comm *c, float *us_out, int capacity, int *n_out, int64_t *calls_out) {
    if (!c || !n_out || (capacity > 0 && !us_out)) return tma::fail(TMA_ERR_INVALID, "tma_comm_pop_timing: null argument");
    int n = 0;
    for (size_t i = 0; i + 1 < c->ev.size() && n < capacity; i += 2) {
        TMA_HIP(hipEventSynchronize(c->ev[i + 1]));
        float ms = 0.0f;
        TMA_HIP(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        us_out[n++] = ms * 1e3f;
    }
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    c->ev.clear();
    c->want = 0;
    *n_out = n;
    if (calls_out) *calls_out = c->calls;
    return TMA_OK;
}

}  // extern "C"
