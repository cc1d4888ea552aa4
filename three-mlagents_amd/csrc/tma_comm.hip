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

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "tma_common.h"
#include "tma_p2p.h"

#include <unistd.h>

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
    ncclComm_t comm = nullptr;  // null: a communicator without RCCL (tma_comm_create_p2p): every all-reduce goes through the peer exchange
    int world = 1, rank = 0, device = -1;
    hipStream_t stream = nullptr;  // tma_comm_bind_stream: the stream tma_comm_allreduce_cb enqueues on
    // timing (bench.py dp_timing): HIP events around the next `want` all-reduces issued through this communicator
    int want = 0;
    std::vector<hipEvent_t> ev;
    int64_t calls = 0;
    // peer exchange (tma_p2p.h): own inbox [2 parities][world][cap] 8-byte words in fine-grained device memory, the peers' inboxes as opened
    // from their IPC handles, the sequence number of the last all-reduce that went through it, the host-mapped timeout flag
    unsigned long long *inbox = nullptr;
    unsigned long long *peer[tma::P2P_MAX_WORLD] = {};
    int64_t cap = 0;
    bool attached = false, p2p_on = false;
    uint32_t seq = 0;
    int *err_host = nullptr, *err_dev = nullptr;
    long long timeout_ticks = 0;
    bool local_peers = false;  // tma_comm_p2p_attach_local: the peers' inboxes are pointers of this process, not IPC mappings
    int64_t p2p_calls = 0;
};


// ---- peer exchange: stand-alone kernels (an all-reduce that is not fused into its producer / consumer: tma_comm_allreduce) ----
namespace tma {

__global__ __launch_bounds__(256) void p2p_push_kernel(const uint32_t *__restrict__ src, int64_t n_words, PeerPush p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_words) p2p_push(p, i, src[i]);
}

__global__ __launch_bounds__(256) void p2p_pull_f32_kernel(float *__restrict__ dst, int64_t n, PeerPull q) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = p2p_pull_f32(q, i);
}

// f64 payloads travel as two words (low half at 2 i, high half at 2 i + 1)
__global__ __launch_bounds__(256) void p2p_pull_f64_kernel(double *__restrict__ dst, int64_t n, PeerPull q) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned long long lo[P2P_MAX_WORLD], hi[P2P_MAX_WORLD];
    if (!p2p_wait(q, 2 * i, lo) || !p2p_wait(q, 2 * i + 1, hi)) {  // timed out: NaN, never a sum of stale words (p2p_pull_f32)
        dst[i] = __longlong_as_double(0x7FF8000000000000LL);
        return;
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < P2P_MAX_WORLD; r++) {
        if (r < q.world) {
            const double v = __longlong_as_double((long long)(((hi[r] & 0xFFFFFFFFull) << 32) | (lo[r] & 0xFFFFFFFFull)));
            s = r == 0 ? v : s + v;
        }
    }
    dst[i] = s;
}

bool tma_comm_p2p_ready(const tma_comm *c, int64_t count_words) { return c && c->p2p_on && c->attached && count_words >= 1 && count_words <= c->cap; }

int tma_comm_p2p_next(tma_comm *c, int64_t count_words, PeerPush *push, PeerPull *pull) {
    if (!tma_comm_p2p_ready(c, count_words)) return fail(TMA_ERR_INVALID, "peer exchange: not attached / enabled, or %lld words exceed the slot", (long long)count_words);
    if (*c->err_host) return fail(TMA_ERR_HIP, "peer exchange: an earlier all-reduce timed out waiting for a peer's words (rank %d of %d)", c->rank, c->world);
    if (++c->seq == 0) c->seq = 2;  // (0 is the sequence number of a word nobody wrote yet; 0xFFFFFFFF was odd, so the wrap lands on an EVEN number: the two-parity argument holds across it)
    const int64_t parity = c->seq & 1;
    *push = PeerPush{};
    for (int d = 0; d < c->world; d++) push->dst[d] = c->peer[d] + (parity * c->world + c->rank) * c->cap;
    push->world = c->world, push->seq = c->seq;
    *pull = PeerPull{c->inbox + parity * c->world * c->cap, c->cap, c->world, c->seq, c->err_dev, c->timeout_ticks};
    c->p2p_calls++;
    c->calls++;
    return TMA_OK;
}

hipStream_t tma_comm_bound_stream(const tma_comm *c) { return c->stream; }

int tma_comm_time_begin(tma_comm *c, hipStream_t s) {
    if (!c || (int)(c->ev.size() / 2) >= c->want) return 0;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return 0;
    if (hipEventCreate(&b) != hipSuccess) {
        (void)hipEventDestroy(a);
        return 0;
    }
    c->ev.push_back(a), c->ev.push_back(b);
    (void)hipEventRecord(a, s);
    return 1;
}

void tma_comm_time_end(tma_comm *c, hipStream_t s) { (void)hipEventRecord(c->ev.back(), s); }

}  // namespace tma

namespace {

void p2p_release(tma_comm *c) {
    for (int r = 0; r < tma::P2P_MAX_WORLD; r++) {
        if (c->peer[r] && c->peer[r] != c->inbox && !c->local_peers) (void)hipIpcCloseMemHandle(c->peer[r]);
        c->peer[r] = nullptr;
    }
    if (c->inbox) (void)hipFree(c->inbox);
    if (c->err_host) (void)hipHostFree(c->err_host);
    c->inbox = nullptr, c->err_host = nullptr, c->err_dev = nullptr, c->attached = false, c->p2p_on = false, c->cap = 0;
}

int p2p_allreduce(tma_comm *c, void *buffer, int64_t count, int dtype, hipStream_t s) {
    const int64_t words = dtype == 0 ? count : 2 * count;
    tma::PeerPush push;
    tma::PeerPull pull;
    int rc = tma::tma_comm_p2p_next(c, words, &push, &pull);
    if (rc) return rc;
    tma::p2p_push_kernel<<<dim3((unsigned)tma::ceil_div(words, 256)), dim3(256), 0, s>>>(static_cast<const uint32_t *>(buffer), words, push);
    TMA_LAUNCH_CHECK();
    const int timed = tma::tma_comm_time_begin(c, s);
    if (dtype == 0) tma::p2p_pull_f32_kernel<<<dim3((unsigned)tma::ceil_div(count, 256)), dim3(256), 0, s>>>(static_cast<float *>(buffer), count, pull);
    else tma::p2p_pull_f64_kernel<<<dim3((unsigned)tma::ceil_div(count, 256)), dim3(256), 0, s>>>(static_cast<double *>(buffer), count, pull);
    TMA_LAUNCH_CHECK();
    if (timed) tma::tma_comm_time_end(c, s);
    return TMA_OK;
}

}  // namespace

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
    p2p_release(c);
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
    hipStream_t s = (hipStream_t)stream;
    if (tma::tma_comm_p2p_ready(c, dtype == 0 ? count : 2 * count)) return p2p_allreduce(c, buffer, count, dtype, s);
    if (!c->comm) return tma::fail(TMA_ERR_INVALID, "tma_comm_allreduce: this communicator has no RCCL side and %lld elements do not fit its peer exchange", (long long)count);
    Rccl *R = rccl();
    if (!R) return tma::fail(TMA_ERR_HIP, "tma_comm_allreduce: %s", rccl_why());
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

// ---- peer exchange: set-up (include/tma.h) ----
int tma_comm_create_p2p(int world, int rank, int device, tma_comm **out) {
    if (!out) return tma::fail(TMA_ERR_INVALID, "tma_comm_create_p2p: null argument");
    if (world < 1 || world > tma::P2P_MAX_WORLD || rank < 0 || rank >= world) return tma::fail(TMA_ERR_INVALID, "tma_comm_create_p2p: rank %d of world %d (at most %d ranks)", rank, world, tma::P2P_MAX_WORLD);
    if (device >= 0) TMA_HIP(hipSetDevice(device));
    tma_comm *c = new tma_comm();
    c->world = world, c->rank = rank, c->device = device;
    *out = c;
    return TMA_OK;
}

// 64-bit identity of THIS host for the ticket: the kernel's boot id (unique per boot of a machine), else the host name.  Two ranks on different
// machines can see equal PCI bus ids; an IPC handle of another host must be refused by name, not by what hipIpcOpenMemHandle makes of it.
static unsigned long long p2p_host_id() {
    char buf[256] = {0};
    FILE *f = fopen("/proc/sys/kernel/random/boot_id", "r");
    size_t n = 0;
    if (f) {
        n = fread(buf, 1, sizeof(buf) - 1, f);
        fclose(f);
    }
    if (n == 0 && gethostname(buf, sizeof(buf) - 1) != 0) snprintf(buf, sizeof(buf), "unknown-host");
    unsigned long long h = 1469598103934665603ull;  // FNV-1a
    for (const char *q = buf; *q; q++) h = (h ^ (unsigned char)*q) * 1099511628211ull;
    return h ? h : 1ull;
}

int tma_comm_p2p_prepare(tma_comm *c, int64_t max_words, unsigned char *ticket_out128) {
    if (!c || !ticket_out128 || max_words < 1) return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_prepare: bad argument");
    if (c->world > tma::P2P_MAX_WORLD) return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_prepare: the peer exchange serves at most %d ranks (world %d)", tma::P2P_MAX_WORLD, c->world);
    if (c->inbox) return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_prepare: already prepared");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    if (c->device >= 0) TMA_HIP(hipSetDevice(c->device));
    const int64_t cap = (max_words + 63) & ~(int64_t)63;
    const size_t bytes = (size_t)2 * c->world * cap * sizeof(unsigned long long);
    void *p = nullptr;
    // fine-grained: stores arriving over xGMI are visible to this GPU's system-scope loads without a cache flush in between
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        p = nullptr;
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            return tma::fail(TMA_ERR_HIP, "tma_comm_p2p_prepare: no fine-grained / uncached device memory for a %zu-byte inbox", bytes);
        }
    }
    c->inbox = static_cast<unsigned long long *>(p), c->cap = cap;
    hipError_t e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->err_host), 64, hipHostMallocMapped);
    if (e == hipSuccess) {
        *c->err_host = 0;
        e = hipHostGetDevicePointer(reinterpret_cast<void **>(&c->err_dev), c->err_host, 0);
    }
    hipIpcMemHandle_t h;
    if (e == hipSuccess && c->world > 1) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        p2p_release(c);
        return tma::fail(TMA_ERR_HIP, "tma_comm_p2p_prepare: %s", hipGetErrorString(e));
    }
    // the ticket: the IPC handle and the PCI bus id of the device the inbox lives on (a peer needs it to find that device among the ones IT sees)
    memset(ticket_out128, 0, 128);
    if (c->world > 1) memcpy(ticket_out128, &h, 64);
    int dev = 0;
    e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetPCIBusId(reinterpret_cast<char *>(ticket_out128) + 64, 47, dev);
    {  // bytes [112, 120): the host identity (bus ids are ~13 characters; the string stays NUL-terminated in front of it)
        const unsigned long long hid = p2p_host_id();
        memcpy(ticket_out128 + 112, &hid, 8);
    }
    if (e != hipSuccess) {
        p2p_release(c);
        return tma::fail(TMA_ERR_HIP, "tma_comm_p2p_prepare: %s", hipGetErrorString(e));
    }
    const char *ts = getenv("TMA_P2P_TIMEOUT_S");
    const double secs = ts ? atof(ts) : 120.0;
    c->timeout_ticks = (long long)((secs > 0.001 ? secs : 0.001) * 1e8);
    return TMA_OK;
}

int tma_comm_p2p_attach(tma_comm *c, const unsigned char *tickets) {
    if (!c || !c->inbox || (!tickets && c->world > 1)) return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_attach: prepare first / null tickets");
    if (c->attached) return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_attach: already attached");
    if (c->device >= 0) TMA_HIP(hipSetDevice(c->device));
    int mydev = 0;
    TMA_HIP(hipGetDevice(&mydev));
    char mybus[64] = {0};
    TMA_HIP(hipDeviceGetPCIBusId(mybus, 63, mydev));
    // every peer's device must be one THIS process can store to: the same device (several ranks on one GPU), or a visible one with peer access
    // (checked and switched on here -- a store through a mapping without it would be a memory fault, not an error code)
    for (int r = 0; r < c->world; r++) {
        if (r == c->rank) continue;
        char bus[64] = {0};
        memcpy(bus, tickets + 128 * (size_t)r + 64, 47);
        unsigned long long hid = 0;
        memcpy(&hid, tickets + 128 * (size_t)r + 112, 8);
        if (hid != p2p_host_id())
            return tma::fail(TMA_ERR_HIP, "tma_comm_p2p_attach: rank %d runs on another host (the peer exchange is one node's xGMI; its bus id %s means nothing here)", r, bus);
        if (strcmp(bus, mybus) == 0) continue;
        int pdev = -1, can = 0;
        if (hipDeviceGetByPCIBusId(&pdev, bus) != hipSuccess || pdev < 0) {
            (void)hipGetLastError();
            return tma::fail(TMA_ERR_HIP, "tma_comm_p2p_attach: rank %d's device %s is not visible to this process", r, bus);
        }
        if (hipDeviceCanAccessPeer(&can, mydev, pdev) != hipSuccess || !can) {
            (void)hipGetLastError();
            return tma::fail(TMA_ERR_HIP, "tma_comm_p2p_attach: no peer access from device %s to rank %d's device %s", mybus, r, bus);
        }
        const hipError_t pe = hipDeviceEnablePeerAccess(pdev, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
            (void)hipGetLastError();
            return tma::fail(TMA_ERR_HIP, "tma_comm_p2p_attach: hipDeviceEnablePeerAccess(%s) failed: %s", bus, hipGetErrorString(pe));
        }
        (void)hipGetLastError();
    }
    for (int r = 0; r < c->world; r++) {
        if (r == c->rank) {
            c->peer[r] = c->inbox;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, tickets + 128 * (size_t)r, 64);
        void *p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (int q = 0; q < r; q++)
                if (q != c->rank && c->peer[q]) (void)hipIpcCloseMemHandle(c->peer[q]), c->peer[q] = nullptr;
            return tma::fail(TMA_ERR_HIP, "tma_comm_p2p_attach: hipIpcOpenMemHandle of rank %d's inbox failed: %s", r, hipGetErrorString(e));
        }
        c->peer[r] = static_cast<unsigned long long *>(p);
    }
    c->attached = true;
    return TMA_OK;
}

int tma_comm_p2p_attach_local(tma_comm *c, tma_comm *const *peers) {
    if (!c || !c->inbox || !peers) return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_attach_local: prepare first / null peers");
    if (c->attached) return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_attach_local: already attached");
    for (int r = 0; r < c->world; r++) {
        const tma_comm *p = peers[r];
        if (!p || !p->inbox || p->world != c->world || p->rank != r || p->cap != c->cap || (r == c->rank && p != c))
            return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_attach_local: peer %d is not a prepared communicator of this world (same slot size, rank %d)", r, r);
    }
    for (int r = 0; r < c->world; r++) c->peer[r] = peers[r]->inbox;  // (same address space: the inbox itself, no IPC mapping; p2p_release skips them)
    c->attached = true, c->local_peers = true;
    return TMA_OK;
}

int tma_comm_p2p_enable(tma_comm *c, int on) {
    if (!c) return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_enable: null communicator");
    if (on && !c->attached) return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_enable: attach first");
    if (!on && !c->comm && c->world > 1) return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_enable: a communicator without RCCL cannot switch its peer exchange off");
    c->p2p_on = on != 0;
    return TMA_OK;
}

int tma_comm_p2p_set_timeout(tma_comm *c, double seconds) {
    if (!c || !(seconds > 0.0)) return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_set_timeout: bad argument");
    c->timeout_ticks = (long long)((seconds > 0.001 ? seconds : 0.001) * 1e8);  // wall_clock64(): 100 MHz
    return TMA_OK;
}

int tma_comm_p2p_status(tma_comm *c, int *enabled, int64_t *calls, int *timed_out, int64_t *slot_words) {
    if (!c) return tma::fail(TMA_ERR_INVALID, "tma_comm_p2p_status: null communicator");
    if (enabled) *enabled = c->p2p_on ? 1 : 0;
    if (calls) *calls = c->p2p_calls;
    if (timed_out) *timed_out = c->err_host ? *c->err_host : 0;
    if (slot_words) *slot_words = c->cap;
    return TMA_OK;
}

}  // extern "C"
