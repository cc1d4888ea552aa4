// tma_p2p.h -- peer exchange: the gradient all-reduce of the data-parallel epoch as direct xGMI stores between the GPUs of one node.
//
// The collective on the path (the per-minibatch SUM of the flat f32 policy gradient, 37 KB for the 64 x 64 policy, 320 times
// per headline iteration) is latency, not bandwidth: a ring all-reduce is 2 (world - 1) dependent hops plus a kernel launch in the middle of
// the minibatch chain, for a message every GPU could hand to every other GPU in ONE hop -- xGMI is point to point, all seven links of a
// GPU work at once.  So: every rank owns an INBOX in fine-grained device memory, exported to its peers (hipIpcGetMemHandle); the kernel that
// produces the reduced gradient (slab_reduce_kernel) stores each element straight into its slot of EVERY rank's inbox, and the kernel that
// needs the summed gradient next (the sum-of-squares pass before the optimizer step) reads the `world` slots of its own inbox and adds
// them in rank order.  No collective launch, no ring, the same summation order on every rank -> bit-identical replicas.
//
// Protocol (what makes it safe without fences): a slot word is 8 bytes {payload: 32 bits | sequence number: 32 bits}, written with ONE
// 8-byte store and read with ONE 8-byte load (both system scope), so a reader sees either the old word or the new one, never a mix; the
// sequence number is the all-reduce's index (every rank counts the same calls), so "new" is recognisable per word and no ordering between
// different words is assumed.  Two parities of slots (sequence & 1): a rank can be at most one all-reduce ahead of a peer that is still
// reading (it cannot start all-reduce k + 2 before it has read every rank's words of k + 1, which the peer only writes after it finished
// reading k).  A reader that does not see its words within the timeout raises the communicator's error flag (host-mapped) and returns;
// nothing spins for ever.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct tma_comm;

namespace tma {

constexpr int P2P_MAX_WORLD = 8;  // one node

struct PeerPush {                          // by value into the producing kernel; world == 0: no exchange
    unsigned long long *dst[P2P_MAX_WORLD];  // rank d's inbox, at [parity][this rank][0]
    int world;
    uint32_t seq;
};

struct PeerPull {                    // by value into the consuming kernel; world == 0: no exchange
    const unsigned long long *slot;  // own inbox at [parity][0][0]
    int64_t stride;                  // words per rank slot
    int world;
    uint32_t seq;
    int *err;                        // host-mapped flag: set to 1 by a reader that timed out
    long long timeout;               // in wall_clock64() ticks (100 MHz)
};

__device__ __forceinline__ void p2p_push(const PeerPush &p, int64_t i, uint32_t bits) {
    const unsigned long long w = ((unsigned long long)p.seq << 32) | (unsigned long long)bits;
    for (int d = 0; d < p.world; d++) __hip_atomic_store(p.dst[d] + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the `world` words of element i, all loads of a round in flight together.  false after a timeout -- or AT ONCE when the communicator's error
// flag is already up (an earlier pull of this or any queued exchange gave up: the ~320 exchanges a learn() iteration has queued then drain
// in microseconds instead of spinning a timeout each); the words are then whatever was there and the caller must not use them.
__device__ __forceinline__ bool p2p_wait(const PeerPull &q, int64_t i, unsigned long long (&w)[P2P_MAX_WORLD]) {
    long long t0 = 0;
    for (int round = 0;; round++) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < P2P_MAX_WORLD; r++) {
            if (r < q.world) {
                w[r] = __hip_atomic_load(q.slot + (int64_t)r * q.stride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = ok && (uint32_t)(w[r] >> 32) == q.seq;
            }
        }
        if (ok) return true;
        if (round == 0) {
            if (__hip_atomic_load(q.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;
            t0 = (long long)wall_clock64();
        } else if ((round & 63) == 0) {
            if (__hip_atomic_load(q.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;
            if ((long long)wall_clock64() - t0 > q.timeout) {
                __hip_atomic_store(q.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// sum over the ranks in rank order (world 1: the rank's own value, bit for bit).  After a timeout the result is NaN: the optimizer step
// that consumes it turns every parameter into NaN, which no later check can miss -- a sum of stale words would train on silently.
__device__ __forceinline__ float p2p_pull_f32(const PeerPull &q, int64_t i) {
    unsigned long long w[P2P_MAX_WORLD];
    if (!p2p_wait(q, i, w)) return __uint_as_float(0x7FC00000u);
    float s = __uint_as_float((uint32_t)w[0]);
#pragma unroll
    for (int r = 1; r < P2P_MAX_WORLD; r++)
        if (r < q.world) s += __uint_as_float((uint32_t)w[r]);
    return s;
}

// Host side (tma_comm.hip).  tma_comm_p2p_next: descriptors of the NEXT all-reduce of `count` 32-bit payload words through the peer
// exchange (advances the sequence number; the caller must enqueue exactly one push and one pull of `count` words with them, in this order,
// on `stream`).  tma_comm_p2p_ready: the exchange is attached, enabled and `count` words fit a slot.  Timing (tma_comm_timing): events
// around the consuming kernel -- what the chain waits for once the producing kernel is done.
bool tma_comm_p2p_ready(const tma_comm *c, int64_t count_words);
int tma_comm_p2p_next(tma_comm *c, int64_t count_words, PeerPush *push, PeerPull *pull);
hipStream_t tma_comm_bound_stream(const tma_comm *c);  // tma_comm_bind_stream's stream
int tma_comm_time_begin(tma_comm *c, hipStream_t s);  // returns 1 when an event pair was opened (close it with tma_comm_time_end)
void tma_comm_time_end(tma_comm *c, hipStream_t s);

}  // namespace tma
