// halo_xfer.h — the ghost exchange INSIDE the engine ("direct peer stores into pre-exchanged IPC buffers").
//
// Every rank owns a RECEIVE REGION in fine-grained device memory and, through hipIpcOpenMemHandle, a pointer to every peer's region
// (one process per GPU; the pointers are exchanged once, when the decomposition is set up).  The kernel that packs a step's coordinates
// stores each peer's rows straight into that peer's region — over xGMI, no host call, no collective — and its last block raises a
// sequence word there; the unpack kernel of the receiving side spins on its senders' words (bounded: a peer that never arrives raises
// an error flag instead of hanging the GPU) before it scatters the rows into its ghost slots.  A region has two halves used by the
// parity of the exchange number: a fast peer may deliver exchange e + 1 while this rank still reads exchange e, never e + 2 (that
// needs this rank's own e + 1, which it sends after reading e).  The collective validity check of the pair lists (MAX over the ranks
// of three floats every rebuild interval) travels the same way into a small table of the region.
#pragma once
#include "common.h"

namespace mhip {

constexpr int XFER_MAX_RANKS = 64;
// header of a receive region (the row halves follow at rows_off)
struct XferHeader {
    uint32_t seq_in[2][XFER_MAX_RANKS];        // [parity][sender rank]: number of the last exchange whose rows are complete in that half
    uint32_t plan_seq[2][XFER_MAX_RANKS];      // [parity][rank]: number of the last validity check whose triple is in plan_val
    float plan_val[2][XFER_MAX_RANKS][4];      // {max |x − x_plan|², max |x − x_prune|², max |v|², –} of that rank
    int64_t rows_cap;                          // rows per half of THIS region, written by its owner: a sender checks its segments against it
    uint64_t dev_key;                          // which physical device the owner runs on (PCI domain / bus / device, + 1): ranks that find a peer on their OWN device
                                               // — several processes sharing one GPU, the test set-up — never wait for it inside a kernel that fills the GPU (engine.hip, halo_fused)
    // the re-plan inside the engine (replan.h): two collective count exchanges per re-plan (leavers per destination, ghost rows per peer) — every rank
    // stores its row of the count matrix into every rank's table — and the sequence words of the payload rows that follow them into the plan area
    uint32_t rp_cnt_seq[2][XFER_MAX_RANKS];    // [phase][sender]: number of the last re-plan whose count row is in rp_cnt
    uint32_t rp_row_seq[2][XFER_MAX_RANKS];    // [phase][sender]: number of the last re-plan whose payload rows are complete in the plan area
    int32_t rp_cnt[2][XFER_MAX_RANKS][XFER_MAX_RANKS + 2];   // [phase][sender][destination]; [..][XFER_MAX_RANKS] = the sender's atom capacity, [.. + 1] = its region's rows_cap
};
constexpr size_t XFER_ROWS_OFF = (sizeof(XferHeader) + 255) & ~(size_t)255;
// behind the two row halves: the PLAN AREA, rows_cap × XFER_PLAN_WORDS reals — payload of a re-plan (migrating atoms: 12 words each; new ghosts: 8 words each)
constexpr int XFER_PLAN_WORDS = 8;
constexpr int XFER_MIG_WORDS = 12, XFER_GHOST_WORDS = 8;
template <class T> constexpr size_t xfer_plan_off(int64_t rows_cap) { return XFER_ROWS_OFF + 2 * (size_t)rows_cap * 3 * sizeof(T); }
template <class T> constexpr size_t xfer_region_bytes(int64_t rows_cap) { return xfer_plan_off<T>(rows_cap) + (size_t)rows_cap * XFER_PLAN_WORDS * sizeof(T); }

__device__ inline void xfer_store_release(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline uint32_t xfer_load_acquire(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
// wait until *p has reached `want` (sequence numbers only grow; wrap-safe compare); false after `ticks` of the 100 MHz wall clock
// (MOLLYHIP_XFER_TIMEOUT_MS, 2 s unless set) — or AT ONCE when the error word is already raised: after one time-out every later
// launch of the chunk is queued already, and each of them waiting its own full time-out turned one lost peer into minutes of stall
// err: int32[4] — [0] the error flags the caller raises, [1..3] filled by the FIRST wait that gave up: who (caller's tag << 8 | the peer waited for) + 1, the
// sequence number wanted, the one last seen (a peer one exchange behind is slow; one that never moved is gone)
__device__ inline bool xfer_wait(const uint32_t* p, uint32_t want, int32_t* err, unsigned long long ticks, int who = 0) {
    if ((int32_t)(xfer_load_acquire(p) - want) >= 0) return true;
    if (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    const unsigned long long t0 = wall_clock64();
    uint32_t seen = 0;
    while ((int32_t)((seen = xfer_load_acquire(p)) - want) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > ticks) {
            if (err && atomicCAS(&err[1], 0, who + 1) == 0) { err[2] = (int32_t)want; err[3] = (int32_t)seen; }
            return false;
        }
    }
    return true;
}

struct XferPeers {                              // by value to the kernels: the mapped regions, indexed by rank
    unsigned char* region[XFER_MAX_RANKS];
};

// what a kernel needs to store rows into the peers' regions and to announce them (k_halo_pack), or to wait for the peers' rows
// (k_halo_unpack) — so that packing + sending, and waiting + unpacking, are ONE launch each
struct XferSend {
    const int32_t* row_peer; const int32_t* row_dst;     // per send row: destination rank, row in that rank's half (nullptr: plain local pack)
    XferPeers P; int64_t rows_cap; int parity; uint32_t seq; int my_rank; const int32_t* peers; int n_peers; unsigned int* done;
};
struct XferWait { const XferHeader* mine; int parity; uint32_t seq; const int32_t* peers; int n_peers; int32_t* err; unsigned long long ticks; };

// the last block of a launch to get here raises this rank's sequence word at every peer (all blocks have fenced their stores before)
__device__ inline void xfer_announce(const XferSend& X) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int before = __hip_atomic_fetch_add(X.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == gridDim.x - 1) {
            __threadfence_system();
            for (int q = 0; q < X.n_peers; ++q)
                xfer_store_release(&reinterpret_cast<XferHeader*>(X.P.region[X.peers[q]])->seq_in[X.parity][X.my_rank], X.seq);
            __hip_atomic_store(X.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// the same for any sequence word of the header (byte offset `word_off` of the sender's own word inside an XferHeader): the payload rows of a re-plan
__device__ inline void xfer_announce_word(XferPeers P, const int32_t* ranks, int n_ranks, size_t word_off, uint32_t seq, unsigned int* done) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int before = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == gridDim.x - 1) {
            __threadfence_system();
            for (int q = 0; q < n_ranks; ++q) xfer_store_release(reinterpret_cast<uint32_t*>(P.region[ranks[q]] + word_off), seq);
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// every block waits for the senders' words itself (a handful of uncached loads), then all its threads go on
__device__ inline void xfer_wait_block(const XferWait& W) {
    if ((int)threadIdx.x < W.n_peers && !xfer_wait(&W.mine->seq_in[W.parity][W.peers[threadIdx.x]], W.seq, W.err, W.ticks, (1 << 8) | W.peers[threadIdx.x])) atomicOr(W.err, 1);
    __syncthreads();
}

// every peer's rows of exchange `seq` are in: one small workgroup that does nothing else (ranks sharing ONE device put it in front of the fused step, whose own
// waits then find the words already there — a grid that fills the GPU must not spin for a process that needs the same GPU to get there)
[[maybe_unused]] static __global__ void k_xfer_wait_all(XferWait W) {
    if ((int)threadIdx.x < W.n_peers && !xfer_wait(&W.mine->seq_in[W.parity][W.peers[threadIdx.x]], W.seq, W.err, W.ticks, (1 << 8) | W.peers[threadIdx.x])) atomicOr(W.err, 1);
}

// validity check of the pair lists over all ranks: my triple into every rank's table (my own included) …
[[maybe_unused]] static __global__ void k_plan_push(const float* __restrict__ mine3, XferPeers P, int world, int my_rank, int parity, uint32_t seq) {
    const int r = threadIdx.x;
    if (r >= world) return;
    XferHeader* h = reinterpret_cast<XferHeader*>(P.region[r]);
    h->plan_val[parity][my_rank][0] = mine3[0]; h->plan_val[parity][my_rank][1] = mine3[1]; h->plan_val[parity][my_rank][2] = mine3[2];
    __threadfence_system();
    xfer_store_release(&h->plan_seq[parity][my_rank], seq);
}
// … and, once every rank's has arrived, their maximum → out3 (device) and host3 (pinned host memory, read behind an event).  host3[3]
// carries the error word (a time-out here or in any exchange before): the host reads it with the triple and gives the chunk up there
[[maybe_unused]] static __global__ void k_plan_reduce(const XferHeader* mine, int world, int parity, uint32_t seq, float* out3, float* host3, int32_t* err, unsigned long long ticks) {
    __shared__ float sh[XFER_MAX_RANKS][3];
    const int r = threadIdx.x;
    if (r < world) {
        if (!xfer_wait(&mine->plan_seq[parity][r], seq, err, ticks, (2 << 8) | r)) atomicOr(err, 2);
        for (int c = 0; c < 3; ++c) sh[r][c] = __hip_atomic_load(&mine->plan_val[parity][r][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (r < 3) {
        float m = sh[0][r];
        for (int q = 1; q < world; ++q) m = fmaxf(m, sh[q][r]);        // (+inf propagates: a rank that must re-plan makes everybody re-plan)
        if (out3) out3[r] = m;
        if (host3) host3[r] = m;
    }
    if (r == 3 && host3) host3[3] = (float)__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace mhip
