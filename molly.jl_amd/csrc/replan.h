// replan.h — the re-plan of a brick decomposition INSIDE the engine (SURVEY §8(e) "Migration: every neighbour rebuild … counts first, then
// payload"; the reference has no multi-device path, README.md:54).
//
// When the collective validity check says that the ghost plan cannot vouch for another prune (mhip_domain_run, action 2), every rank
//   A. MIGRATION   classifies its owned atoms by the brick their wrapped coordinate lies in (one stable radix sort over caller indices: stayers
//                  first, then the leavers grouped by destination), tells every rank how many atoms it sends to whom (one row of a count matrix
//                  stored into every rank's region header), stores the leavers' records — position, velocity, parameters, global id — straight
//                  into the destination's plan area, and compacts stayers + arrivals into the other half of its double-buffered atom arrays;
//   B. GHOST PLAN  selects, per neighbour direction, the owned atoms within the ghost reach of that face (per-block counts, one scan, one ordered
//                  write), exchanges the counts the same way, stores the new ghosts' coordinates (already shifted by the periodic image) and
//                  parameters into the peers' plan areas, and derives the per-step message tables (which local atom goes into which row of
//                  which peer's receive half, which received row fills which ghost slot) from the count matrix alone.
// No host call sits between the kernels; the host reads ONE small table at the end (new atom counts, error word) and runs the sort + search it
// would have run for an outer-list rebuild anyway.  Every order is fixed by construction — stayers in caller order, arrivals by (source rank,
// sender's caller order), ghosts by (source rank, direction, sender's caller order): the layout the host planner of molly.jl_amd/domain.py
// produces, so that both planners lead to the same sorted order and the same sums.  All waits for a peer are bounded (halo_xfer.h).
#pragma once
#include "kernels.h"

namespace mhip {

constexpr int RP_MAX_DIRS = 26;
constexpr int RP_W = XFER_MAX_RANKS;

// geometry of the decomposition as one rank sees it (by value to the kernels)
template <class T> struct ReplanGeom {
    int world, me, n_dirs, cm_rows;
    int grid[3];
    T box[3], brick[3];               // global box; brick side rounded to T (what the host planner divides by)
    T near_lo[3], near_hi[3];         // lo + r_ghost and hi − r_ghost of MY brick, rounded to T: x < near_lo / x >= near_hi select the two faces of an axis
    int cut[3];                       // the axis is cut (grid > 1): only such axes have faces
    signed char dvec[RP_MAX_DIRS][4]; // the neighbour directions over the cut axes, sorted by (peer rank, direction vector)
    int dir_peer[RP_MAX_DIRS];        // rank that owns the brick in that direction
    T dir_shift[RP_MAX_DIRS][3];      // what a ghost's coordinates get added on its way there (± a box length across the periodic faces)
};

// the one table of a re-plan: written on the device, copied to pinned host memory at the end
struct RpTab {
    int32_t n_stay, n_leave, n_arrive, n_owned, n_send, n_ghost, err, uni_bad, pad[8];
    int32_t leave_cnt[RP_W], leave_pre[RP_W + 1], mig_dst_off[RP_W], arr_from[RP_W], arr_pre[RP_W + 1];
    int32_t dir_cnt[32], dir_base[33];
    int32_t send_to[RP_W], send_pre[RP_W + 1], gh_dst_off[RP_W], gh_from[RP_W], gh_pre[RP_W + 1], step_dst_off[RP_W];
};
enum { RP_ERR_TIMEOUT = 1, RP_ERR_ATOMS = 2, RP_ERR_PLAN_AREA = 4, RP_ERR_ROWS = 8, RP_ERR_SEND = 16 };

struct RpPlanPtrs { unsigned char* area[XFER_MAX_RANKS]; };      // every rank's plan area (behind its two row halves), indexed by rank

__device__ inline int rp_pord(int rank, int me) { return rank < me ? rank : rank - 1; }      // ordinal of a peer among the other ranks in ascending order

// brick that owns a coordinate: wrap into the global box on every axis (the cut axes are open inside the engine), then floor(x / brick) per axis
template <class T> __device__ inline int rp_owner(T x, T y, T z, const ReplanGeom<T>& g, T w[3]) {
    w[0] = wrap_1d(x, g.box[0]); w[1] = wrap_1d(y, g.box[1]); w[2] = wrap_1d(z, g.box[2]);
    int r = 0, mul = 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int c = (int)M<T>::floor(w[d] / g.brick[d]);
        c = min(max(c, 0), g.grid[d] - 1);
        r += c * mul; mul *= g.grid[d];
    }
    return r;
}

// ---- A1: sort key per owned atom, in caller order: 0 = stays, 1 + destination rank = leaves ------------------------------------------------
template <class T>
__global__ void k_rp_owner_keys(int64_t n_owned, const typename Vec<T>::T4* __restrict__ pos, const int32_t* __restrict__ inv, ReplanGeom<T> g,
                                uint32_t* key, int32_t* val, RpTab* tab) {
    const int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (o >= n_owned) return;
    const auto p = pos[inv[o]];
    T w[3];
    const int dest = rp_owner<T>(p.x, p.y, p.z, g, w);
    key[o] = dest == g.me ? 0u : 1u + (uint32_t)dest;
    val[o] = (int32_t)o;
    if (dest != g.me) atomicAdd(&tab->leave_cnt[dest], 1);      // (a few atoms per thousand and re-plan)
}

// ---- A2 / B2: one row of the count matrix to every rank, everybody's rows back, the offsets that follow from the matrix -------------------
// PHASE 0: counts = leavers per destination; PHASE 1: ghost rows per peer.  One workgroup of 64 lanes, lane t = rank t.
template <int PHASE>
__global__ void __launch_bounds__(64) k_rp_exchange(XferPeers P, const XferHeader* mine, int world, int me, uint32_t seq, RpTab* tab, int cm_rows,
                                                    int n_owned_old, int cap_atoms, int rows_cap, int32_t* err, unsigned long long ticks) {
    __shared__ int32_t Msh[RP_W][RP_W + 3];
    const int t = threadIdx.x;
    const int32_t* cnt = PHASE == 0 ? tab->leave_cnt : tab->send_to;
    int own = 0;      // atoms this rank owns before the arrivals (phase 0) / after them (phase 1): what its capacity is checked against
    if (PHASE == 0) { int lv = 0; for (int d = 0; d < world; ++d) lv += cnt[d]; own = n_owned_old - lv; }
    else own = tab->n_owned;
    if (t < world) {      // my row into rank t's table; the diagonal (nobody sends to itself) carries my own atom count
        XferHeader* h = reinterpret_cast<XferHeader*>(P.region[t]);
        for (int d = 0; d < world; ++d) h->rp_cnt[PHASE][me][d] = d == me ? own : cnt[d];
        h->rp_cnt[PHASE][me][RP_W] = cap_atoms; h->rp_cnt[PHASE][me][RP_W + 1] = rows_cap;
        __threadfence_system();
        xfer_store_release(&h->rp_cnt_seq[PHASE][me], seq);
    }
    if (t < world && !xfer_wait(&mine->rp_cnt_seq[PHASE][t], seq, err, ticks, ((3 + PHASE) << 8) | t)) atomicOr(err, RP_ERR_TIMEOUT);
    __syncthreads();
    if (t < world) {
        for (int d = 0; d < world; ++d) Msh[t][d] = __hip_atomic_load(&mine->rp_cnt[PHASE][t][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        Msh[t][RP_W] = __hip_atomic_load(&mine->rp_cnt[PHASE][t][RP_W], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        Msh[t][RP_W + 1] = __hip_atomic_load(&mine->rp_cnt[PHASE][t][RP_W + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    // per destination rank t: what arrives there, where MY rows land there
    int e = 0;
    if (t < world) {
        int in_t = 0, before_me = 0;
        for (int q = 0; q < world; ++q) { if (q == t) continue; const int c = Msh[q][t]; in_t += c; if (q < me) before_me += c; }
        const int own_t = Msh[t][t];      // the diagonal carries the sender's own atom count (see the store below)
        const int cap_t = Msh[t][RP_W], rows_t = Msh[t][RP_W + 1];
        if (own_t + in_t > cap_t) e |= RP_ERR_ATOMS;
        int out_t = 0;
        for (int d = 0; d < world; ++d) if (d != t) out_t += Msh[t][d];
        (void)out_t;
        if (PHASE == 0) {
            if ((long long)in_t * XFER_MIG_WORDS > (long long)rows_t * XFER_PLAN_WORDS) e |= RP_ERR_PLAN_AREA;
            tab->mig_dst_off[t] = before_me;
            tab->arr_from[t] = t == me ? 0 : Msh[t][me];
        } else {
            if (in_t + (world - 1) * cm_rows > rows_t) e |= RP_ERR_ROWS;
            if (out_t + (world - 1) * cm_rows > rows_t) e |= RP_ERR_SEND;      // (its send tables hold rows_cap rows; every rank sees the same verdict)
            tab->gh_dst_off[t] = before_me;
            tab->step_dst_off[t] = before_me + rp_pord(me, t) * cm_rows;      // (the ranks before me among t's peers, each with its momentum rows)
            tab->gh_from[t] = t == me ? 0 : Msh[t][me];
        }
    }
    if (e) atomicOr(err, e);
    __syncthreads();
    if (t == 0) {
        int run = 0;
        if (PHASE == 0) {
            int lv = 0;
            for (int d = 0; d < world; ++d) { tab->leave_pre[d] = lv; lv += cnt[d]; }
            tab->leave_pre[world] = lv; tab->n_leave = lv; tab->n_stay = n_owned_old - lv;
            for (int q = 0; q < world; ++q) { tab->arr_pre[q] = run; run += tab->arr_from[q]; }
            tab->arr_pre[world] = run; tab->n_arrive = run; tab->n_owned = n_owned_old - lv + run;
        } else {
            int sd = 0;
            for (int d = 0; d < world; ++d) { tab->send_pre[d] = sd; sd += cnt[d]; }
            tab->send_pre[world] = sd; tab->n_send = sd;
            for (int q = 0; q < world; ++q) { tab->gh_pre[q] = run; run += tab->gh_from[q]; }
            tab->gh_pre[world] = run; tab->n_ghost = run;
        }
        tab->err = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// wait (one small workgroup, bounded) until every other rank's payload rows of this phase are complete in my plan area
template <int PHASE>
__global__ void __launch_bounds__(64) k_rp_wait_rows(const XferHeader* mine, int world, int me, uint32_t seq, RpTab* tab, int32_t* err, unsigned long long ticks) {
    const int t = threadIdx.x;
    if (t < world && t != me && !xfer_wait(&mine->rp_row_seq[PHASE][t], seq, err, ticks, ((5 + PHASE) << 8) | t)) atomicOr(err, RP_ERR_TIMEOUT);
    __syncthreads();
    if (t == 0) tab->err = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- A3: the leavers' records into their destinations' plan areas -------------------------------------------------------------------------
// sorted position n_stay + k holds the k-th leaver (grouped by destination, caller order inside a group); record = x y z q | vx vy vz m | σ ϵ | gid
template <class T>
__global__ void __launch_bounds__(256) k_rp_mig_send(const RpTab* __restrict__ tab, const uint32_t* __restrict__ key_sorted, const int32_t* __restrict__ val_sorted,
                                                     const int32_t* __restrict__ inv, const typename Vec<T>::T4* __restrict__ pos, const typename Vec<T>::T4* __restrict__ vel,
                                                     const typename Vec<T>::T2* __restrict__ lj, const int64_t* __restrict__ gid, ReplanGeom<T> g,
                                                     RpPlanPtrs A, XferPeers P, const int32_t* __restrict__ ranks, int n_ranks, uint32_t seq, unsigned int* done) {
    const int n_stay = tab->n_stay, n_leave = tab->n_leave;
    if (tab->err == 0) {
        for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_leave; k += gridDim.x * blockDim.x) {
            const int o = val_sorted[n_stay + k], dest = (int)key_sorted[n_stay + k] - 1;
            const int s = inv[o];
            const auto p = pos[s]; const auto v = vel[s]; const auto l = lj[s];
            T w[3]; (void)rp_owner<T>(p.x, p.y, p.z, g, w);
            T* r = reinterpret_cast<T*>(A.area[dest]) + (size_t)(tab->mig_dst_off[dest] + (k - tab->leave_pre[dest])) * XFER_MIG_WORDS;
            r[0] = w[0]; r[1] = w[1]; r[2] = w[2]; r[3] = p.w; r[4] = v.x; r[5] = v.y; r[6] = v.z; r[7] = v.w; r[8] = l.x; r[9] = l.y;
            const int64_t id = gid[o];
            if constexpr (sizeof(T) == 8) { r[10] = __longlong_as_double(id); r[11] = T(0); }
            else { r[10] = __int_as_float((int)(uint32_t)(id & 0xffffffffll)); r[11] = __int_as_float((int)(uint32_t)((uint64_t)id >> 32)); }
        }
    }
    xfer_announce_word(P, ranks, n_ranks, offsetof(XferHeader, rp_row_seq) + ((size_t)0 * XFER_MAX_RANKS + (size_t)g.me) * sizeof(uint32_t), seq, done);
}

// ---- A4: stayers (caller order) + arrivals (source rank, sender order) → the new atom arrays, new caller order = position --------------------
template <class T>
__global__ void __launch_bounds__(256) k_rp_compact(const RpTab* __restrict__ tab, const int32_t* __restrict__ val_sorted, const int32_t* __restrict__ inv,
                                                    const typename Vec<T>::T4* __restrict__ pos, const typename Vec<T>::T4* __restrict__ vel, const typename Vec<T>::T2* __restrict__ lj,
                                                    const int64_t* __restrict__ gid, ReplanGeom<T> g, const unsigned char* my_area,
                                                    typename Vec<T>::T4* pos_n, typename Vec<T>::T4* vel_n, typename Vec<T>::T2* lj_n, int64_t* gid_n) {
    if (tab->err != 0) return;
    const int n_stay = tab->n_stay, n_new = tab->n_owned;
    const T* rows = reinterpret_cast<const T*>(my_area);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_new; j += gridDim.x * blockDim.x) {
        if (j < n_stay) {
            const int o = val_sorted[j], s = inv[o];
            auto p = pos[s];
            T w[3]; (void)rp_owner<T>(p.x, p.y, p.z, g, w);
            p.x = w[0]; p.y = w[1]; p.z = w[2];
            pos_n[j] = p; vel_n[j] = vel[s]; lj_n[j] = lj[s]; gid_n[j] = gid[o];
        } else {
            const T* r = rows + (size_t)(j - n_stay) * XFER_MIG_WORDS;
            T q[XFER_MIG_WORDS];
#pragma unroll
            for (int c = 0; c < XFER_MIG_WORDS; ++c) q[c] = __hip_atomic_load(r + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            pos_n[j] = make4<T>(q[0], q[1], q[2], q[3]); vel_n[j] = make4<T>(q[4], q[5], q[6], q[7]); lj_n[j] = make2<T>(q[8], q[9]);
            if constexpr (sizeof(T) == 8) gid_n[j] = __double_as_longlong(q[10]);
            else gid_n[j] = (int64_t)((uint64_t)(uint32_t)__float_as_int(q[10]) | ((uint64_t)(uint32_t)__float_as_int(q[11]) << 32));
        }
    }
}

// ---- B1: which faces an owned atom is near → bit mask over the directions; per-block counts per direction --------------------------------------
template <class T> __device__ inline uint32_t rp_dir_mask(T x, T y, T z, const ReplanGeom<T>& g) {
    const T c[3] = {x, y, z};
    bool lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { lo[d] = g.cut[d] && c[d] < g.near_lo[d]; hi[d] = g.cut[d] && c[d] >= g.near_hi[d]; }
    uint32_t m = 0;
    for (int k = 0; k < g.n_dirs; ++k) {
        bool sel = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) { const int v = g.dvec[k][d]; sel = sel && (v == 0 || (v < 0 ? lo[d] : hi[d])); }
        m |= (sel ? 1u : 0u) << k;
    }
    return m;
}
template <class T>
__global__ void __launch_bounds__(256) k_rp_ghost_count(const RpTab* __restrict__ tab, const typename Vec<T>::T4* __restrict__ pos_n, ReplanGeom<T> g, uint32_t* mask_out, int32_t* blk_cnt) {
    __shared__ int32_t sh[4][RP_MAX_DIRS];
    const int n = tab->err == 0 ? tab->n_owned : 0;
    const int j = blockIdx.x * blockDim.x + threadIdx.x, wv = threadIdx.x >> 6;
    uint32_t m = 0;
    if (j < n) { const auto p = pos_n[j]; m = rp_dir_mask<T>(p.x, p.y, p.z, g); mask_out[j] = m; }
    for (int k = 0; k < g.n_dirs; ++k) {
        const int c = __popcll(__builtin_amdgcn_ballot_w64(((m >> k) & 1u) != 0u));
        if ((threadIdx.x & 63) == 0) sh[wv][k] = c;
    }
    __syncthreads();
    if ((int)threadIdx.x < g.n_dirs) blk_cnt[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// ---- B2a: per direction the exclusive scan of the block counts (one wave per direction), then the per-peer totals -------------------------------
template <class T>
__global__ void __launch_bounds__(1024) k_rp_ghost_scan(RpTab* tab, ReplanGeom<T> g, int nb, const int32_t* __restrict__ blk_cnt, int32_t* blk_off) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = wv; k < g.n_dirs; k += (int)(blockDim.x >> 6)) {
        int run = 0;
        for (int b0 = 0; b0 < nb; b0 += 64) {
            const int b = b0 + lane;
            const int v = b < nb ? blk_cnt[(size_t)k * nb + b] : 0;
            int x = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(x, o, 64); if (lane >= o) x += u; }
            if (b < nb) blk_off[(size_t)k * nb + b] = run + x - v;
            run += __shfl(x, 63, 64);
        }
        if (lane == 0) tab->dir_cnt[k] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int d = 0; d < g.world; ++d) tab->send_to[d] = 0;
        for (int k = 0; k < g.n_dirs; ++k) { tab->dir_base[k] = run; run += tab->dir_cnt[k]; tab->send_to[g.dir_peer[k]] += tab->dir_cnt[k]; }
        tab->dir_base[g.n_dirs] = run;
    }
}

// ---- B3: the ghost rows, ordered by (direction, caller index): the per-step send tables and the new ghosts' records into the peers' plan areas ---
// per-step send buffer of mhip_halo_plan: each peer's coordinate rows followed by cm_rows momentum rows → row k of my ghost order sits at
// k + ordinal(peer) · cm_rows.  The LAST workgroup writes the momentum rows' entries.  record = x + shift, y + shift, z + shift, q, σ, ϵ, m, 0
template <class T>
__global__ void __launch_bounds__(256) k_rp_ghost_send(const RpTab* __restrict__ tab, const typename Vec<T>::T4* __restrict__ pos_n, const typename Vec<T>::T4* __restrict__ vel_n,
                                                       const typename Vec<T>::T2* __restrict__ lj_n, const uint32_t* __restrict__ mask_in, const int32_t* __restrict__ blk_off, int nb,
                                                       ReplanGeom<T> g, int32_t* send_idx, T* send_shift, int32_t* send_cm_pos, int32_t* row_peer, int32_t* row_dst,
                                                       RpPlanPtrs A, XferPeers P, const int32_t* __restrict__ ranks, int n_ranks, uint32_t seq, unsigned int* done) {
    __shared__ int32_t sh[4][RP_MAX_DIRS];
    const size_t word = offsetof(XferHeader, rp_row_seq) + ((size_t)1 * XFER_MAX_RANKS + (size_t)g.me) * sizeof(uint32_t);
    const bool ok = tab->err == 0;
    if ((int)blockIdx.x == nb) {      // the momentum rows: cm_rows per peer behind its coordinate rows
        if (ok) for (int q = threadIdx.x; q < (g.world - 1) * g.cm_rows; q += blockDim.x) {
            const int pi = q / g.cm_rows, r = q - pi * g.cm_rows, peer = pi < g.me ? pi : pi + 1;
            const int row = tab->send_pre[peer] + tab->send_to[peer] + pi * g.cm_rows + r;
            send_idx[row] = -1 - r; send_shift[3 * (size_t)row] = T(0); send_shift[3 * (size_t)row + 1] = T(0); send_shift[3 * (size_t)row + 2] = T(0);
            row_peer[row] = peer; row_dst[row] = tab->step_dst_off[peer] + tab->send_to[peer] + r;
            send_cm_pos[q] = row;
        }
        xfer_announce_word(P, ranks, n_ranks, word, seq, done);
        return;
    }
    const int n = ok ? tab->n_owned : 0;
    const int j = blockIdx.x * blockDim.x + threadIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t m = j < n ? mask_in[j] : 0u;
    for (int k = 0; k < g.n_dirs; ++k) {
        const int c = __popcll(__builtin_amdgcn_ballot_w64(((m >> k) & 1u) != 0u));
        if (lane == 0) sh[wv][k] = c;
    }
    __syncthreads();
    // (the ballots need the whole wave: every lane runs the loop, selected lanes store)
    for (int k = 0; k < g.n_dirs; ++k) {
        const bool sel = ((m >> k) & 1u) != 0u;
        const unsigned long long ball = __builtin_amdgcn_ballot_w64(sel);
        if (sel) {
            int before = __popcll(ball & ((1ull << lane) - 1ull));
            for (int w = 0; w < wv; ++w) before += sh[w][k];
            const int kk = tab->dir_base[k] + blk_off[(size_t)k * nb + blockIdx.x] + before;      // my number in this rank's ghost order
            const int peer = g.dir_peer[k], pi = rp_pord(peer, g.me), within = kk - tab->send_pre[peer];
            const int row = kk + pi * g.cm_rows;
            const auto p = pos_n[j]; const auto l = lj_n[j]; const T mass = vel_n[j].w;
            const T sx = g.dir_shift[k][0], sy = g.dir_shift[k][1], sz = g.dir_shift[k][2];
            send_idx[row] = j; send_shift[3 * (size_t)row] = sx; send_shift[3 * (size_t)row + 1] = sy; send_shift[3 * (size_t)row + 2] = sz;
            row_peer[row] = peer; row_dst[row] = tab->step_dst_off[peer] + within;
            T* r = reinterpret_cast<T*>(A.area[peer]) + (size_t)(tab->gh_dst_off[peer] + within) * XFER_GHOST_WORDS;
            r[0] = p.x + sx; r[1] = p.y + sy; r[2] = p.z + sz; r[3] = p.w; r[4] = l.x; r[5] = l.y; r[6] = mass; r[7] = T(0);
        }
    }
    xfer_announce_word(P, ranks, n_ranks, word, seq, done);
}

// ---- B4: the new ghosts behind the owned atoms, the per-step receive table, and the one-type check over every local atom -----------------------
// per-step receive half: peer by peer (ascending rank) its coordinate rows, then cm_rows momentum rows → ghost g of the peer with ordinal pi sits in
// row g + pi · cm_rows; recv_dst[row] = g, or −1 − (pi · cm_rows + r) for the momentum rows (mhip_halo_plan)
template <class T>
__global__ void __launch_bounds__(256) k_rp_ghost_recv(RpTab* tab, ReplanGeom<T> g, const unsigned char* my_area, typename Vec<T>::T4* pos_n, typename Vec<T>::T4* vel_n,
                                                       typename Vec<T>::T2* lj_n, int32_t* recv_dst) {
    if (tab->err != 0) return;
    const int n_owned = tab->n_owned, n_ghost = tab->n_ghost;
    const T* rows = reinterpret_cast<const T*>(my_area);
    const auto l0 = lj_n[0];
    bool bad = false;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_owned + n_ghost; j += gridDim.x * blockDim.x) {
        typename Vec<T>::T2 l;
        if (j < n_owned) l = lj_n[j];
        else {
            const int gi = j - n_owned;
            int src = 0;
            while (src + 1 < g.world && tab->gh_pre[src + 1] <= gi) ++src;
            const T* r = rows + (size_t)gi * XFER_GHOST_WORDS;
            T q[7];
#pragma unroll
            for (int c = 0; c < 7; ++c) q[c] = __hip_atomic_load(r + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            pos_n[j] = make4<T>(q[0], q[1], q[2], q[3]); vel_n[j] = make4<T>(T(0), T(0), T(0), q[6]);
            l = make2<T>(q[4], q[5]); lj_n[j] = l;
            recv_dst[gi + rp_pord(src, g.me) * g.cm_rows] = gi;
        }
        bad = bad || !(l.x == l0.x && l.y == l0.y);
    }
    if (blockIdx.x == 0) for (int q = threadIdx.x; q < (g.world - 1) * g.cm_rows; q += blockDim.x) {
        const int pi = q / g.cm_rows, r = q - pi * g.cm_rows, peer = pi < g.me ? pi : pi + 1;
        recv_dst[tab->gh_pre[peer] + tab->gh_from[peer] + pi * g.cm_rows + r] = -1 - (pi * g.cm_rows + r);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0 && __hip_atomic_load(&tab->uni_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&tab->uni_bad, 1);
}

// caller-order export of what the host planner keeps per owned atom: global id and {q, σ, ϵ, m}
template <class T>
__global__ void k_rp_export(int64_t n_owned, const int32_t* __restrict__ inv, const typename Vec<T>::T4* __restrict__ pos, const typename Vec<T>::T4* __restrict__ vel,
                            const typename Vec<T>::T2* __restrict__ lj, const int64_t* __restrict__ gid, int64_t* gid_out, T* par4_out) {
    const int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (o >= n_owned) return;
    const int s = inv[o];
    if (gid_out) gid_out[o] = gid[o];
    if (par4_out) { par4_out[4 * o] = pos[s].w; par4_out[4 * o + 1] = lj[s].x; par4_out[4 * o + 2] = lj[s].y; par4_out[4 * o + 3] = vel[s].w; }
}

}  // namespace mhip
