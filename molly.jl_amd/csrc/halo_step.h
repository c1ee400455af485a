// halo_step.h — the tables of the fused ghosted step (kernels.h, HaloStep): made on the device from the per-step message tables of the current ghost plan
// (mhip_halo_plan / mhip_halo_routes, or the in-engine re-plan's) and from the tiles of the inner pair list.  Two lifetimes:
//   per plan and sort  (k_hx_rows, k_hx_send_count / _fill, k_hx_cm_dst, k_hx_blk_send): which row of the receive half holds which ghost slot and which peer's sums;
//                      per owned atom (sorted slot) the rows it is sent to, as absolute addresses in the peers' regions + the periodic shift
//   per prune          (k_hx_tile, k_hx_order): the tile by source, the per-block flags, the order in which the launch's workgroups take the blocks
#pragma once
#include "kernels.h"

namespace mhip {

// recv table (row → ghost index d >= 0, or −1 − (peer·cm_rows + r)) → ghost_row[sorted ghost slot − n_owned] = row, cm_row[peer·cm_rows + r] = row
[[maybe_unused]] static __global__ void k_hx_rows(int64_t n_rows, const int32_t* __restrict__ recv_dst, int64_t first, const int32_t* __restrict__ inv, int64_t n_owned,
                                                  int32_t* ghost_row, int32_t* cm_row) {
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= n_rows) return;
    const int d = recv_dst[k];
    if (d >= 0) ghost_row[inv[first + d] - n_owned] = (int32_t)k;      // (ghosts sort behind every owned atom)
    else cm_row[-1 - d] = (int32_t)k;
}
// send table (row → caller index of an owned atom, or < 0: a centre-of-mass row): rows per sorted slot …
[[maybe_unused]] static __global__ void k_hx_send_count(int64_t n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ inv, int32_t* cnt) {
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= n_rows) return;
    const int i = idx[k];
    if (i >= 0) atomicAdd(&cnt[inv[i]], 1);
}
// … and, behind the scan, the entries (the order of an atom's rows is immaterial: they are different rows)
[[maybe_unused]] static __global__ void k_hx_send_fill(int64_t n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ inv, const float* __restrict__ shift,
                                                       const int32_t* __restrict__ row_peer, const int32_t* __restrict__ row_dst, XferPeers P,
                                                       const int32_t* __restrict__ start, int32_t* cursor, HaloSend* out) {
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= n_rows) return;
    const int i = idx[k];
    if (i < 0) return;
    const int s = inv[i];
    const int at = start[s] + atomicAdd(&cursor[s], 1);
    HaloSend e;
    e.dst = reinterpret_cast<float*>(P.region[row_peer[k]] + XFER_ROWS_OFF) + 3 * (size_t)row_dst[k];
    e.sx = shift[3 * k]; e.sy = shift[3 * k + 1]; e.sz = shift[3 * k + 2]; e.pad = 0;
    out[at] = e;
}
[[maybe_unused]] static __global__ void k_hx_cm_dst(int n_cm, const int32_t* __restrict__ cm_pos, const int32_t* __restrict__ row_peer, const int32_t* __restrict__ row_dst, XferPeers P, float** cm_dst) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_cm) return;
    const int k = cm_pos[q];
    cm_dst[q] = reinterpret_cast<float*>(P.region[row_peer[k]] + XFER_ROWS_OFF) + 3 * (size_t)row_dst[k];
}
// per block of BI sorted atoms: is any of them sent anywhere?
[[maybe_unused]] static __global__ void k_hx_blk_send(int n_blocks, int BI, int64_t n_owned, const int32_t* __restrict__ start, int32_t* blk_send) {
    const int b = blockIdx.x;
    const int64_t lo = (int64_t)b * BI, hi = min(lo + BI, n_owned);
    if (threadIdx.x == 0) blk_send[b] = (lo < hi && start[hi] > start[lo]) ? 1 : 0;
}
// the tile of every block by source + the block's flags (one 256-lane workgroup per block)
[[maybe_unused]] static __global__ void k_hx_tile(int n_blocks, int T_cap, int64_t n_owned, const int32_t* __restrict__ tile_idx, const int32_t* __restrict__ tile_cnt,
                                                  const int32_t* __restrict__ ghost_row, const int32_t* __restrict__ blk_send, int32_t* tsrc, int32_t* flags) {
    __shared__ int any;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    const int n = tile_cnt[b];
    bool g = false;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int s = tile_idx[(int64_t)b * T_cap + t];
        const bool gh = s >= n_owned;
        tsrc[(int64_t)b * T_cap + t] = gh ? -1 - ghost_row[s - n_owned] : s;
        g = g || gh;
    }
    if (g) any = 1;
    __syncthreads();
    if (threadIdx.x == 0) flags[b] = any | (blk_send[b] << 1);
}
// workgroup w of the pass (behind its head) runs on XCD (w + 1) mod 8 and takes a block of the XCD's contiguous run of Hilbert-ordered blocks (k_forces): the
// blocks of a run WITHOUT ghosts first, in their order, then the others — those wait for the peers' rows and should not hold compute units while there is other work.
// One workgroup per run; order[j · 8 + k] = j-th block of run k, −1 behind the run's end.
[[maybe_unused]] static __global__ void __launch_bounds__(256) k_hx_order(int n_blocks, int bpx, const int32_t* __restrict__ flags, int32_t* order) {
    __shared__ int n_first, run;
    const int k = blockIdx.x, lo = k * bpx, hi = min(lo + bpx, n_blocks), tid = threadIdx.x;
    for (int j = tid; j < bpx; j += blockDim.x) order[j * 8 + k] = -1;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {      // pass 0 places the blocks without ghosts, pass 1 the others behind them
        if (tid == 0) run = pass == 0 ? 0 : n_first;
        __syncthreads();
        for (int base = lo; base < hi; base += blockDim.x) {
            const int b = base + tid;
            const bool take = b < hi && ((flags[b] & 1) != 0) == (pass == 1);
            // ordered compaction of this chunk: rank inside the wave by ballot, waves one after the other
            const unsigned long long m = __ballot(take);
            const int in_wave = __popcll(m & ((1ull << (tid & 63)) - 1ull));
            __shared__ int wcnt[4];
            if ((tid & 63) == 0) wcnt[tid >> 6] = __popcll(m);
            __syncthreads();
            int off = run;
            for (int w = 0; w < (tid >> 6); ++w) off += wcnt[w];
            if (take) order[(off + in_wave) * 8 + k] = b;
            __syncthreads();
            if (tid == 0) run += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
        }
        if (pass == 0 && tid == 0) n_first = run;
        __syncthreads();
    }
}

}  // namespace mhip
