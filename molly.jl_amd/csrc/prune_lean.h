// prune_lean.h — the prune of the dual pair list as a kernel of its own for the fp32 one-type fluids: distance test + emission only.
//
// The PRUNE variant of k_forces walks the outer list (twice the entries of the inner one), evaluates forces on the way and emits the
// inner list: 0.52 ms at 1M atoms, 2.9× the time per entry of a plain pass.  Measured (tools/gpu_prune_ab.sh): leaving out the LJ
// arithmetic changes nothing (−8 µs), leaving out the emission −136 µs; what it waits for is latency at half the occupancy — its tile
// of 3 700 atoms as three fp32 arrays takes 49 KiB of LDS (two 512-lane blocks per CU) and it needs 115 VGPRs.  A prune only has to
// produce a SUPERSET of the pairs within rc_max + skin, so this kernel keeps the tile as 16-bit fixed-point coordinates — one 8-byte
// LDS word per atom, one ds_read_b64 per partner instead of three ds_read_b32 —, tests distances in integers (v_pk_sub_i16,
// v_dot2_i32_i16; two quantisation units of slack on the radius) and carries no force state: four blocks per CU, eight waves per SIMD.
// The step's forces then come from an ordinary pass over the fresh inner list.
#pragma once
#include "kernels.h"

#ifndef MHIP_LEXP
#define MHIP_LEXP 0     // timing experiments (tools/gpu_prune_ab.sh)
#endif

namespace mhip {

constexpr float PRUNE_Q = 2048.f;              // fixed-point units per nm (4.9e-4 nm), coordinates clamped to ±12000 units = ±5.9 nm around the
                                               // block centre: projecting onto a cube never lengthens a distance — still a superset — and keeps
                                               // every difference inside int16 and every squared distance inside int32
__host__ __device__ inline size_t prune_lean_lds_bytes(int t_cap, int nthr) { return (size_t)((t_cap + 3) & ~1) * 8 + (size_t)(((t_cap + 8) & ~7) * 2) + ((size_t)nthr + 4) * 4 + 8 * 8 * 4 + 64; }

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(1024) k_prune_lean(ForceArgs<float> A) {
    extern __shared__ __align__(32) unsigned char smem[];
    const GridP<float>& G = A.G;
    const int wg = blockIdx.x;
    const int b = (wg & 7) * A.blocks_per_xcd + (wg >> 3);
    if (b >= A.n_blocks) return;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int tile_n = A.tile_cnt[b];
    uint2* l_q = reinterpret_cast<uint2*>(smem);                                   // {x | y << 16, z} as int16 fixed point
    uint16_t* l_new = reinterpret_cast<uint16_t*>(l_q + ((A.T_lds + 3) & ~1));   // (an even number of 8-byte words: the boxes behind stay 16-byte aligned)
    int32_t* l_scan = reinterpret_cast<int32_t*>(l_new + ((A.T_lds + 8) & ~7));
    float* l_box = reinterpret_cast<float*>(l_scan + nthr + 4);
    const float4 ctr = A.blk_center[b];
    const int li = tid & (A.BI - 1), js = tid >> A.BI_shift;
    const int64_t si = (int64_t)b * A.BI + li;
    const bool valid = si < A.n_owned;
    const float4 pi_raw = A.pos[valid ? si : (int64_t)b * A.BI];
    float4 pi = pi_raw;
    local_xyz_t<false>(pi.x, pi.y, pi.z, ctr, G);
    const int rows = __builtin_amdgcn_readfirstlane(A.wave_rows[(b * A.JS + js) * (A.BI >> 6) + (li >> 6)]);
    const uint2* my_rows = A.nbr + (((int64_t)b * A.JS + js) * A.R_cap) * A.BI + li;
    const int32_t* tix = A.tile_idx + (int64_t)b * A.T_cap;
    uint2* out_rows = A.nbr_dst + (((int64_t)b * A.JS + js) * A.R_cap) * A.BI + li;
    // the first rows are requested before anything else: their latency passes behind the staging
    const int last = max(rows - 1, 0);
    uint2 e0 = my_rows[0], e1 = my_rows[(int64_t)min(1, last) * A.BI];
    auto quant = [](float v) -> int { return (int)fminf(fmaxf(rintf(v * PRUNE_Q), -12000.f), 12000.f); };
    // eight bounding boxes of the i-atoms (runs of BI/8 atoms), as in the PRUNE pass of k_forces
    {
        const int lpb = A.BI >> 3;
        float mn[3] = {pi.x, pi.y, pi.z}, mx[3] = {pi.x, pi.y, pi.z};
        for (int o = lpb >> 1; o > 0; o >>= 1)
#pragma unroll
            for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], o, WAVE)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o, WAVE)); }
        if (js == 0 && (li & (lpb - 1)) == 0) {
            reinterpret_cast<float4*>(l_box)[(li / lpb) * 2] = make_float4(mn[0], mn[1], mn[2], 0.f);
            reinterpret_cast<float4*>(l_box)[(li / lpb) * 2 + 1] = make_float4(mx[0], mx[1], mx[2], 0.f);
        }
        if (A.snap_dst && js == 0 && valid) A.snap_dst[si] = pi_raw;
    }
    __syncthreads();
    // stage the tile: quantised coordinates into l_q, the verdict "can an inner-list entry of this block name this atom" into l_new
    // (the walk below keeps a pair when its QUANTISED distance is within the radius + 2 units, i.e. its true distance within + 2 + √3
    // units: the atoms that can be named must cover that)
    const float reach = sqrtf(A.r_prune2) + 4.f / PRUNE_Q, reach2 = reach * reach * 1.0001f;
    for (int t0 = 0; t0 < tile_n; t0 += 4 * nthr) {
        int s[4]; float4 p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = tix[min(t0 + k * nthr + tid, tile_n - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = A.pos[s[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = t0 + k * nthr + tid;
            if (t < tile_n) {
                float x = p[k].x, y = p[k].y, z = p[k].z;
                local_xyz_t<false>(x, y, z, ctr, G);
                float best = 3.0e38f;
#if MHIP_LEXP == 2     // timing experiment: no box tests
                best = 0.f;
#else
#pragma unroll 1
                for (int w = 0; w < 8; ++w) {
                    const float4 lo = reinterpret_cast<const float4*>(l_box)[2 * w], hi = reinterpret_cast<const float4*>(l_box)[2 * w + 1];
                    const float ex = fmaxf(fmaxf(lo.x - x, x - hi.x), 0.f), ey = fmaxf(fmaxf(lo.y - y, y - hi.y), 0.f), ez = fmaxf(fmaxf(lo.z - z, z - hi.z), 0.f);
                    best = fminf(best, ex * ex + ey * ey + ez * ez);
                }
#endif
                l_new[t] = best <= reach2 ? 1 : 0;
                l_q[t] = make_uint2(((uint32_t)quant(x) & 0xffffu) | ((uint32_t)quant(y) << 16), (uint32_t)quant(z) & 0xffffu);
            }
        }
    }
    if (tid == 0) l_q[tile_n] = make_uint2(20000u, 0u);          // sentinel slot: x = 20000 units, at least 8000 (3.9 nm) from every clamped coordinate
    __syncthreads();
    // ordered numbering of the atoms that stay (the compacted tile keeps the cell-major order)
    int n_new = 0;
    {
        const int per = (tile_n + nthr - 1) / nthr, t0 = min(tid * per, tile_n), t1 = min(t0 + per, tile_n);
        int cnt = 0;
        for (int t = t0; t < t1; ++t) cnt += l_new[t];
        l_scan[tid] = cnt;
        __syncthreads();
        if (tid < WAVE) {
            int run = 0;
            for (int base = 0; base < nthr; base += WAVE) {
                int v = (base + tid < nthr) ? l_scan[base + tid] : 0, x = v;
#pragma unroll
                for (int o = 1; o < WAVE; o <<= 1) { int u = __shfl_up(x, o, WAVE); if (tid >= o) x += u; }
                if (base + tid < nthr) l_scan[base + tid] = run + x - v;
                run += __shfl(x, WAVE - 1, WAVE);
            }
            if (tid == 0) l_scan[nthr] = run;
        }
        __syncthreads();
        int run = l_scan[tid];
        for (int t = t0; t < t1; ++t) {      // the new entry value (slot · 4) rides in the spare half of the atom's second LDS word: no lookup of its own in the walk
            if (l_new[t]) { l_q[t].y |= (uint32_t)(run << ESHIFT_SCALED) << 16; A.tile_idx_dst[(int64_t)b * A.T_cap + run] = tix[t]; ++run; }
        }
        n_new = l_scan[nthr];
        __syncthreads();
    }
    // walk the outer rows: integer distances, kept entries re-emitted under their new numbers
    const s16x2 pxy = {(short)quant(pi.x), (short)quant(pi.y)}, pz0 = {(short)quant(pi.z), 0};
    const float rq = sqrtf(A.r_prune2) * PRUNE_Q + 2.f;                            // two units of slack: both ends of a pair were rounded
    const int rp2q = (int)(rq * rq) + 1;
    // the last eight entries kept, newest in the top half of w3; appended without branches, a row of four stored at most once per walked row
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0; int kept = 0;
    auto push = [&](uint32_t e, bool k) {
        const uint32_t n0 = __builtin_amdgcn_alignbit(w1, w0, 16), n1 = __builtin_amdgcn_alignbit(w2, w1, 16), n2 = __builtin_amdgcn_alignbit(w3, w2, 16), n3 = __builtin_amdgcn_alignbit(e, w3, 16);
        w0 = k ? n0 : w0; w1 = k ? n1 : w1; w2 = k ? n2 : w2; w3 = k ? n3 : w3;
        kept += k ? 1 : 0;
    };
    auto row_out = [&](int before) {
        if ((kept >> 2) != (before >> 2)) {
            const int r = kept & 3;
            const bool up = r < 2, odd = (r & 1) != 0;
            const uint32_t b0 = up ? w1 : w0, b1 = up ? w2 : w1, b2 = up ? w3 : w2;
            const uint32_t lo = odd ? __builtin_amdgcn_alignbit(b1, b0, 16) : b1, hi = odd ? __builtin_amdgcn_alignbit(b2, b1, 16) : b2;
            out_rows[(int64_t)((kept >> 2) - 1) * A.BI] = make_uint2(lo, hi);
        }
    };
    auto emit = [&](uint32_t e) { const int before = kept; push(e, true); row_out(before); };
    typedef __attribute__((address_space(3))) const unsigned char* lds_bptr;
    const lds_bptr qbase = (lds_bptr)(uintptr_t)0;                                 // (l_q starts at LDS address 0: no static __shared__ here)
    auto row = [&](const uint2 e4) {
        const uint32_t oa = e4.x & 0xffffu, ob = e4.x >> 16, oc = e4.y & 0xffffu, od = e4.y >> 16;     // byte offsets slot·4
        typedef __attribute__((address_space(3))) const uint64_t* lds_q;
        const uint64_t qa = *(lds_q)(qbase + 2 * oa), qb = *(lds_q)(qbase + 2 * ob), qc = *(lds_q)(qbase + 2 * oc), qd = *(lds_q)(qbase + 2 * od);
        auto r2 = [&](const uint64_t q) -> int {
            const s16x2 dxy = __builtin_bit_cast(s16x2, (uint32_t)q) - pxy, dz = __builtin_bit_cast(s16x2, (uint32_t)(q >> 32) & 0xffffu) - pz0;
            return __builtin_amdgcn_sdot2(dxy, dxy, __builtin_amdgcn_sdot2(dz, dz, 0, false), false);
        };
        const int ra = r2(qa), rb = r2(qb), rc = r2(qc), rd = r2(qd);
#if MHIP_LEXP == 1     // timing experiment: no emission
        if (((ra <= rp2q) + (rb <= rp2q) + (rc <= rp2q) + (rd <= rp2q)) == 77) emit((uint32_t)(qa >> 48));
#else
        const int before = kept;
        push((uint32_t)(qa >> 48), ra <= rp2q); push((uint32_t)(qb >> 48), rb <= rp2q); push((uint32_t)(qc >> 48), rc <= rp2q); push((uint32_t)(qd >> 48), rd <= rp2q);
        row_out(before);
#endif
    };
    if (rows > 0 && valid) {
        int r = 0;
        for (; r + 2 <= rows; r += 2) {
            row(e0); e0 = my_rows[(int64_t)min(r + 2, last) * A.BI];
            row(e1); e1 = my_rows[(int64_t)min(r + 3, last) * A.BI];
        }
        if (r < rows) row(e0);
    }
    // pad to the wave's row count, counts, displacement since the outer search
    const uint32_t SENTP = (uint32_t)n_new << ESHIFT_SCALED;
    const int rows_wave = wave_max((kept + 3) >> 2);
    while (((kept + 3) >> 2) < rows_wave || (kept & 3)) emit(SENTP);
    if ((tid & 63) == 0) A.rows_dst[(b * A.JS + js) * (A.BI >> 6) + (li >> 6)] = rows_wave;
    if (tid == 0) A.tile_cnt_dst[b] = n_new;
    float d2 = 0.f;
    if (valid && js == 0) {
        const float4 q = A.pos_snap[si];
        float ex = pi_raw.x - q.x, ey = pi_raw.y - q.y, ez = pi_raw.z - q.z;
        disp_image(ex, ey, ez, G);
        d2 = ex * ex + ey * ey + ez * ez;
    }
    d2 = wave_max(d2);
    if (js == 0 && (tid & 63) == 0) A.blk_disp2[b * (A.BI >> 6) + (li >> 6)] = d2;
}

}  // namespace mhip
