// forces_gs.hip — the pair pass of SMALL systems with per-atom parameters (the solvated-protein class: fp32, Lennard-Jones with the
// distance cutoff + CoulombEwald direct space with the A&S erfc, or + CoulombReactionField), cut into workgroups a quarter the size.
//
// Why.  k_forces runs one 1024-lane workgroup per block of 64 atoms; 6mrr has 250 blocks, the GPU 256 compute units, so a pass is ONE
// round of workgroups and lasts as long as its slowest one.  The per-wave time stamps of profiles/r04_force_ab.txt §5 show what that costs:
// an atom of the protein interior has 645 neighbours inside the inner radius, one in bulk water 380 (mean 447), a block is a compact
// piece of space, so every wave of a dense block walks 13 rows where the average wave walks 8.7 — row walk 20 µs for the slowest wave,
// 9.9 µs on average, 5.9 µs at the 10th percentile — and 250 compute units wait for a handful.  Nothing inside one workgroup can fix
// that; the work has to be cut finer than the compute units so that each of them gets a MIX of it.
//
// How.  A block's J-SPLIT waves are dealt to GS = 4 workgroups ("groups") of JS/GS waves, and the tile with them: group g owns the
// tile slots n ≡ g (mod GS) — a uniform sample of the block's neighbourhood, a quarter of the staging work and of the LDS — and every
// atom's entries that name such a slot, dealt evenly over the group's waves (k_regroup, once per prune: entries become group-local
// indices n / GS, so the walk needs no translation).  A compute unit holds four such workgroups; workgroup w goes to group w / n_blocks
// of block (w mod n_blocks + group · n_blocks/4) — the hardware fills the compute units round-robin, so the four that share one come
// from four blocks a quarter of the Hilbert curve apart.  Each group reduces its waves through LDS in fixed order and leaves ITS partial
// force: group 0 in the force array, the others in side arrays that the per-atom sums of the bonded slots add (bonded.h: the one launch
// that touches every atom next) — no atomics, the sum of an atom's four partials is taken in fixed order, forces stay bit-reproducible.
// Same arithmetic per pair as k_forces (pair_eval2 / pair_eval of physics.h); same lists (the inner list of the dual scheme, re-dealt).
#include "kernels.h"
#include "forces_launch.h"
#include "step_fused.h"

namespace mhip {

// ---- k_regroup: the inner list the prune wrote → the group-split list ------------------------------------------------------------------
// src [b][js][r][lane] rows of four 16-bit entries (slot | special << 15), cnt [b][js][lane] real entries per sub-list and lane.
// dst: same geometry; an atom's entries with slot ≡ g (mod GS), numbered k = 0, 1, … in the order of the source sub-lists, are DEALT to the
// JSW waves of group g: entry k goes to sub-list g·JSW + (k mod JSW), position k / JSW, as slot / GS | special << 15; each sub-list is
// padded with the group-local sentinel index qmax = ⌈tile_n / GS⌉ to the row count of its wave (rows_dst).
// The destination has a row capacity of its own, R_cap_dst = GS · R_cap: a group's share of an atom's entries is only statistically a GS-th of
// them, but it can never exceed ALL of them (JS sub-lists × 4 · R_cap entries, dealt over JS / GS waves), so no share overflows its sub-list.
// One pass to count, one to scatter — every lane fetches its rows eight at a time (a version that walked the source rows one dependent
// load after the other took 70 µs per prune on 6mrr, more than the prune itself).  The running numbers of the four groups travel as four
// 16-bit fields of one 64-bit word: the prefix over the source sub-lists is a packed sum, and no register array is indexed dynamically.
__global__ void __launch_bounds__(1024) k_regroup(RegroupArgs A) {
    extern __shared__ __align__(16) unsigned char rg_smem[];
    unsigned long long* l_c = reinterpret_cast<unsigned long long*>(rg_smem);     // [JS][BI]: entries of (source sub-list, atom) per group, 16 bits each
    const int b = blockIdx.x, tid = threadIdx.x, li = tid & (A.BI - 1), js = tid >> A.BI_shift;
    const int GS = A.GS, JSW = A.JS / GS, lgW = 31 - __builtin_clz(JSW);
    const uint32_t gmask = (uint32_t)GS - 1u;
    const int64_t sub0 = (int64_t)b * A.JS;
    const int n = (int)A.cnt[(sub0 + js) * A.BI + li], nrow = (n + 3) >> 2;
    const uint2* rows = A.src + ((sub0 + js) * A.R_cap) * A.BI + li;
    constexpr int NB = 8;
    // SPECIAL entries (1-4 pairs: bit 15) are counted apart and dealt BEHIND an atom's other entries: the pass takes its two-partner packed loop only for rows
    // without a special entry in any lane, and a wave of protein atoms had one in 60 % of its rows (3 094 special pairs, all of them in the ≈ 40 protein blocks
    // — the blocks a launch waits for); at the tail of the lists they sit in the last row or two.
    unsigned long long* l_cs = l_c + (size_t)A.JS * A.BI;                          // [JS][BI]: the special ones among them
    unsigned long long C = 0, CS = 0;
#if MHIP_STAMPS
    if (A.dbg && tid == 0) A.dbg[(size_t)b * 8 + 0] = wall_clock64();
#endif

    for (int r0 = 0; r0 < nrow; r0 += NB) {
        uint2 rw[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) rw[j] = rows[(int64_t)min(r0 + j, nrow - 1) * A.BI];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const uint32_t e[4] = {rw[j].x & 0xffffu, rw[j].x >> 16, rw[j].y & 0xffffu, rw[j].y >> 16};
#pragma unroll
            for (int t = 0; t < 4; ++t) if (4 * (r0 + j) + t < n) { const unsigned long long one = 1ull << ((e[t] & gmask) * 16); if (e[t] & 0x8000u) CS += one; else C += one; }
        }
    }
    l_c[js * A.BI + li] = C; l_cs[js * A.BI + li] = CS;
#if MHIP_STAMPS
    if (A.dbg && tid == 0) A.dbg[(size_t)b * 8 + 1] = wall_clock64();
#endif

    __syncthreads();
    unsigned long long K = 0, TOT = 0, KS = 0, TOTS = 0;                            // numbers of my first (special) entries per group; the atom's totals
    for (int q = 0; q < A.JS; ++q) {
        const unsigned long long v = l_c[q * A.BI + li], vs = l_cs[q * A.BI + li];
        TOT += v; TOTS += vs; if (q < js) { K += v; KS += vs; }
    }
    KS += TOT;                                                                      // (the special entries are numbered behind all the others of the atom and group)
    // as a DESTINATION lane (group gd, wave wd of the group): how many entries come my way, the row count of my wave
    const int gd = js >> lgW, wd = js & (JSW - 1);
    const int tot_d = (int)(((TOT + TOTS) >> (gd * 16)) & 0xffffull);
    const int n_mine = max(tot_d - wd + JSW - 1, 0) >> lgW;
    int rows_w = (n_mine + 3) >> 2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rows_w = max(rows_w, __shfl_xor(rows_w, o, WAVE));
    if ((tid & (WAVE - 1)) == 0) A.rows_dst[(sub0 + js) * (A.BI >> 6) + (li >> 6)] = rows_w;
    // The scatter goes through LDS when the block's new list fits behind the counters (rows of the longest wave × sub-lists × 512 bytes:
    // ≈ 100 KB for a 6mrr block) and leaves as whole 8-byte rows, a wave's 512 bytes at a time; 2-byte stores scattered straight into
    // global memory cost 45 µs per prune where the prune itself takes 61.
    int* l_rmax = reinterpret_cast<int*>(l_c + 2 * (size_t)A.JS * A.BI);
    if (tid == 0) *l_rmax = 0;
    __syncthreads();
    if ((tid & (WAVE - 1)) == 0) atomicMax(l_rmax, rows_w);
    __syncthreads();
    const int R_l = *l_rmax;
#if MHIP_STAMPS
    if (A.dbg && tid == 0) A.dbg[(size_t)b * 8 + 2] = wall_clock64();
#endif

    uint16_t* l_dst = reinterpret_cast<uint16_t*>(l_rmax + 4);
    const bool via_lds = (size_t)A.JS * R_l * A.BI * 8 <= (size_t)A.lds_list_bytes;
    uint16_t* dst16 = via_lds ? l_dst : reinterpret_cast<uint16_t*>(A.dst);
    auto at = [&](int sub, int p) -> int64_t {
        return via_lds ? (int64_t)(((((sub * R_l) + (p >> 2)) * A.BI + li) << 2) + (p & 3)) : ((((sub0 + sub) * A.R_cap_dst + (p >> 2)) * A.BI + li) << 2) + (p & 3);
    };
    // scatter my entries
    for (int r0 = 0; r0 < nrow; r0 += NB) {
        uint2 rw[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) rw[j] = rows[(int64_t)min(r0 + j, nrow - 1) * A.BI];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const uint32_t e[4] = {rw[j].x & 0xffffu, rw[j].x >> 16, rw[j].y & 0xffffu, rw[j].y >> 16};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (4 * (r0 + j) + t < n) {
                    const uint32_t g = e[t] & gmask;
                    const bool spc = (e[t] & 0x8000u) != 0u;
                    const int k = (int)(((spc ? KS : K) >> (g * 16)) & 0xffffull);
                    if (spc) KS += 1ull << (g * 16); else K += 1ull << (g * 16);
                    dst16[at((int)g * JSW + (k & (JSW - 1)), k >> lgW)] = (uint16_t)(((e[t] & 0x7fffu) >> A.lgGS) | (e[t] & 0x8000u));
                }
            }
        }
    }
    // pad my destination sub-list (positions nobody else writes)
    const uint16_t SENT = (uint16_t)((A.tile_cnt[b] + GS - 1) >> A.lgGS);
    for (int p = n_mine; p < 4 * rows_w; ++p) dst16[at(js, p)] = SENT;
#if MHIP_STAMPS
    if (A.dbg && tid == 0) A.dbg[(size_t)b * 8 + 3] = wall_clock64();
#endif

    if (via_lds) {
        __syncthreads();
        const uint2* l_rows = reinterpret_cast<const uint2*>(l_dst);
        uint2* out = A.dst + ((sub0 + js) * A.R_cap_dst) * A.BI + li;
        for (int r = 0; r < rows_w; ++r) out[(int64_t)r * A.BI] = l_rows[(js * R_l + r) * A.BI + li];
    }
#if MHIP_STAMPS
    if (A.dbg && tid == 0) A.dbg[(size_t)b * 8 + 4] = wall_clock64();
#endif
}

// ---- k_gs_balance: which (block, group) a workgroup of the pass takes ---------------------------------------------------------------------------
// The time stamps of the pass (tools/gs_times.py) show the hardware placing workgroup w on compute unit w mod 256, every workgroup resident from the start, and
// a compute unit finishing when the rows of its four workgroups are walked (correlation 0.93) — 116 … 192 wave-rows per unit around a mean of 166.  So the
// (block, group) items are ranked by their rows (counting rank of a thousand items: sixteen lanes per item) and handed out in a serpentine: the `period` heaviest to
// workgroups 0 … period − 1, the next `period` in the opposite direction, and so on, which pairs heavy with light on every compute unit.
__global__ void __launch_bounds__(1024) k_gs_balance(const int32_t* __restrict__ rows_gs, int n_blocks, int JS, int GS, int wps, int period, uint16_t* item_of) {
    extern __shared__ int32_t l_work[];
    const int n_items = n_blocks * GS, JSW = JS / GS;
    for (int t = threadIdx.x; t < n_items; t += blockDim.x) {      // (every workgroup: the rows of all items)
        const int g = t / n_blocks, b = t - g * n_blocks;
        int w = 0;
        for (int q = 0; q < JSW * wps; ++q) w += rows_gs[((int64_t)b * JS + g * JSW) * wps + q];
        l_work[t] = w;
    }
    __syncthreads();
    // 64 items per workgroup, sixteen lanes per item: each counts its sixteenth of the items that come before (more rows; equal rows and a smaller number)
    const int t = (int)blockIdx.x * 64 + (int)(threadIdx.x >> 4), sub = (int)(threadIdx.x & 15);
    const int tc = min(t, n_items - 1), mine = l_work[tc];
    int rank = 0;
    for (int u = sub; u < n_items; u += 16) { const int o = l_work[u]; rank += (o > mine || (o == mine && u < tc)) ? 1 : 0; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) rank += __shfl_xor(rank, o, 64);
    if (t < n_items && sub == 0) {
        const int layer = rank / period, pos = rank - layer * period;
        const int in_layer = min(period, n_items - layer * period);
        item_of[layer * period + ((layer & 1) ? in_layer - 1 - pos : pos)] = (uint16_t)t;
    }
}

// ---- k_forces_gs: one group of one block ----------------------------------------------------------------------------------------------------
template <int COULM, bool MINIMG>
__device__ inline void forces_gs_body(const GsArgs& A, int wg, unsigned char* smem) {
    const GridP<float>& G = A.G;
    const int item = A.item_of ? (int)A.item_of[wg] : wg;
    const int g = item / A.n_blocks, bq = item - g * A.n_blocks;
    const int b = A.item_of ? bq : (bq + g * A.spread) % A.n_blocks;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int li = tid & (A.BI - 1), jw = tid >> A.BI_shift, JSW = A.JS >> A.lgGS, js = g * JSW + jw;
    // timing experiment (-DMHIP_STAMPS=1): per wave — [0] shader clock at entry, [1] rows walked, [2] block | group << 20 | HW_ID << 32, [3] XCC_ID,
    // [4..7] the 100 MHz wall clock at entry, behind the staging barrier, behind the row walk, at the end
    [[maybe_unused]] auto stamp = [&](int k, unsigned long long v) {
#if MHIP_STAMPS
        if (A.dbg && (tid & 63) == 0) A.dbg[((size_t)wg * (nthr >> 6) + (tid >> 6)) * 8 + k] = v;
#endif
    };
    stamp(0, __builtin_readcyclecounter()); stamp(4, wall_clock64());
    const int tile_n = A.tile_cnt[b], qmax = (tile_n + A.GS - 1) >> A.lgGS;        // group-local slots 0 .. qmax − 1, sentinel qmax
    float4* l_pos = reinterpret_cast<float4*>(smem);
    float2* l_lj = reinterpret_cast<float2*>(l_pos + (A.Q_lds + 1));
    const float4 ctr = A.blk_center[b];
    const int64_t si = (int64_t)b * A.BI + li;
    const bool valid = si < A.n_owned;
    float4 pi = A.pos[valid ? si : (int64_t)b * A.BI];
    if constexpr (!MINIMG) local_xyz_t<false>(pi.x, pi.y, pi.z, ctr, G);
    auto pre_e = [](float2 v) { v.y = v.x == 0.f ? 0.f : __builtin_amdgcn_sqrtf(v.y); v.x *= 0.5f; return v; };   // (σ/2, √ϵ — 0 where σ = 0: physics.h PRE_E)
    const float2 lji = pre_e(A.lj[valid ? si : (int64_t)b * A.BI]);
    const int rows = __builtin_amdgcn_readfirstlane(A.wave_rows[((int64_t)b * A.JS + js) * (A.BI >> 6) + (li >> 6)]);
    const uint2* my_rows = A.nbr + (((int64_t)b * A.JS + js) * A.R_cap) * A.BI + li;
    const int32_t* tix = A.tile_idx + (int64_t)b * A.T_cap;
    // stage my quarter of the tile: slots g, g + GS, …; four atoms per lane and round, their dependent fetches issued together
    for (int u0 = 0; u0 <= qmax; u0 += 4 * nthr) {
        int s[4]; float4 p[4]; float2 q[4]; bool real[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int u = u0 + k * nthr + tid, n = (u << A.lgGS) + g; real[k] = u < qmax && n < tile_n; s[k] = real[k] ? tix[n] : 0; }
#pragma unroll
        for (int k = 0; k < 4; ++k) if (real[k]) { p[k] = A.pos[s[k]]; q[k] = A.lj[s[k]]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = u0 + k * nthr + tid;
            if (u > qmax) continue;
            if (real[k]) {
                if constexpr (!MINIMG) local_xyz_t<false>(p[k].x, p[k].y, p[k].z, ctr, G);
                l_pos[u] = p[k]; l_lj[u] = pre_e(q[k]);
            } else { l_pos[u] = make_float4(1e4f, 1e4f, 1e4f, 0.f); l_lj[u] = make_float2(0.f, 0.f); }   // far away (beyond every cutoff), no charge, no LJ
        }
    }
    __syncthreads();
    stamp(5, wall_clock64()); stamp(1, (unsigned long long)rows);
    stamp(2, (unsigned long long)b | ((unsigned long long)g << 20) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32)); stamp(3, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20));
    float fx = 0.f, fy = 0.f, fz = 0.f;
    const Pk2Consts K(A.I);
    const bool pk2_ok = Pk2Consts::usable(A.I);
    const float kqi = A.I.ke * pi.w, ei24 = 24.f * lji.y;
    auto disp = [&](const float4& pj, float& dx, float& dy, float& dz) {
        if constexpr (MINIMG) min_image_exact<float>(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z, G, dx, dy, dz);
        else { dx = pj.x - pi.x; dy = pj.y - pi.y; dz = pj.z - pi.z; }
    };
    uint2 e_next = rows > 0 ? my_rows[0] : make_uint2(0, 0);
    for (int r = 0; r < rows; ++r) {
        const uint2 e4 = e_next;
        if (r + 1 < rows) e_next = my_rows[(int64_t)(r + 1) * A.BI];
        if (pk2_ok && __builtin_amdgcn_ballot_w64(((e4.x | e4.y) & 0x80008000u) != 0u) == 0ull) {
            v2f fxy = {0.f, 0.f}; float fzs = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t ew = h ? e4.y : e4.x;
                const uint32_t sa = ew & 0x7fffu, sb = (ew >> 16) & 0x7fffu;
                const float4 pa = l_pos[sa], pb = l_pos[sb];
                const float2 la = l_lj[sa], lb = l_lj[sb];
                v2f da, db; float dza, dzb;
                if constexpr (MINIMG) { float x, y; disp(pa, x, y, dza); da = (v2f){x, y}; disp(pb, x, y, dzb); db = (v2f){x, y}; }
                else { const v2f pixy = {pi.x, pi.y}; da = (v2f){pa.x, pa.y} - pixy; db = (v2f){pb.x, pb.y} - pixy; dza = pa.z - pi.z; dzb = pb.z - pi.z; }
                const v2f qa = da * da, qb = db * db;
                const v2f r2 = {__builtin_fmaf(dza, dza, qa.x + qa.y), __builtin_fmaf(dzb, dzb, qb.x + qb.y)};
                const v2f fr = pair_eval2<COULM>(A.I, K, r2, kqi, (v2f){pa.w, pb.w}, lji.x, (v2f){la.x, lb.x}, ei24, (v2f){la.y, lb.y});
                fxy += da * (v2f){fr.x, fr.x}; fxy += db * (v2f){fr.y, fr.y};
                fzs = __builtin_fmaf(dza, fr.x, __builtin_fmaf(dzb, fr.y, fzs));
            }
            fx -= fxy.x; fy -= fxy.y; fz -= fzs;            // force on i is −f (force.jl:873)
            continue;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = ((k < 2 ? e4.x : e4.y) >> (16 * (k & 1))) & 0xffffu;
            const uint32_t slot = e & 0x7fffu;
            const float4 pj = l_pos[slot]; const float2 ljj = l_lj[slot];
            float dx, dy, dz; disp(pj, dx, dy, dz);
            const float r2 = dx * dx + dy * dy + dz * dz;
            float pe = 0.f;
            const float fr = pair_eval<float, LJ_DIST, COULM, false, true>(A.I, r2, pi.w, pj.w, lji.x, ljj.x, lji.y, ljj.y, (e >> 15) != 0u, pe);
            fx -= fr * dx; fy -= fr * dy; fz -= fr * dz;
        }
    }
    stamp(6, wall_clock64());
    // the group's waves through LDS, fixed order; the group's partial force of the block's atoms
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    red[(jw * 3 + 0) * A.BI + li] = fx; red[(jw * 3 + 1) * A.BI + li] = fy; red[(jw * 3 + 2) * A.BI + li] = fz;
    __syncthreads();
    if (jw == 0 && valid) {
        for (int q = 1; q < JSW; ++q) { fx += red[(q * 3 + 0) * A.BI + li]; fy += red[(q * 3 + 1) * A.BI + li]; fz += red[(q * 3 + 2) * A.BI + li]; }
        float4* dst = g == 0 ? A.frc : A.parts + (int64_t)(g - 1) * A.part_stride;
        dst[si] = make_float4(fx, fy, fz, 0.f);
    }
    stamp(7, wall_clock64());
}

template <int COULM, bool MINIMG>
__global__ void __launch_bounds__(256, 4) k_forces_gs(GsArgs A) {     // (four workgroups per compute unit: at most 128 registers)
    extern __shared__ __align__(32) unsigned char smem[];
    forces_gs_body<COULM, MINIMG>(A, (int)blockIdx.x, smem);
}

// The pair pass, the charge spreading and the bonded terms of one MD step in ONE launch (cf. step_fused.h, which pairs the last two): none of
// the three depends on another, all are 256-lane workgroups, and the pair workgroups — first in the grid, the longest — leave the
// arithmetic units idle half of the time (latency of the row walk) while the spreading is bound by LDS atomics and latency.  The short jobs
// sit at the HEAD of the grid: every pair group is resident from the start anyway (3.9 per compute unit) and spends its first 4.8 µs staging its
// tile quarter, so the spreading and the terms run through that phase instead of queueing behind the pair groups for a free slot.
// Every role carves its LDS from the launch's dynamic pool (the spreading: its tables and whatever is left as the sub-mesh).
template <int COULM, bool MINIMG, int ORDER>
__global__ void __launch_bounds__(256, 4) k_pair_spread_bonded(GsArgs A, int n_pair, int64_t n_atoms, float* rgrid, PmeP<float> P, int n_spread, int lds_bytes, BondedArgs<float> B,
                                                               int n_term_wg, const double* cm_fin_in, int cm_fin_n, double* cm_fin_out) {
    extern __shared__ __align__(32) unsigned char smem[];
    int wg = (int)blockIdx.x;
    {      // the short jobs at the head of the grid: they start with the pair groups and run through those groups' staging phase (26.4 -> 25.8 us per launch, profiles/r04_force_ab.txt §16)
        const int n_short = (int)gridDim.x - n_pair;
        wg = wg < n_short ? n_pair + wg : wg - n_short;
    }
    if (wg < n_pair) { forces_gs_body<COULM, MINIMG>(A, wg, smem); return; }
    if (wg < n_pair + n_spread) { pme_spread_blocks<float, ORDER, 64, true>(wg - n_pair, n_spread, n_atoms, A.pos, rgrid, P, smem, lds_bytes); return; }
    // one more workgroup (cm_fin_in != nullptr): the Σ m v partials of the integrator launch before this step become ONE partial here (cm_finalize_in_block: the
    // summation of block_vcm, by a 256-lane workgroup as there), so that the step's last launch — which integrates, step_fused.h — reads four words instead of re-summing
    if (wg >= n_pair + n_spread + n_term_wg) { if (cm_fin_in) cm_finalize_in_block(cm_fin_in, cm_fin_n, cm_fin_out, smem); return; }
    double e = 0;
    bonded_terms<float, false, true>(B, (wg - n_pair - n_spread) * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), 64, e);   // four 64-lane term blocks per workgroup
}

size_t gs_lds_bytes(int q_lds, int BI, int JSW) { return std::max((size_t)(q_lds + 1) * (sizeof(float4) + sizeof(float2)), (size_t)JSW * 3 * BI * sizeof(float)) + 64; }

void launch_regroup(const RegroupArgs& A, int n_blocks, hipStream_t stream) {
    const size_t lds = 2 * (size_t)A.JS * A.BI * sizeof(unsigned long long) + 16 + (size_t)A.lds_list_bytes;
    if (lds > 64 * 1024) MHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_regroup), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      // (per launch, like every other launcher: a capacity retry or a second context may need more than the first call did)
    hipLaunchKernelGGL(k_regroup, dim3(n_blocks), dim3(A.BI * A.JS), lds, stream, A);
}

void launch_gs_balance(const int32_t* rows_gs, int n_blocks, int JS, int GS, int waves_per_sub, int period, uint16_t* item_of, hipStream_t stream) {
    // (items are uint16 and staged as n_blocks·GS ints of dynamic LDS: Engine::gs_groups() keeps n_blocks·GS <= 16 384)
    hipLaunchKernelGGL(k_gs_balance, dim3((unsigned)((n_blocks * GS + 63) / 64)), dim3(1024), (size_t)n_blocks * GS * sizeof(int32_t), stream, rows_gs, n_blocks, JS, GS, waves_per_sub, period, item_of);
}

void launch_forces_gs(const GsArgs& A, int coulm, bool minimg, hipStream_t stream) {
    const size_t lds = gs_lds_bytes(A.Q_lds, A.BI, A.JS >> A.lgGS);
    const dim3 grid((unsigned)(A.n_blocks * A.GS)), block((unsigned)(A.BI * (A.JS >> A.lgGS)));
    auto go = [&](auto kern) {
        if (lds > 64 * 1024) MHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, block, lds, stream, A);
    };
    if (coulm == MHIP_COUL_REACTION_FIELD) { if (minimg) go(k_forces_gs<MHIP_COUL_REACTION_FIELD, true>); else go(k_forces_gs<MHIP_COUL_REACTION_FIELD, false>); }
    else { if (minimg) go(k_forces_gs<MHIP_COUL_EWALD_DIRECT, true>); else go(k_forces_gs<MHIP_COUL_EWALD_DIRECT, false>); }
}

// the fused form: pair groups + spreading + bonded terms; lds_bytes >= the pair groups' need and the spreading's head + a useful sub-mesh
void launch_pair_spread_bonded(const GsArgs& A, int n_pair, int coulm, bool minimg, int order, int64_t n_atoms, float* rgrid, const PmeP<float>& P, int n_spread, const BondedArgs<float>& B, int n_term_wg,
                               size_t lds_bytes, hipStream_t stream, const double* cm_fin_in, int cm_fin_n, double* cm_fin_out) {
    const dim3 grid((unsigned)(n_pair + n_spread + n_term_wg + (cm_fin_in ? 1 : 0))), block(256);
    auto go = [&](auto kern) {
        if (lds_bytes > 64 * 1024) MHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, A, n_pair, n_atoms, rgrid, P, n_spread, (int)lds_bytes, B, n_term_wg, cm_fin_in, cm_fin_n, cm_fin_out);
    };
    auto by_order = [&](auto coul_tag, auto mi_tag) {
        constexpr int C = decltype(coul_tag)::value; constexpr bool M = decltype(mi_tag)::value;
        if (order == 4) go(k_pair_spread_bonded<C, M, 4>); else if (order == 5) go(k_pair_spread_bonded<C, M, 5>); else go(k_pair_spread_bonded<C, M, 6>);
    };
    using RF = std::integral_constant<int, MHIP_COUL_REACTION_FIELD>; using EW = std::integral_constant<int, MHIP_COUL_EWALD_DIRECT>;
    if (coulm == MHIP_COUL_REACTION_FIELD) { if (minimg) by_order(RF{}, std::true_type{}); else by_order(RF{}, std::false_type{}); }
    else { if (minimg) by_order(EW{}, std::true_type{}); else by_order(EW{}, std::false_type{}); }
}
size_t spread_head_bytes_f32(int order) { return order == 4 ? pme_spread_head_bytes<float, 4, 64>() : order == 5 ? pme_spread_head_bytes<float, 5, 64>() : pme_spread_head_bytes<float, 6, 64>(); }

}  // namespace mhip
