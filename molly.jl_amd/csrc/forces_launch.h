// forces_launch.h — host-side entry points of the pair-kernel translation units (forces_inst.hip × 8, forces_uniform.hip)
#pragma once
#include "kernels.h"

namespace mhip {

// every k_forces<T, ·, COULM, ·, ·, ·, ·> variant, one explicit specialisation per translation unit
template <class T, int COULM>
void launch_forces_tc(const ForceArgs<T>& A, int ljm, bool energy, bool minimg, bool seg, bool prune, size_t lds, unsigned threads, hipStream_t stream);

// the fp32 one-type LJ variants (forces only, block-local coordinates)
void launch_forces_uniform_f32(const ForceArgs<float>& A, bool seg, bool prune, size_t lds, unsigned threads, hipStream_t stream, bool step = false, bool halo = false, bool lang = false);

// ---- the group-split pair pass of small systems (forces_gs.hip) ----
struct RegroupArgs {
    int BI, BI_shift, JS, GS, lgGS, R_cap;
    const uint2* src; const uint16_t* cnt; const int32_t* tile_cnt;      // the inner list as the prune wrote it, entries per (sub-list, lane), compacted tile sizes
    uint2* dst; int32_t* rows_dst;                                        // the group-split list
    int lds_list_bytes;                                                    // LDS the launch sets aside for staging a block's new list (the scatter goes to global memory if it does not fit)
    int R_cap_dst;                                                         // row capacity of a destination sub-list: GS · R_cap (a group's share of an atom's entries cannot exceed all of them)
    unsigned long long* dbg;                                              // builds with -DMHIP_STAMPS=1: [block][8] wall-clock stamps of wave 0 (entry, counted, dealt, scattered, end)
};
struct GsArgs {
    GridP<float> G; InterP<float> I;
    int64_t n_owned; int BI, BI_shift, JS, GS, lgGS, R_cap, T_cap, Q_lds, n_blocks, spread;
    const float4* pos; const float2* lj; const int32_t* tile_idx; const int32_t* tile_cnt; const uint2* nbr; const int32_t* wave_rows; const float4* blk_center;
    float4* frc; float4* parts; int64_t part_stride;                      // group 0 → frc, group g → parts + (g − 1)·part_stride
    const uint16_t* item_of;                                              // [n_blocks·GS] nullable: which (group · n_blocks + block) workgroup w takes (k_gs_balance)
    unsigned long long* dbg;                                              // builds with -DMHIP_STAMPS=1: [workgroup][wave][8] time stamps (engine: MOLLYHIP_DBG_TIMES, tools/gs_times.py)
};
size_t gs_lds_bytes(int q_lds, int BI, int JSW);
void launch_regroup(const RegroupArgs& A, int n_blocks, hipStream_t stream);
void launch_gs_balance(const int32_t* rows_gs, int n_blocks, int JS, int GS, int waves_per_sub, int period, uint16_t* item_of, hipStream_t stream);
void launch_forces_gs(const GsArgs& A, int coulm, bool minimg, hipStream_t stream);
template <class T> struct PmeP; template <class T> struct BondedArgs;
void launch_pair_spread_bonded(const GsArgs& A, int n_pair, int coulm, bool minimg, int order, int64_t n_atoms, float* rgrid, const PmeP<float>& P, int n_spread, const BondedArgs<float>& B, int n_term_wg,
                               size_t lds_bytes, hipStream_t stream, const double* cm_fin_in = nullptr, int cm_fin_n = 0, double* cm_fin_out = nullptr);
size_t spread_head_bytes_f32(int order);

}  // namespace mhip
