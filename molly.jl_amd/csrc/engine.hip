// engine.hip — host side of libmollyhip.so: context, neighbour rebuild pipeline, kernel launches and the
// C ABI of include/mollyhip.h.  One context = one GPU = one HIP stream (one process per GPU).
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include <unistd.h>

#include "bonded.h"
#include "stochastic.h"
#include "pme.h"
#include "step_fused.h"
#include "hilbert.h"
#include "kernels.h"
#include "replan.h"
#include "halo_step.h"
#include "forces_launch.h"
#include "sortscan.h"

namespace mhip {

static thread_local std::string g_create_error;

template <class U> struct DBuf {
    U* p = nullptr; size_t n = 0;
    void reserve(size_t m) { if (m > n) { if (p) (void)hipFree(p); p = nullptr; MHIP_HIP(hipMalloc((void**)&p, std::max<size_t>(m, 1) * sizeof(U))); n = m; } }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

struct EngineBase {
    std::string err;
    virtual ~EngineBase() {}
    virtual void set_stream(void* s) = 0;
    virtual void synchronize() = 0;
    virtual void set_profiling(bool) = 0;
    virtual void set_atom_counts(int64_t, int64_t) = 0;
    virtual void set_atoms(const void*, const void*, const void*, const void*, const void*, int) = 0;
    virtual void set_exceptions(const int32_t*, const int32_t*, int64_t, const int32_t*, const int32_t*, int64_t) = 0;
    virtual void set_bonds(int64_t, const int32_t*, const int32_t*, const void*, const void*) = 0;
    virtual void set_angles(int64_t, const int32_t*, const int32_t*, const int32_t*, const void*, const void*) = 0;
    virtual void set_torsions(int64_t, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const void*, const void*) = 0;
    virtual void set_ewald_exclusions(int64_t, const int32_t*, const int32_t*) = 0;
    virtual void set_state(const void*, const void*, int) = 0;
    virtual void get_state(void*, void*, int) = 0;
    virtual void forces(int64_t, int, void*, int) = 0;
    virtual void specific_forces(int, void*, int) = 0;
    virtual double potential_energy(int64_t) = 0;
    virtual double specific_potential_energy() = 0;
    virtual double kinetic_energy() = 0;
    virtual void remove_cm() = 0;
    virtual void check_finite() = 0;
    virtual void vv_run(int64_t, int64_t, double, int) = 0;
    virtual void vv_init(int64_t) = 0;
    virtual void langevin_run(int64_t, int64_t, double, double, double, int, uint64_t, uint64_t) = 0;
    virtual void redraw_velocities(int, double, double, uint64_t, uint64_t) = 0;
    virtual void set_andersen(double, double, uint64_t) = 0;
    virtual void vv_stage1(double) = 0;
    virtual void vv_stage2(int64_t, double) = 0;
    virtual void rebuild_now(int64_t) = 0;
    virtual int64_t export_neighbors(int32_t*, int32_t*, uint8_t*, int64_t) = 0;
    virtual void export_order(int32_t*, int64_t) = 0;
    virtual void get_stats(mhip_stats*) = 0;
    virtual void gather_coords(const int32_t*, const void*, int64_t, void*) = 0;
    virtual void scatter_coords(int64_t, int64_t, const void*) = 0;
    virtual void cm_momentum(double*) = 0;
    virtual void shift_velocities(const double*) = 0;
    virtual void cm_momentum_dev(double*) = 0;
    virtual void remove_cm_dev(const double*) = 0;
    virtual void pairwise_virial(int64_t, double*) = 0;
    virtual void specific_virial(double*) = 0;
    virtual void general_virial(double*) = 0;
    virtual void set_pme(int32_t, const int32_t*, double, double) = 0;
    virtual void set_triclinic(const double*, int32_t) = 0;
    virtual void set_box(const double*, const double*) = 0;
    virtual void general_forces(int, void*, int) = 0;
    virtual double general_potential_energy() = 0;
    virtual void set_ghost_margin(double) = 0;
    virtual void plan_disp2_dev(float*) = 0;
    virtual void plan_state_dev(float*) = 0;
    virtual int plan_decide(int64_t, const float*, int32_t*) = 0;
    virtual void set_halo_plan(const mhip_halo_plan*) = 0;
    virtual void halo_start(double) = 0;
    virtual void halo_mid(int64_t, double, int32_t, double*, int32_t) = 0;
    virtual void request_prune() = 0;
    virtual void halo_begin(double, const int32_t*, const void*, int64_t, void*) = 0;
    virtual int halo_interior(int64_t) = 0;
    virtual void halo_end(int64_t, double, int64_t, int64_t, const void*, double*) = 0;
    virtual void halo_end_parts(int64_t, double, int64_t, int64_t, const void*, double*, int32_t) = 0;
    virtual void remove_cm_parts_dev(const double*, int32_t) = 0;
    virtual void halo_region(int64_t, int32_t, int32_t, void*) = 0;
    virtual void halo_open_peer(int32_t, const void*) = 0;
    virtual void set_halo_routes(const mhip_halo_routes*) = 0;
    virtual int halo_selftest() = 0;
    virtual void set_launch_config(int, int) = 0;
    virtual int tune_launch(int, mhip_launch_trial*, int) = 0;
    virtual void domain_run(int64_t, int64_t, double, int32_t, double*, int32_t, int64_t*, int32_t*, int64_t*) = 0;
    virtual void set_domain(const mhip_domain_geometry*, const int64_t*) = 0;
    virtual void domain_info(int64_t*) = 0;
    virtual void domain_export(int64_t*, void*) = 0;
};

// hipEvent stage timers (only active while profiling is on)
struct Prof {
    static constexpr int NS = 8;
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[NS];
    size_t used[NS] = {};
    double ms[NS] = {};
    int64_t calls[NS] = {};
    void begin(int st, hipStream_t s) {
        if (!on) return;
        if (used[st] == ev[st].size()) { hipEvent_t a, b; MHIP_HIP(hipEventCreate(&a)); MHIP_HIP(hipEventCreate(&b)); ev[st].push_back({a, b}); }
        MHIP_HIP(hipEventRecord(ev[st][used[st]].first, s));
    }
    void end(int st, hipStream_t s) {
        if (!on) return;
        MHIP_HIP(hipEventRecord(ev[st][used[st]].second, s));
        if (++used[st] >= 8192) resolve(s);
    }
    void resolve(hipStream_t s) {
        MHIP_HIP(hipStreamSynchronize(s));
        for (int st = 0; st < NS; ++st) {
            for (size_t k = 0; k < used[st]; ++k) { float t = 0; MHIP_HIP(hipEventSynchronize(ev[st][k].second)); MHIP_HIP(hipEventElapsedTime(&t, ev[st][k].first, ev[st][k].second)); ms[st] += t; ++calls[st]; }
            used[st] = 0;
        }
    }
    void reset() { for (int st = 0; st < NS; ++st) { used[st] = 0; ms[st] = 0; calls[st] = 0; } }
    // events for the first launches are made before the pass that uses them: creating two per launch on the way (≈ 10 µs each on the
    // host) let the stream run dry between a begin event and its kernel, and the gap counted as kernel time
    void reserve(int st, size_t n) { while (ev[st].size() < n) { hipEvent_t a, b; MHIP_HIP(hipEventCreate(&a)); MHIP_HIP(hipEventCreate(&b)); ev[st].push_back({a, b}); } }
    void release() { for (int st = 0; st < NS; ++st) { for (auto& e : ev[st]) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); } ev[st].clear(); } }
};

static int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v && *v ? std::atoi(v) : dflt; }


template <class T> class Engine final : public EngineBase {
    using T4 = typename Vec<T>::T4;
    using T2 = typename Vec<T>::T2;

    mhip_config cfg;
    GridP<T> G;
    InterP<T> I;
    int ljm = LJ_OFF, coulm = MHIP_COUL_NONE;
    int64_t cap, n_owned, n_ghost, n_tot;
    hipStream_t stream = nullptr; bool own_stream = false;
    int device = 0;

    // per-atom state in sorted order, double-buffered for the re-sort
    DBuf<T4> pos[2], vel[2], frc[2]; DBuf<T2> lj[2]; DBuf<int32_t> orig[2]; DBuf<int32_t> inv;
    int cur = 0;
    // sort / cells
    DBuf<uint32_t> key_in, key_out, cell_rank; DBuf<int32_t> idx_in, perm, cell_cnt, cell_start; DBuf<unsigned char> cub_tmp;
    bool hilbert_ok = true;
    // exceptions (CSR over caller indices)
    DBuf<int32_t> xl_start; DBuf<uint32_t> xl_list; bool has_exc = false; int xl_span = 0; int64_t n_special = 0;
    static constexpr int X_CAP = 32;
    static constexpr int MAX_THREADS = BlockLimits<T>::max_threads;
    // blocks
    int BI = 256, JS = 1, n_blocks = 0, T_cap = 0, R_cap = 0, C_cap = 0, max_tile = 0, max_rows = 0;
    int ljm_base = LJ_OFF;   // LJ mode implied by the interaction; ljm may be upgraded to the uniform fast path
    // entry format of the pair lists in use (kernels.h: entry_slot / make_entry), fixed at the outer search: byte-offset entries for the
    // fp32 one-type LJ fluids (their packed loop then spends one instruction per partner on its LDS address), slot + flag otherwise
    int eshift = 0;
    bool scaled_entries_off = false;   // a tile too large for 14-bit slots was met: back to the slot + flag format for good
    int want_eshift() const {
        return (!scaled_entries_off && std::is_same<T, float>::value && ljm == LJ_DIST_UNIFORM && coulm == MHIP_COUL_NONE && n_special == 0 && !tri_mode && !G.no_list) ? ESHIFT_SCALED : 0;
    }
    int slot_cap() const { return ((want_eshift() ? SLOT_MAX_SCALED : TILE_SLOT_MAX) - 1) & ~3; }
    DBuf<int32_t> tile_idx, tile_cnt, wave_rows; DBuf<uint2> nbr; DBuf<T4> blk_center;
    // dual pair list: outer list (nbr / wave_rows, radius r_list + margin) and the inner list filtered from it
    DBuf<int32_t> wave_rows_in, tile_idx_in, tile_cnt_in, rows_x, tile_idx_x, tile_cnt_x; DBuf<uint2> nbr_in, nbr_x; DBuf<T4> pos_snap; DBuf<float> blk_disp2; int max_tile_in = 0;
    bool inner_valid = false, prune_disp_exceeded = false;
    // entries per (sub-list, lane) of the inner list as the last prune wrote it / of the outer list as the search wrote it: what k_regroup deals to the groups
    DBuf<uint16_t> cnt_in, cnt_outer; bool cnt_outer_valid = false;
    // the group-split pair pass of small systems (forces_gs.hip): the inner list re-dealt into GS groups per block after every prune, the
    // partial forces of groups 1 .. GS − 1 (group 0 writes the force array), and which prune the list belongs to
    DBuf<uint2> nbr_gs; DBuf<int32_t> rows_gs; DBuf<T4> frc_parts; int64_t gs_list_id = -1; bool gs_used = false;
    // A prune that could drop nothing — the inner skin has grown to the reference's own r_list − cutoff, the outer list has no margin and was searched at
    // this very step (6mrr at 0.5 fs: every rebuild) — is skipped: the outer list IS the inner list (inner_is_outer: the passes over the inner list read
    // the outer arrays), and k_regroup deals it to the groups straight away.  It was a 61 µs pass (the PRUNE variant of k_forces, 1024-lane blocks) per
    // rebuild where the group-split launch that now computes the same forces takes 25 µs, spreading and bonded terms included.
    bool inner_is_outer = false; const bool adopt_env = env_int("MOLLYHIP_ADOPT_OUTER", 1) != 0; int64_t n_adopted = 0;
    bool fuse_spread_next = false, spread_fused = false, fuse_terms_next = false, terms_fused = false;
    const int gs_env = env_int("MOLLYHIP_GROUP_SPLIT", -1);      // 0: off; 2 / 4: groups per block; −1: automatic
    int gs_groups() const {
        if (!std::is_same<T, float>::value || ljm != LJ_DIST || !(coulm == MHIP_COUL_REACTION_FIELD || (coulm == MHIP_COUL_EWALD_DIRECT && I.approx_erfc))) return 0;
        if (!dual || tri_mode || n_ghost > 0 || G.no_list || gs_env == 0) return 0;
        const int want = gs_env > 0 ? gs_env : 4;
        if (want != 4 || JS % want != 0 || BI * (JS / want) != 256) return 0;      // (k_forces_gs is a 256-lane workgroup: 64 atoms × 4 waves of a 16-way j-split)
        if (gs_env < 0 && (int64_t)n_blocks * JS * (BI / WAVE) > 2 * 4096) return 0;      // enough workgroups already: large systems balance themselves
        if ((int64_t)n_blocks * want > 16384) return 0;      // (forced or not: the (block, group) items are uint16 and k_gs_balance stages n_blocks·GS ints in 64 KiB of LDS)
        return want;
    }
    DBuf<int32_t> blk_ghost, blk_ghost_in; bool ghost_flags_ok = false, ghost_flags_in_ok = false, interior_done = false;
    int64_t last_prune_step = 0;
    int64_t pass_step = 0;       // the MD step whose coordinates the pair pass being launched sees (recorded as the step of a prune)
    DBuf<T4> pos_snap_in;        // coordinates at the last prune (validity of the inner list: 2·displacement <= skin)
    double skin = 0; int64_t n_disp_checks = 0;
    // The inner list of the dual scheme only has to hold every pair inside the CUTOFFS (mhip_export_neighbors filters the outer list to
    // r_list itself): it is pruned to rc_max + skin_in, skin_in <= skin.  A smaller radius means fewer entries per force pass (∝ r³) and
    // more frequent prunes; MOLLYHIP_INNER_SKIN_PM (picometres, default 100) sets it, the ghosted path keeps skin_in = skin.
    double skin_in = 0, rc_max_ = 0; T r_prune2 = 0;
    // ghosted sub-domain whose ghost shell reaches r_list + ghost_margin: the ghost PLAN then lives as long as an outer list
    // (until some atom moved ghost_margin/2), so the dual list works here too and the host re-plans only when mhip_plan_disp2_dev says so
    double ghost_margin = 0; const double* cm_ext = nullptr;
    int tri_mode = 0; double tri_bv[9] = {};   // TriclinicBoundary: 0 off, 1 approx_images, 2 exact images; basis vectors row-major
    bool tri_grid = false;                     // … with a cell grid in height-scaled fractional coordinates (else: one cell, every block sees every atom)
    long long grid_key = -1;
    double skin_in_adapted = 0;
    bool engine_sched = false;   // … unless it hands the reduced displacements to mhip_plan_decide: then the engine's own criteria (inner skin, drift bound) decide
    bool host_prune = false;     // ghost plans: the HOST decides, collectively over the ranks, when the inner list is re-pruned (mhip_request_prune)
    // single list, same idea: a rebuild step whose displacement check shows the list still covers every cutoff sphere is skipped
    bool lazy_single = false; int64_t n_skipped = 0;
    bool dual = false, dual_disabled = false, margin_zero = false, want_margin_zero = false; int margin_halvings = 0; int early_outer = 0; double outer_margin = 0; int64_t last_outer_step = 0, n_outer = 0, n_filters = 0; T r_in = 0, r_in2 = 0;
    DBuf<int32_t> flags; int32_t* h_flags = nullptr;
    int64_t total_rows = 0, outer_rows = 0;      // rows (of four entries per lane, per wave) of the list the plain passes walk | of the outer list as searched
    // reductions
    DBuf<double> red_part, red_out, cm_step; double* h_red = nullptr; DBuf<T> vcm;
    int n_cm_step = 0;   // cm_pending == 2: v_cm still lives as the per-block partials of the last k_vv2 (cm_step[0..4*n_cm_step))
    // staging for host pointers
    DBuf<T> stage_a, stage_b; DBuf<int32_t> stage_i;
    // bonded
    Bonded<T> bonded;
    // general interaction: PME reciprocal space (ewald.jl:361-929)
    Pme<T> pme; double pc_sum = 0, pc_abs2_sum = 0; bool pc_valid = false;
    // small systems (bonded terms + PME): the last launch of the force chain leaves bonded sums + reciprocal-space forces in frc_side; the integrator
    // (or fold_side_forces) adds it.  (Side streams, CU-masked streams and any-order launches for these chains all measured slower than one stream: DESIGN §4.)
    DBuf<T4> frc_side; const T4* pend_a = nullptr;

    int cm_pending = 0; bool stale = true, minimg = false, params_set = false, state_set = false, frc_valid = false;
    bool coords_moved = false, export_needs_search = false; int64_t n_set_state_refresh = 0;
    int64_t last_build_step = std::numeric_limits<int64_t>::min();
    int64_t n_rebuilds = 0, n_force_calls = 0, n_gs_passes = 0; double last_rebuild_ms = 0;
    size_t lds_force = 0; int tile_lds = 0; bool segmented = false; int last_pass_tile = 0;
    Prof prof;
    // MOLLYHIP_DEBUG=2: drain the stream, then name the launch that follows on stderr — the last name a dying process printed is the
    // kernel that faulted (a GPU fault aborts the process from the runtime's callback, no status ever comes back)
    const bool trace_on = env_int("MOLLYHIP_DEBUG", 0) >= 2;
    const bool debug_on = env_int("MOLLYHIP_DEBUG", 0) == 1 || env_int("MOLLYHIP_DEBUG", 0) >= 3;      // MOLLYHIP_DEBUG=1 (or 3: both): the list-maintenance decisions on stderr (read once, when the context is made)
    void tr(const char* what) {
        if (!trace_on) return;
        (void)hipStreamSynchronize(stream);
        std::fprintf(stderr, "[mhip %d] %s (n_owned %lld n_ghost %lld BI %d JS %d T_cap %d R_cap %d)\n", (int)getpid(), what, (long long)n_owned, (long long)n_ghost, BI, JS, T_cap, R_cap);
        std::fflush(stderr);
    }

  public:
    explicit Engine(const mhip_config& c) : cfg(c) {
        cap = c.n_atoms; n_owned = cap; n_ghost = 0; n_tot = cap;
        if (cap <= 0) throw ApiError{MHIP_ERR_INVALID, "n_atoms must be positive"};
        if (cap > (int64_t)1 << 30) throw ApiError{MHIP_ERR_INVALID, "n_atoms too large for int32 indexing"};
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw ApiError{MHIP_ERR_NO_DEVICE, "no HIP device visible (libmollyhip has no CPU fallback)"};
        device = c.device_id;
        if (device < 0 || device >= ndev) throw ApiError{MHIP_ERR_INVALID, "device_id out of range"};
        MHIP_HIP(hipSetDevice(device));
        MHIP_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); own_stream = true;
        setup_inter(); setup_grid();
        for (int k = 0; k < 2; ++k) { pos[k].reserve(cap); vel[k].reserve(cap); frc[k].reserve(cap); lj[k].reserve(cap); orig[k].reserve(cap); }
        inv.reserve(cap); key_in.reserve(cap); key_out.reserve(cap); idx_in.reserve(cap); perm.reserve(cap);
        flags.reserve(N_FLAGS); red_out.reserve(8); vcm.reserve(4); cm_step.reserve(2 * 4 * 1024);   // two halves: k_vv_mid reads one while it writes the other
        MHIP_HIP(hipHostMalloc((void**)&h_flags, N_FLAGS * sizeof(int32_t)));
        MHIP_HIP(hipHostMalloc((void**)&h_red, 8 * sizeof(double)));
        for (int k = 0; k < 2; ++k) { MHIP_HIP(hipMemsetAsync(pos[k].p, 0, cap * sizeof(T4), stream)); MHIP_HIP(hipMemsetAsync(vel[k].p, 0, cap * sizeof(T4), stream)); MHIP_HIP(hipMemsetAsync(frc[k].p, 0, cap * sizeof(T4), stream)); MHIP_HIP(hipMemsetAsync(lj[k].p, 0, cap * sizeof(T2), stream)); }
        std::vector<int32_t> iota(cap); for (int64_t i = 0; i < cap; ++i) iota[i] = (int32_t)i;
        MHIP_HIP(hipMemcpyAsync(orig[0].p, iota.data(), cap * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        MHIP_HIP(hipMemcpyAsync(inv.p, iota.data(), cap * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        size_t tb = 0; MHIP_HIP(sort_pairs_u32(nullptr, tb, key_in.p, key_out.p, idx_in.p, perm.p, (int)cap, 32, stream));
        size_t tb2 = 0; MHIP_HIP(exclusive_sum_i32(nullptr, tb2, cell_cnt.p, cell_start.p, 2 * G.ncell + 1, stream));
        cub_tmp.reserve(std::max(tb, tb2) + 256);
        choose_blocking();
    }
    ~Engine() override {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        for (int k = 0; k < 2; ++k) { pos[k].release(); vel[k].release(); frc[k].release(); lj[k].release(); orig[k].release(); }
        inv.release(); key_in.release(); key_out.release(); cell_rank.release(); idx_in.release(); perm.release(); cell_cnt.release(); cell_start.release(); cub_tmp.release();
        pos_snap_in.release(); cnt_in.release(); cnt_outer.release();
        nbr_gs.release(); rows_gs.release(); frc_parts.release(); wave_rows_in.release(); nbr_in.release(); pos_snap.release(); blk_disp2.release(); tile_idx_in.release(); tile_cnt_in.release(); rows_x.release(); nbr_x.release(); tile_idx_x.release(); tile_cnt_x.release(); blk_ghost.release(); blk_ghost_in.release();
        xl_start.release(); xl_list.release(); tile_idx.release(); tile_cnt.release(); wave_rows.release(); nbr.release(); blk_center.release();
        flags.release(); red_part.release(); red_out.release(); cm_step.release(); vcm.release(); stage_a.release(); stage_b.release(); stage_i.release(); bonded.release(); pme.release(); frc_scratch.release(); nl_counter.release(); state_changed.release(); pos_alt.release(); cm_blk.release(); cm_pub.release();
        xf_release(); dom_release(); hx.release();
        prof.release();
        frc_side.release();
        if (h_flags) (void)hipHostFree(h_flags);
        if (h_trk) (void)hipHostFree(h_trk);
        if (ev_trk) (void)hipEventDestroy(ev_trk);
        if (h_prune) (void)hipHostFree(h_prune);
        if (ev_prune) (void)hipEventDestroy(ev_prune);
        if (h_red) (void)hipHostFree(h_red);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }

  private:
    // ---------------------------------------------------------------------------------------------
    void setup_inter() {
        const mhip_interactions& p = cfg.inter;
        std::memset(&I, 0, sizeof(I));
        I.lj = p.lj_enabled; I.lj_cut = p.lj_cutoff_kind; I.lj_rc = T(p.lj_rc); I.lj_rc2 = I.lj_rc * I.lj_rc; I.lj_ra = T(p.lj_ra); I.lj_w = T(p.lj_weight_special);
        I.coul = p.coul_kind; I.coul_cut = p.coul_cutoff_kind; I.c_rc = T(p.coul_rc); I.c_rc2 = I.c_rc * I.c_rc; I.c_ra = T(p.coul_ra);
        I.ke = T(p.coul_ke); I.c_w = T(p.coul_weight_special); I.alpha = T(p.ewald_alpha); I.approx_erfc = p.ewald_approx_erfc;
        I.two_over_sqrt_pi = T(2) / std::sqrt(T(M_PI));
        if (p.lj_cutoff_kind < 0 || p.lj_cutoff_kind > 5 || p.coul_cutoff_kind < 0 || p.coul_cutoff_kind > 5 || p.coul_kind < 0 || p.coul_kind > 3)
            throw ApiError{MHIP_ERR_INVALID, "unknown cutoff / coulomb kind"};
        if (p.coul_kind == MHIP_COUL_REACTION_FIELD) {   // coulomb.jl:764-768, 799-803, evaluated in T like the reference
            T rc = I.c_rc, rc3 = rc * rc * rc, e = T(p.rf_dielectric);
            if (std::isinf(p.rf_dielectric)) { I.krf = T(1) / (T(2) * rc3); I.crf = T(3) * (T(1) / (T(2) * rc)); }
            else { I.krf = (T(1) / rc3) * (e - T(1)) / (T(2) * e + T(1)); I.crf = (T(1) / rc) * (T(3) * e) / (T(2) * e + T(1)); }
        }
        ljm = !p.lj_enabled ? LJ_OFF : (p.lj_cutoff_kind == MHIP_CUTOFF_DISTANCE ? LJ_DIST : LJ_GENERIC);
        ljm_base = ljm;
        coulm = p.coul_kind;
        if ((p.lj_enabled && p.lj_cutoff_kind != MHIP_CUTOFF_NONE && !(p.lj_rc > 0)) ||
            (p.coul_kind >= MHIP_COUL_REACTION_FIELD && !(p.coul_rc > 0)))
            throw ApiError{MHIP_ERR_INVALID, "cutoff distance must be positive"};
    }

    void setup_grid() {
        std::memset(&G, 0, sizeof(G));
        G.no_list = !(cfg.r_list > 0) || std::isinf(cfg.r_list);
        r_in = G.no_list ? std::numeric_limits<T>::infinity() : T(cfg.r_list);
        r_in2 = G.no_list ? std::numeric_limits<T>::infinity() : r_in * r_in;                 // dist_cutoff^2, neighbors.jl:400
        // dual pair list: search with r_list + margin (MOLLYHIP_OUTER_MARGIN_PM in picometres, 0 disables), prune to the inner radius when displacement
        // says so, search again when the margin is used up
        outer_margin = (G.no_list || dual_disabled) ? 0.0 : std::ldexp(env_int("MOLLYHIP_OUTER_MARGIN_PM", 200) * 1e-3, -margin_halvings);
        // margin 0 keeps the two-list machinery without the wider search: the list is built with r_list and pruned once, right away —
        // which compacts the tile to the atoms the rows refer to (a tile that needed two LDS segments in fp64 then fits in one)
        if (margin_zero && n_ghost == 0) outer_margin = 0;
        if (n_ghost > 0) outer_margin = std::min(outer_margin, ghost_margin);   // the shell handed over must cover the outer radius
        // extent of the cell grid per axis: the box side, or — TriclinicBoundary — the perpendicular height of the cell along that
        // axis (the grid lives in u = s·h, common.h): h = V / |b × c|, V / |c × a|, V / |a × b| for the basis a ∥ x, b in the xy plane
        double ext[3] = {cfg.box[0], cfg.box[1], cfg.box[2]};
        if (tri_mode) {
            const double* a = tri_bv; const double* b = tri_bv + 3; const double* c = tri_bv + 6;
            const double V = a[0] * b[1] * c[2];
            auto cross_norm = [](const double* u, const double* v) { const double x = u[1] * v[2] - u[2] * v[1], y = u[2] * v[0] - u[0] * v[2], z = u[0] * v[1] - u[1] * v[0]; return std::sqrt(x * x + y * y + z * z); };
            ext[0] = V / cross_norm(b, c); ext[1] = V / cross_norm(c, a); ext[2] = V / cross_norm(a, b);
        }
        for (int d = 0; d < 3; ++d) if (cfg.periodic[d] && cfg.r_list + outer_margin > 0.5 * ext[d]) outer_margin = 0;   // keep r_outer <= L/2 (triclinic: half the cell height)
        // walking the outer list is only equivalent to walking the reference's list if every interaction vanishes beyond a
        // cutoff <= r_list; a NoCutoff interaction summed over a neighbour list depends on list membership itself
        const mhip_interactions& ip = cfg.inter;
        const bool lj_cut_ok = !ip.lj_enabled || (ip.lj_cutoff_kind != MHIP_CUTOFF_NONE && ip.lj_rc <= cfg.r_list);
        const bool coul_cut_ok = ip.coul_kind == MHIP_COUL_NONE || (ip.coul_kind == MHIP_COUL_PLAIN ? (ip.coul_cutoff_kind != MHIP_CUTOFF_NONE && ip.coul_rc <= cfg.r_list) : ip.coul_rc <= cfg.r_list);
        {   // skin = r_list − largest cutoff: how far a pair may close in before the inner list must be re-pruned
            double rc_max = 0;
            if (ip.lj_enabled) rc_max = std::max(rc_max, ip.lj_rc);
            if (ip.coul_kind != MHIP_COUL_NONE) rc_max = std::max(rc_max, ip.coul_rc);
            skin = G.no_list ? 0.0 : cfg.r_list - rc_max;
            rc_max_ = rc_max;
            skin_in = ((n_ghost > 0 || host_prune) && !engine_sched) ? skin : std::min(skin, std::max(skin_in_adapted, std::max(1, env_int("MOLLYHIP_INNER_SKIN_PM", 100)) * 1e-3));   // (a skin the run has grown stays grown)
            const T rp = T(rc_max + skin_in);
            r_prune2 = (skin_in < skin) ? rp * rp : r_in2;
        }
        const int S = 2;      // cells of >= r_search / 2 per side of the stencil
        // A TriclinicBoundary keeps the dual list and the displacement-skipped rebuilds when the box still gets a real cell grid with the
        // wider search radius (displacements are measured on the nearest image, disp_image; pruning and searching work on block-local
        // Cartesian coordinates, kernels.h); a box too small for a grid keeps the one-cell form: plain fixed-cadence lists, exact images.
        bool tri_lists_ok = !tri_mode;
        if (tri_mode && !G.no_list)
            for (int d = 0; d < 3; ++d) tri_lists_ok = tri_lists_ok || (int)std::floor(ext[d] / ((cfg.r_list + outer_margin) / S)) > 2 * S + 1;
        if (!tri_lists_ok) outer_margin = 0;
        dual = (outer_margin > 0 || (margin_zero && n_ghost == 0 && !G.no_list && !dual_disabled)) && lj_cut_ok && coul_cut_ok && skin > 0 && tri_lists_ok;   // ghosted: only with a ghost margin (else re-planned every rebuild)
        lazy_single = !dual && !G.no_list && lj_cut_ok && coul_cut_ok && skin > 0 && n_ghost == 0 && tri_lists_ok;
        if (debug_on) std::fprintf(stderr, "[mhip] grid: dual %d margin %.3f skin %.3f lj_ok %d coul_ok %d ghosts %lld\n", (int)dual, outer_margin, skin, (int)lj_cut_ok, (int)coul_cut_ok, (long long)n_ghost);
        const double r_search = G.no_list ? 0.0 : cfg.r_list + (dual ? outer_margin : 0.0);
        G.r_list = G.no_list ? std::numeric_limits<T>::infinity() : T(r_search);
        G.r_list2 = G.no_list ? std::numeric_limits<T>::infinity() : (dual ? G.r_list * G.r_list : r_in2);
        tri_grid = false;
        for (int pass = 0; pass < 2; ++pass) {
            for (int d = 0; d < 3; ++d) {
                if (!(cfg.box[d] > 0) || std::isinf(cfg.box[d]) || std::isnan(cfg.box[d])) throw ApiError{MHIP_ERR_INVALID, "box side lengths must be positive and finite"};
                G.L[d] = T(cfg.box[d]); G.invL[d] = T(1) / G.L[d]; G.origin[d] = cfg.periodic[d] ? T(0) : T(cfg.origin[d]); G.periodic[d] = cfg.periodic[d] ? 1 : 0;
                int nc = 1;
                if (!G.no_list && (!tri_mode || tri_grid)) { nc = (int)std::floor(ext[d] / (r_search / S)); nc = std::max(1, std::min(nc, 1024)); }
                G.nc[d] = nc; G.cs[d] = T(ext[d] / nc); G.inv_cs[d] = T(nc / ext[d]);
                G.stencil[d] = S; G.all_cells[d] = (G.no_list || 2 * S + 1 >= nc) ? 1 : 0;
                G.hgt[d] = T(ext[d]);
            }
            // A triclinic box gets a real cell grid when at least one axis has more cells than a stencil spans (else every block would
            // see every atom anyway: the one-cell form of round 1, all distances by the exact in-loop minimum image).
            if (pass == 0 && tri_mode && !G.no_list) {
                bool any = false;
                for (int d = 0; d < 3; ++d) any = any || (int)std::floor(ext[d] / (r_search / S)) > 2 * S + 1;
                if (any) { tri_grid = true; continue; }
            }
            break;
        }
        G.tri_grid = tri_grid ? 1 : 0;
        // keep the cell table small: coarsen until ncell <= 2^22
        while ((int64_t)G.nc[0] * G.nc[1] * G.nc[2] > (1 << 22)) for (int d = 0; d < 3; ++d) { G.nc[d] = std::max(1, G.nc[d] / 2); G.cs[d] = T(ext[d] / G.nc[d]); G.inv_cs[d] = T(G.nc[d] / ext[d]); }
        G.ncell = G.nc[0] * G.nc[1] * G.nc[2];
        G.triclinic = tri_mode;
        if (tri_mode) {   // constants of TriclinicBoundary's constructor (spatial.jl:187-210), rounded to T as the struct stores them
            const double* a = tri_bv; const double* b = tri_bv + 3; const double* c = tri_bv + 6;
            for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) G.bv[r][k] = T(tri_bv[3 * r + k]);
            G.rs[0] = T(1) / T(a[0]); G.rs[1] = T(1) / T(b[1]); G.rs[2] = T(1) / T(c[2]);
            G.cot_bc = T(std::fabs((b[1] * c[1] + b[2] * c[2]) / (b[1] * c[2] - b[2] * c[1])));
            G.cxz = T(c[0] / std::fabs(c[2])); G.cyz = T(c[1] / std::fabs(c[2])); G.cot_ab = T(b[0] / b[1]);
        }
        std::vector<uint32_t> rank;
        hilbert_ok = hilbert_cell_ranks(G.nc[0], G.nc[1], G.nc[2], rank);
        cell_rank.reserve(G.ncell);
        MHIP_HIP(hipMemcpy(cell_rank.p, rank.data(), (size_t)G.ncell * sizeof(uint32_t), hipMemcpyHostToDevice));
        cell_cnt.reserve(2 * (size_t)G.ncell + 1); cell_start.reserve(2 * (size_t)G.ncell + 1);
    }

    static int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

    size_t build_lds_bytes(int tcap, int bi, int ccap, bool walk = true) const {
        // tile (coords + slot) | max(cell tables, per-wave candidate bit masks) | scan scratch
        // tile (coords + caller index) | cell tables | scan scratch | per-lane exception lists
        return (size_t)tcap * (3 * sizeof(float) + (has_exc ? 4 : 0)) + (2 * (size_t)ccap + 2) * 4 + 8 + (size_t)bi * JS * 4 + (has_exc ? (size_t)X_CAP * bi * 4 : 0) + (walk ? ((size_t)ccap + 2) * 4 : 0) + 64;   // … | first tile slot per box cell (walk)
    }
    size_t force_lds_bytes(int tlds) const {
        const bool per_atom_lj = (ljm == LJ_DIST || ljm == LJ_GENERIC);
        size_t tile = (size_t)(tlds + 1) * (sizeof(T4) + (per_atom_lj ? sizeof(T2) : 0));
        size_t red = (size_t)JS * 4 * BI * sizeof(T);
        return std::max(std::max(tile, red), (size_t)BI * sizeof(double)) + 32;
    }

    void choose_blocking() {
        // i-block size and j-split: enough waves to fill 256 CUs × 4 SIMDs even for small systems
        int bi, js;
        if (n_owned >= 100000) { bi = 256; js = 2; }        // measured on MI355X: lj1m 256x2, 6mrr 64x16 (profiles/)
        else if (n_owned >= 40000) { bi = 128; js = 4; }
        else { bi = 64; js = 16; }
        if (user_bi) { bi = user_bi; js = user_js; }         // mhip_set_launch_config / the winner of mhip_optimize_launch_config
        if (bi != 64 && bi != 128 && bi != 256) throw ApiError{MHIP_ERR_INVALID, "block_atoms must be 64, 128 or 256"};
        js = std::min(js, MAX_THREADS / bi);                // the block kernels' launch bound (fp64: 512 lanes, 256 VGPRs per lane)
        if (js < 1) throw ApiError{MHIP_ERR_INVALID, "j_split must be positive"};
        BI = bi; JS = js;
        estimate_capacities();
    }

    // ≙ set_cuda_launch_config! / reset_cuda_launch_config! (src/cuda_config.jl:17-47): the workgroup shape of the search and pair
    // kernels, block_atoms i-atoms × j_split waves per atom's list; (0, 0) returns to the automatic choice.  Lists are rebuilt.
    int user_bi = 0, user_js = 0;
    void set_launch_config(int bi, int js) override {
        if (bi == 0 && js == 0) { user_bi = user_js = 0; }
        else {
            if (bi != 64 && bi != 128 && bi != 256) throw ApiError{MHIP_ERR_INVALID, "block_atoms must be 64, 128 or 256 (or 0, 0 for the automatic choice)"};
            if (js < 1 || (js & (js - 1)) || bi * js > MAX_THREADS) throw ApiError{MHIP_ERR_INVALID, "j_split must be a power of two with block_atoms * j_split within the launch bound (1024 lanes in fp32, 512 in fp64)"};
            user_bi = bi; user_js = js;
        }
        flush_cm();
        choose_blocking(); stale = true;
    }

    // ≙ optimize_cuda_launch_config! (ext/MollyCUDAExt.jl:594-642, src/cuda_config.jl:53-62): time a small candidate set on THIS system
    // and keep the winner.  A trial = lists rebuilt in that shape (search + prune, untimed), then n_passes plain force passes between
    // two HIP events.  Shapes whose tile does not fit the LDS, or that the build had to shrink, are reported with us_per_pass < 0 /
    // under the shape they ended in.  A shape set by mhip_set_launch_config is replaced by the winner.
    int tune_launch(int n_passes, mhip_launch_trial* trials, int max_trials) override {
        if (!state_set || !params_set) throw ApiError{MHIP_ERR_STATE, "set_atoms and set_state must be called before tuning"};
        if (n_ghost > 0) throw ApiError{MHIP_ERR_UNSUPPORTED, "launch shapes are tuned on single-domain contexts (the ranks of a decomposition must agree on one)"};
        if (n_passes < 1) throw ApiError{MHIP_ERR_INVALID, "n_passes must be positive"};
        static const int cand[][2] = {{256, 4}, {256, 2}, {128, 8}, {128, 4}, {128, 2}, {64, 16}, {64, 8}};
        flush_cm();
        const int64_t step = last_build_step == std::numeric_limits<int64_t>::min() ? 0 : last_build_step;
        hipEvent_t e0, e1;
        MHIP_HIP(hipEventCreate(&e0)); MHIP_HIP(hipEventCreate(&e1));
        int n = 0, best_bi = 0, best_js = 0; float best = 0;
        for (auto& c : cand) {
            if (c[0] * c[1] > MAX_THREADS || (int64_t)c[0] > std::max<int64_t>(n_owned, 64)) continue;
            bool seen = false;
            for (int k = 0; k < n && k < max_trials; ++k) seen |= trials[k].block_atoms == c[0] && trials[k].j_split == c[1];
            if (seen) continue;                                // an earlier candidate was shrunk to this shape
            float us = -1.f;
            try {
                user_bi = c[0]; user_js = c[1];
                choose_blocking(); stale = true; cur_dt = 0;
                ensure_built(step); pass_step = step;
                launch_pair_kernel(false);                     // with a dual list: the pruning pass
                if (prune_disp_exceeded) { after_forces(step); launch_pair_kernel(false); }
                launch_pair_kernel(false);
                MHIP_HIP(hipEventRecord(e0, stream));
                for (int k = 0; k < n_passes; ++k) launch_pair_kernel(false);
                MHIP_HIP(hipEventRecord(e1, stream));
                MHIP_HIP(hipEventSynchronize(e1));
                float ms = 0; MHIP_HIP(hipEventElapsedTime(&ms, e0, e1));
                us = ms * 1000.f / (float)n_passes;
            } catch (const ApiError& err) {
                if (err.code != MHIP_ERR_CAPACITY) { user_bi = user_js = 0; (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); throw; }
            }
            if (n < max_trials && trials) { trials[n].block_atoms = us < 0 ? c[0] : BI; trials[n].j_split = us < 0 ? c[1] : JS; trials[n].us_per_pass = us; }
            ++n;
            if (us > 0 && (best_bi == 0 || us < best)) { best = us; best_bi = BI; best_js = JS; }
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        user_bi = best_bi; user_js = best_js;                  // (0, 0 = automatic, when nothing could be timed)
        choose_blocking(); stale = true; frc_valid = false;
        return n;
    }

    void estimate_capacities() {
        n_blocks = cdiv(n_owned, BI);
        double vol = 1; for (int d = 0; d < 3; ++d) vol *= cfg.box[d];
        double rho = (double)n_tot / vol;
        if (G.no_list) { T_cap = (int)n_tot + 8; R_cap = cdiv(n_tot, 4) + 2; C_cap = 8; }
        else if (tri_mode && !tri_grid) { T_cap = (int)n_tot + 8; R_cap = (int)std::min<double>(1.5 * rho * 4.0 / 3.0 * M_PI * std::pow(cfg.r_list, 3) / 4 / JS + 8, n_tot / 4.0 + 2); C_cap = 8; }
        else {
            double r = (cfg.r_list + (dual ? outer_margin : 0.0)) * 1.001, a = std::cbrt(BI / rho);
            double v_tile = a * a * a + 6 * a * a * r + 3 * M_PI * a * r * r + 4.0 / 3.0 * M_PI * r * r * r;
            if (tri_grid) v_tile = (a + 2 * r) * (a + 2 * r) * (a + 2 * r) * 1.3;   // per-axis (Chebyshev) pruning in a skewed frame: a box, not a rounded one
            T_cap = (int)std::min<double>(1.4 * rho * v_tile + 64, (double)n_tot + 8);
            R_cap = (int)std::min<double>(1.5 * rho * 4.0 / 3.0 * M_PI * r * r * r / 4 / JS + 8, n_tot / 4.0 + 2);
            double cells = 1;
            for (int d = 0; d < 3; ++d) cells *= std::min<double>(G.nc[d], (1.3 * a + 2 * r) / (double)G.cs[d] + 2);
            C_cap = (int)std::min<double>(cells * 1.2 + 8, MAX_BOX_CELLS);
        }
        T_cap = std::max(T_cap, 16); R_cap = std::max(R_cap, 2); C_cap = std::max(C_cap, 8);
        T_cap = std::min((T_cap + 3) & ~3, slot_cap());   // multiple of 4: keeps the LDS carve-up 8-byte aligned
    }

    template <class K> void set_lds_limit(K kern, size_t bytes) {
        if (bytes > (size_t)MAX_LDS_BYTES) throw ApiError{MHIP_ERR_CAPACITY, "atom tile does not fit the 160 KiB LDS"};
        if (bytes > 64 * 1024) MHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    }

    const double* cm_src() const { return cm_ext ? cm_ext : (const double*)cm_step.p; }
    void flush_cm() {
        if (!cm_pending) return;
        hipLaunchKernelGGL(k_shift_vel<T>, dim3(std::min(cdiv(n_owned, 256), 1024)), dim3(256), 0, stream, n_owned, vel[cur].p, (const T*)vcm.p,
                           cm_pending == 2 ? cm_src() : (const double*)nullptr, n_cm_step);
        cm_pending = 0; cm_ext = nullptr;
    }

    const T* to_device(const void* host_or_dev, size_t count, int mem_kind, DBuf<T>& stage) {
        if (!host_or_dev) return nullptr;
        if (mem_kind == MHIP_MEM_DEVICE) return (const T*)host_or_dev;
        stage.reserve(count);
        MHIP_HIP(hipMemcpyAsync(stage.p, host_or_dev, count * sizeof(T), hipMemcpyHostToDevice, stream));
        return stage.p;
    }

    // ---------------------------------------------------------------------------------------------
    // the neighbour rebuild pipeline (≙ find_neighbors + the reorder/compress/tile-search stages of
    // ext/MollyCUDAExt.jl:845-873, redesigned O(N))
    // a search that does not fit the LDS with the outer radius falls back to the plain single-radius list
    void rebuild(int64_t step_n) {
        try { rebuild_impl(step_n); }
        catch (const ApiError& e) {
            if (e.code != MHIP_ERR_CAPACITY || !dual) throw;
            // dense small systems (a 64-atom block of water with a 1.4 nm shell is half of 6mrr): halve the outer margin before giving up on it
            // (a sub-domain of a multi-GPU run never shrinks its margin on its own: the ranks decide prunes and re-plans from the same
            // numbers, mhip_plan_decide — without the dual list mhip_plan_state_dev reports +inf and every rank re-plans at every rebuild step)
            if (n_ghost > 0 || ghost_margin > 0 || host_prune) dual_disabled = true;
            else if (outer_margin > 0.06) ++margin_halvings; else if (!margin_zero) margin_zero = true; else dual_disabled = true;
            if (debug_on) std::fprintf(stderr, "[mhip] dual list %s (capacity): %s\n", dual_disabled ? "off" : (margin_zero ? "without outer margin" : "margin halved"), e.msg.c_str());
            setup_grid(); choose_blocking(); stale = true;
            rebuild(step_n);
        }
    }

    void rebuild_impl(int64_t step_n) {
        next_check_step = -1;
        const int sub_bits = (ilog2(2 * G.ncell + 1) + 1 + 6 <= 32) ? 6 : 0;      // atoms of a cell ordered by a 6-bit Morton position inside it
        if (!params_set || !state_set) throw ApiError{MHIP_ERR_STATE, "set_atoms and set_state must be called before forces"};
        auto t0 = std::chrono::steady_clock::now();
        const int o = cur, n = 1 - cur;
        const int ncell2 = 2 * G.ncell + 1;
        prof.begin(3, stream);
        const int nb256 = cdiv(n_tot, 256);
        MHIP_HIP(hipMemsetAsync(cell_cnt.p, 0, (size_t)ncell2 * sizeof(int32_t), stream));
        tr("k_cell_keys");
        hipLaunchKernelGGL(k_cell_keys<T>, dim3(nb256), dim3(256), 0, stream, n_tot, n_owned, (const T4*)pos[o].p, (const int32_t*)inv.p,
                           (const uint32_t*)cell_rank.p, key_in.p, idx_in.p, cell_cnt.p, G, sub_bits);
        tr("sort_pairs");
        size_t tb = cub_tmp.n;
        MHIP_HIP(sort_pairs_u32(cub_tmp.p, tb, key_in.p, key_out.p, idx_in.p, perm.p, (int)n_tot, std::min(32, ilog2(2 * G.ncell + 1) + 1 + sub_bits), stream));
        tb = cub_tmp.n;
        MHIP_HIP(exclusive_sum_i32(cub_tmp.p, tb, cell_cnt.p, cell_start.p, ncell2, stream));
        tr("k_permute");
        hipLaunchKernelGGL(k_permute<T>, dim3(nb256), dim3(256), 0, stream, n_tot, (const int32_t*)perm.p, (const T4*)pos[o].p, (const T4*)vel[o].p,
                           (const T4*)frc[o].p, (const T2*)lj[o].p, (const int32_t*)orig[o].p, pos[n].p, vel[n].p, frc[n].p, lj[n].p, orig[n].p, inv.p);
        prof.end(3, stream);
        cur = n;
        if (n_ghost > 0 && has_exc) throw ApiError{MHIP_ERR_UNSUPPORTED, "exclusion lists with ghost atoms are not supported"};
        if (tri_mode && !tri_grid && n_tot + 8 > TILE_SLOT_MAX) throw ApiError{MHIP_ERR_UNSUPPORTED, "a TriclinicBoundary whose cell heights allow no cell grid (fewer than 6 cells of r_list/2 on every axis) is limited to 32 759 atoms"};

        for (int attempt = 0; attempt < 12; ++attempt) {
            n_blocks = cdiv(n_owned, BI);
            bool walk = !G.no_list && (!tri_mode || tri_grid);
            size_t lds = build_lds_bytes(T_cap, BI, C_cap, walk);
            if (walk && !tri_grid && lds > (size_t)MAX_LDS_BYTES) { walk = false; lds = build_lds_bytes(T_cap, BI, C_cap, false); }   // no room for the cell offsets: transposed search (its box tests are Cartesian: not on a triclinic grid)
            if (lds > (size_t)MAX_LDS_BYTES) {
                if (BI > 64) { BI /= 2; JS = std::min(JS * 2, MAX_THREADS / BI); estimate_capacities(); continue; }
                throw ApiError{MHIP_ERR_CAPACITY, "neighbourhood tile of one 64-atom block does not fit the 160 KiB LDS (r_list too large for this density): T_cap " + std::to_string(T_cap) +
                               " C_cap " + std::to_string(C_cap) + " bytes " + std::to_string(lds)};
            }
            tile_idx.reserve((size_t)n_blocks * T_cap); tile_cnt.reserve(n_blocks); wave_rows.reserve((size_t)n_blocks * JS * (BI / WAVE));
            nbr.reserve((size_t)n_blocks * JS * R_cap * BI); blk_center.reserve(n_blocks);
            MHIP_HIP(hipMemsetAsync(flags.p, 0, N_FLAGS * sizeof(int32_t), stream));
            BuildArgs<T> A;
            A.G = G; A.n_owned = n_owned; A.n_tot = n_tot; A.BI = BI; A.BI_shift = ilog2(BI); A.JS = JS; A.T_cap = T_cap; A.R_cap = R_cap; A.C_cap = C_cap;
            A.pos = pos[cur].p; A.orig = orig[cur].p; A.cell_start = cell_start.p; A.cell_rank = cell_rank.p;
            A.xl_start = has_exc ? xl_start.p : nullptr; A.xl_list = xl_list.p; A.xl_span = xl_span; A.X_cap = X_CAP;
            A.tile_idx = tile_idx.p; A.tile_cnt = tile_cnt.p; A.nbr = nbr.p; A.wave_rows = wave_rows.p; A.blk_center = blk_center.p; A.flags = flags.p;
            A.margin = G.no_list ? T(0) : G.r_list * T(1e-3);
            A.approx = dual ? 1 : 0;      // (an outer list is a candidate set: no exact band decisions)
            A.walk = walk ? 1 : 0;
            A.eshift = eshift = want_eshift();
            A.cnt_out = nullptr; cnt_outer_valid = false;
            A.dbg = stamps_begin((size_t)n_blocks * 16 * 8);
            if (gs_groups() > 0 && adopt_env) { cnt_outer.reserve((size_t)n_blocks * JS * BI); A.cnt_out = cnt_outer.p; cnt_outer_valid = true; }
            // (k_build<T, true, false, true> — the walk with BOTH the exact band decisions and the exception lookups compiled in, the single exact list of a system with
            // exclusions — was kept to 64-atom blocks from the end of round 5: at 128- and 256-atom blocks every i-wave but the first had come out with an empty list
            // (tools/micro/xl_waves.py: 2.86 M of 5.70 M pairs of a 46 656-atom charged fluid).  Round 6 bisected it: commit e502d9c shows the defect with the detour
            // switched off, its successor 72454c0 does not.  That commit — made for speed, behind the detour — took the staged atom's local frame out of two private ARRAYS
            // written through a pointer chosen at run time (the two branches' stores merged) and living in scratch; in this one instantiation, under the 128-VGPR bound with
            // 104 SGPR spills, the arrays of the waves behind the first were not addressed where they were read back (an i-atom with garbage local coordinates finds no
            // neighbour).  With the frame in six scalars the variant is right at every shape — fp32 and fp64, 128 x 4 and 256 x 2, cubic and triclinic grids, pair SET and
            // special flags against the oracle — and the detour is gone; tests/test_gpu_parity.py::test_single_pair_list_with_128_and_256_atom_blocks[charged-*] and
            // tests/test_gpu_triclinic.py::test_triclinic_single_list_with_exceptions_beyond_64_atom_blocks run it at those shapes.)
            prof.begin(1, stream);
            tr("k_build");
            auto go = [&](auto kern) { set_lds_limit(kern, lds); hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(BI * JS), lds, stream, A); };
            if (A.walk && !A.xl_start) { if (A.approx) go(k_build<T, true, true, false>); else go(k_build<T, true, false, false>); }       // no exception lists: their lookups are not compiled
            else if (A.walk) { if (A.approx) go(k_build<T, true, true>); else go(k_build<T, true, false>); }
            else { if (A.approx) go(k_build<T, false, true>); else go(k_build<T, false, false>); }
            hipLaunchKernelGGL(k_build_summary, dim3(std::min(64, cdiv(n_blocks * JS * (BI / WAVE), 256))), dim3(256), 0, stream, n_blocks, n_blocks * JS * (BI / WAVE), R_cap, (const int32_t*)tile_cnt.p, wave_rows.p, (const float*)nullptr, flags.p);
            prof.end(1, stream);
            MHIP_HIP(hipGetLastError());
            MHIP_HIP(hipMemcpyAsync(h_flags, flags.p, N_FLAGS * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            MHIP_HIP(hipStreamSynchronize(stream));
            if (A.dbg) stamps_dump(".build", (size_t)n_blocks * 16 * 8);      // (stamp builds: tools/build_times.py)
            int ovf = h_flags[FLAG_OVERFLOW];
            if (!ovf) break;
            if (ovf & OVF_SLOT) {
                if (eshift) { scaled_entries_off = true; estimate_capacities(); continue; }
                if (BI > 64) { BI /= 2; JS = std::min(JS * 2, MAX_THREADS / BI); estimate_capacities(); continue; }
                throw ApiError{MHIP_ERR_CAPACITY, "more atoms within r_list of one 64-atom block than a 16-bit list entry can name"};
            }
            if (ovf & OVF_BOXCELLS) {
                if (h_flags[FLAG_MAX_CELLS] <= MAX_BOX_CELLS) { C_cap = std::min<int>(MAX_BOX_CELLS, (int)(h_flags[FLAG_MAX_CELLS] * 1.1) + 8); continue; }
                if (BI > 64) { BI /= 2; JS = std::min(JS * 2, MAX_THREADS / BI); estimate_capacities(); continue; }
                throw ApiError{MHIP_ERR_CAPACITY, "block neighbourhood spans more than 8192 cells"};
            }
            if (ovf & OVF_TILE) T_cap = std::min<int>(slot_cap(), (((int)(h_flags[FLAG_MAX_TILE] * 1.15) + 32) + 3) & ~3);
            if (ovf & OVF_ROWS) R_cap = (int)(h_flags[FLAG_MAX_ROWS] * 1.2) + 4;
            if (attempt == 11) throw ApiError{MHIP_ERR_CAPACITY, "neighbour structures did not converge"};
        }
        minimg = h_flags[FLAG_MINIMG] != 0;
        max_tile = h_flags[FLAG_MAX_TILE]; max_rows = h_flags[FLAG_MAX_ROWS]; total_rows = outer_rows = h_flags[FLAG_TOTAL_ROWS];
        carve_force_lds(max_tile);
        red_part.reserve(std::max<size_t>(7 * (size_t)n_blocks, 4 * (size_t)cdiv(n_owned, 256)) + 8);
        bonded.on_reorder();
        ++n_outer; last_outer_step = step_n;
        if (dual) {   // remember where everybody was; the next force pass prunes the outer list into the inner one
            pos_snap.reserve(cap);
            MHIP_HIP(hipMemcpyAsync(pos_snap.p, pos[cur].p, (size_t)n_tot * sizeof(T4), hipMemcpyDeviceToDevice, stream));
            inner_valid = false; inner_is_outer = false; prune_disp_exceeded = false; ghost_flags_in_ok = false;
            if (cur_dt > 0 && skin_in < skin && !host_prune) {   // inside a run: how fast is the fastest atom? (sizes the inner skin before the first prune)
                (void)max_disp2_since(pos_snap);
                adapt_inner_skin(drift_ahead(0.0, 1, cfg.rebuild_every > 0 ? cfg.rebuild_every : 10));
            }
        }
        if (lazy_single) {
            pos_snap_in.reserve(cap);
            MHIP_HIP(hipMemcpyAsync(pos_snap_in.p, pos[cur].p, (size_t)n_tot * sizeof(T4), hipMemcpyDeviceToDevice, stream));
            last_prune_step = step_n;
        }
        stale = false; coords_moved = false; export_needs_search = false; ghost_flags_ok = false; ghost_flags_in_ok = false; last_build_step = step_n; ++n_rebuilds;
        frc_run_total = false;   // the sort moved the atoms
        hx.plan_ok = hx.tile_ok = false; hx.trk_step = -1;
        last_rebuild_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }

    // LDS carve-up of the force kernel: the whole tile if it fits the budget, else segments of the budget's size
    void carve_force_lds(int tile_max, size_t reserve = 0) {   // reserve: bytes the caller places behind the carve-up (prune pass: marks + scan scratch)
        size_t budget = std::min<size_t>((size_t)env_int("MOLLYHIP_LDS_BUDGET_KB", 160) * 1024, MAX_LDS_BYTES);
        if (budget > reserve + 8192) budget -= reserve;
        const bool per_atom_lj = (ljm == LJ_DIST || ljm == LJ_GENERIC);
        const size_t per_atom = sizeof(T4) + (per_atom_lj ? sizeof(T2) : 0);
        const size_t fixed = std::max((size_t)JS * 4 * BI * sizeof(T), (size_t)BI * sizeof(double)) + 64;
        int fit = (int)((budget > fixed + 2 * per_atom ? budget - fixed : 2 * per_atom) / per_atom) - 1;
        fit = std::max(fit & ~1, 2);
        segmented = tile_max > fit;
        tile_lds = segmented ? fit : std::max(tile_max, 1);
        lds_force = force_lds_bytes(tile_lds);
        if (lds_force > (size_t)MAX_LDS_BYTES) throw ApiError{MHIP_ERR_CAPACITY, "force-kernel LDS carve-up exceeds 160 KiB"};
    }

    // the exact list of radius r_list at the current coordinates for mhip_export_neighbors: the outer list filtered with the reference's predicate into its
    // OWN arrays (nbr_x, tile_idx_x: the inner list of the force passes keeps referring to tile_idx_in), + max displacement since the outer build.
    // (The prune of the dual scheme as a kernel of this kind followed by a plain pass measured slower than pruning inside the pass: 0.72 + 0.09 against 0.58 ms.)
    void launch_filter() {
        rows_x.reserve((size_t)n_blocks * JS * (BI / WAVE)); nbr_x.reserve((size_t)n_blocks * JS * R_cap * BI);
        tile_idx_x.reserve((size_t)n_blocks * T_cap); tile_cnt_x.reserve(n_blocks);
        MHIP_HIP(hipMemsetAsync(flags.p, 0, N_FLAGS * sizeof(int32_t), stream));
        blk_disp2.reserve(n_blocks);
        FilterArgs<T> F;
        F.G = G; F.n_owned = n_owned; F.BI = BI; F.BI_shift = ilog2(BI); F.JS = JS; F.T_cap = T_cap; F.R_cap = R_cap; F.n_blocks = n_blocks;
        F.pos = pos[cur].p; F.pos_snap = pos_snap.p; F.tile_idx = tile_idx.p; F.tile_cnt = tile_cnt.p; F.nbr_out = nbr.p; F.rows_out = wave_rows.p;
        F.nbr_in = nbr_x.p; F.rows_in = rows_x.p; F.tile_idx_in = tile_idx_x.p; F.tile_cnt_in = tile_cnt_x.p; F.blk_center = blk_center.p; F.blk_disp2 = blk_disp2.p; F.flags = flags.p;
        F.eshift = eshift;
        F.r_in = r_in; F.r_in2 = r_in2; F.exact_all = minimg ? 1 : 0;
        F.T_lds = std::min<int>(max_tile, (MAX_LDS_BYTES - 256) / (int)sizeof(float4));
        F.T_lds = minimg ? 0 : F.T_lds;
        size_t lds = (size_t)F.T_lds * sizeof(float4) + (size_t)((T_cap + 8) & ~7) + (size_t)((T_cap + 2) & ~1) * 2 + (size_t)BI * JS * 4 + 64;
        set_lds_limit(k_filter<T>, lds);
        prof.begin(4, stream);
        tr("k_filter");
        hipLaunchKernelGGL(k_filter<T>, dim3(n_blocks), dim3(BI * JS), lds, stream, F);
        hipLaunchKernelGGL(k_build_summary, dim3(std::min(64, cdiv(n_blocks * JS * (BI / WAVE), 256))), dim3(256), 0, stream, n_blocks, n_blocks * JS * (BI / WAVE), R_cap,
                           (const int32_t*)tile_cnt_x.p, rows_x.p, (const float*)blk_disp2.p, flags.p);
        prof.end(4, stream);
        MHIP_HIP(hipGetLastError());
    }

    // max |x − x_snap|² over all local atoms, and the largest speed among the owned atoms (one small kernel + one host sync)
    double last_vmax = 0, prev_vmax = 0;
    float max_disp2_since(const DBuf<T4>& snap) {
        tr("k_max_disp");
        MHIP_HIP(hipMemsetAsync(flags.p + FLAG_MAX_DISP2, 0, 2 * sizeof(int32_t), stream));
        hipLaunchKernelGGL(k_max_disp<T>, dim3(std::min(cdiv(n_tot, 1024), 512)), dim3(1024), 0, stream, n_tot, (const T4*)pos[cur].p, (const T4*)snap.p,   // (every block ends in two atomics on the same two words: 4096 blocks of 256 lanes took 94 us at 1M atoms, 1024 blocks 30 us)
                           reinterpret_cast<unsigned int*>(flags.p + FLAG_MAX_DISP2), G, (const T4*)vel[cur].p, n_owned, reinterpret_cast<unsigned int*>(flags.p + FLAG_MAX_V2));
        MHIP_HIP(hipMemcpyAsync(h_flags, flags.p, N_FLAGS * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        float d2, v2; std::memcpy(&d2, &h_flags[FLAG_MAX_DISP2], sizeof(float)); std::memcpy(&v2, &h_flags[FLAG_MAX_V2], sizeof(float));
        prev_vmax = last_vmax; last_vmax = std::sqrt((double)v2);
        ++n_disp_checks;
        return d2;
    }
    // Upper estimate of how much further anybody gets until the next displacement check, `every` steps from now.  Inside a run the
    // time step is known: the fastest atom's speed now, stretched by how much the top speed grew since the last check (at least 10 %),
    // times the interval.  Driven from outside through forces(step_n) there is no time step: the displacement rate seen so far, times 1.5.
    double cur_dt = 0;
    // The inner list must outlive at least one check interval: if the fastest atoms cover more than a third of the inner skin between
    // two checks, the skin grows (up to the reference's own r_list − cutoff) and the list is pruned afresh with the larger radius.
    void adapt_inner_skin(double drift_per_interval) {
        if (!dual || (host_prune && !engine_sched) || !(skin_in < skin) || inner_skin_fixed) return;
        const double need = std::min(skin, 3.0 * drift_per_interval / 0.98);
        if (need <= skin_in) return;
        skin_in = need; skin_in_adapted = need;
        const T rp = T(rc_max_ + skin_in);
        r_prune2 = (skin_in < skin) ? rp * rp : r_in2;
        inner_valid = false;
        // An outer list serves a second prune only while nobody moved (outer_margin + skin − skin_in)/2 since its search, and the inner
        // list is not due before ≈ skin_in/2: with outer_margin <= 2·skin_in − skin every outer list is pruned exactly once, and its
        // margin only makes the search dearer.
        if (n_ghost == 0 && outer_margin > 0 && outer_margin <= 2.0 * skin_in - skin + 0.02) want_margin_zero = true;
        if (debug_on) std::fprintf(stderr, "[mhip] inner skin raised to %.3f nm (drift per check interval %.4f nm)\n", skin_in, drift_per_interval);
    }
    double drift_ahead(double d_so_far, int64_t steps_so_far, int every) const {
        const double empirical = 1.5 * d_so_far * (double)every / (double)std::max<int64_t>(steps_so_far, 1);
        if (!(cur_dt > 0)) return empirical;
        const double growth = prev_vmax > 0 ? std::min(std::max(last_vmax / prev_vmax, 1.1), 3.0) : 1.25;
        return last_vmax * growth * cur_dt * every;
    }

    // Validity checks without a kernel or a pipeline drain of their own (mhip_vv_run's fused loop, dual list, no ghosts).  The integrator
    // launch that makes the coordinates of a check step s also takes the maxima the check needs — |x − snapshot|² against the inner
    // and the outer list's snapshot, |v|² — per block; one tiny launch reduces them into pinned memory behind an event.  The host
    // reads them at step s + 1, when they have long arrived, and applies the decision there: the pass of step s was covered by the
    // previous decision's horizon (it reached up to the check step), a prune or a search that the measurement asks for happens at
    // s + 1 (with that step of headroom in the outer-list test).  (Taking the maxima inside k_forces instead cost its packed loop
    // 14 % through a different register assignment, with the same instructions: measured, dropped.)
    DBuf<float> trk_part, trk_out; float* h_trk = nullptr; hipEvent_t ev_trk = nullptr;
    bool trk_issued = false; int64_t trk_step = -1, trk_prune_id = -1, trk_outer_id = -1; double trk_prev_vmax = 0;   // (ids: the running counts of prunes / outer searches)
    bool in_vv_fused = false;
    bool in_lang_fused = false;      // inside mhip_langevin_run of a small system whose last force launch integrates (the pair launch's extra workgroup then sums the Σ m v partials, as inside mhip_vv_run)
    bool in_lang_async = false;      // … and whose list checks are measured by that launch (no Andersen coupling behind it: the speeds the check reads would not be the run's)
    bool async_ok() const { return (in_vv_fused || in_lang_async) && dual && n_ghost == 0 && !host_prune && inner_valid && !stale; }
    void resolve_track(int64_t step) {
        if (!trk_issued) return;
        MHIP_HIP(hipEventSynchronize(ev_trk));
        trk_issued = false;
        const double d = std::sqrt((double)h_trk[0]), d_outer = std::sqrt((double)h_trk[1]);
        prev_vmax = trk_prev_vmax; last_vmax = std::sqrt((double)h_trk[2]);   // (the speed of the check before, as it was when this one was issued: a run cut into chunks decides alike)
        ++n_disp_checks;
        if (!dual || stale || !inner_valid || n_ghost > 0 || n_filters != trk_prune_id || n_outer != trk_outer_id) return;   // the lists it measured have been replaced meanwhile
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        const int64_t so_far = trk_step - last_prune_step;
        const double ahead = drift_ahead(d, so_far, every);
        if (debug_on) std::fprintf(stderr, "[mhip] step %lld: measured at %lld: d %.5f d_outer %.5f v_max %.4f\n", (long long)step, (long long)trk_step, d, d_outer, last_vmax);
        adapt_inner_skin(ahead);
        bool reprune = !inner_valid || 2.0 * (d + ahead) > skin_in * 0.98;
        next_check_step = -1;
        if (reprune && inner_valid)
            if (const int k = steps_within(d, 0.49 * skin_in, so_far, every, true)) { next_check_step = trk_step + k; reprune = false; }
        if (!reprune) return;
        if (2.0 * (d_outer + last_vmax * cur_dt * 1.25 * (double)(step - trk_step)) > prune_margin() * 0.98) { rebuild(step); return; }
        inner_valid = false;
    }

    // Checks between the cadence steps.  A list that cannot be vouched for over a whole interval (the fastest atom could use up the
    // remaining slack in `every` steps) may still be good for k < every steps: instead of giving it up now, look again in k steps.
    // Light, fast atoms (hydrogens at 0.5 fs: 0.06 nm of possible drift per 10 steps against 0.1 nm of slack) otherwise cost a
    // search at nearly every interval.  Only inside mhip_vv_run / mhip_langevin_run, which own the step loop.
    int64_t next_check_step = -1;
    bool check_due(int64_t step, int every) const { return step % every == 0 || (next_check_step >= 0 && step >= next_check_step); }
    // largest k < every such that a displacement of d now stays within `limit` for k more steps (0: none worth a check of its own)
    int steps_within(double d, double limit, int64_t steps_so_far, int every, bool caller_owns_loop = false) const {
        if (!(in_run || caller_owns_loop)) return 0;
        const double per_step = drift_ahead(d, steps_so_far, 1);
        if (!(per_step > 0)) return 0;
        const int k = (int)std::min<double>(std::floor((limit - d) / per_step), every - 1);
        return k >= 3 ? k : 0;
    }
    bool in_run = false;
    struct InRun { bool& f; explicit InRun(bool& b) : f(b) { f = true; } ~InRun() { f = false; } };

    // rebuild step of the cadence (find_neighbors at step_n % n_steps == 0): a fresh search, or — with the dual list —
    // a filter pass, falling back to the search when it is due or an atom moved more than half the margin
    void refresh(int64_t step_n) {
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        if (want_margin_zero && !margin_zero && n_ghost == 0) {
            if (debug_on) std::fprintf(stderr, "[mhip] outer margin dropped (inner skin %.3f nm leaves it no second prune)\n", skin_in);
            margin_zero = true; setup_grid(); choose_blocking(); stale = true;
        }
        if (!dual && lazy_single && !stale && step_n > last_prune_step) {
            // single list built with r_list at step last_prune_step: it still holds every pair within the cutoffs unless somebody moved skin/2
            const double d = std::sqrt((double)max_disp2_since(pos_snap_in));
            if (debug_on) std::fprintf(stderr, "[mhip] step %lld: max disp %.5f nm since the build of step %lld (skin %.3f), v_max %.4f\n", (long long)step_n, d, (long long)last_prune_step, skin, last_vmax);
            next_check_step = -1;
            if (2.0 * (d + drift_ahead(d, step_n - last_prune_step, every)) <= skin * 0.98) { last_build_step = step_n; ++n_skipped; return; }
            if (const int k = steps_within(d, 0.49 * skin, step_n - last_prune_step, every)) { next_check_step = step_n + k; last_build_step = step_n; ++n_skipped; return; }
        }
        if (!dual || stale || (n_ghost == 0 && step_n < last_outer_step)) { rebuild(step_n); return; }
        // The inner list (pairs within r_list when it was pruned) provably contains every pair within the cutoffs as long as no atom
        // moved more than skin/2 since then — the condition the reference's fixed cadence only assumes.  Check it; re-prune (inside
        // the next force pass) only when it is about to fail.  mhip_export_neighbors always returns the exact list of NOW.
        if (host_prune && inner_valid) { last_build_step = step_n; ++n_rebuilds; return; }   // the host calls mhip_request_prune
        bool reprune = !inner_valid;
        if (!reprune && trk_issued && trk_step == step_n && async_ok()) {   // measured by the integrator launch that made these coordinates; decided one step later
            last_build_step = step_n; ++n_rebuilds;
            return;
        }
        if (!reprune) {
            const float d2 = max_disp2_since(pos_snap_in);
            // headroom for the drift until the next check: the displacement so far, extrapolated one more interval
            const double d = std::sqrt((double)d2), ahead = drift_ahead(d, step_n - last_prune_step, every);
            adapt_inner_skin(ahead);
            reprune = !inner_valid || 2.0 * (d + ahead) > skin_in * 0.98;
            next_check_step = -1;
            if (reprune && inner_valid && n_ghost == 0)
                if (const int k = steps_within(d, 0.49 * skin_in, step_n - last_prune_step, every)) { next_check_step = step_n + k; reprune = false; }
        }
        if (reprune && n_ghost == 0 && !stale) {
            // a prune is only as good as the outer list behind it (nobody moved more than half the margin since the outer search):
            // if that is already used up, search again now instead of running a prune pass that would have to be thrown away
            const double d_outer = std::sqrt((double)max_disp2_since(pos_snap));
            if (2.0 * d_outer > prune_margin() * 0.98) { rebuild(step_n); return; }
        }
        if (reprune) inner_valid = false;         // the next force pass re-prunes the outer list at the then-current coordinates
        last_build_step = step_n; ++n_rebuilds;
    }

    // after a force pass that pruned: if an atom outran half the margin the outer list can no longer vouch for the inner one
    void after_forces(int64_t step_n) {
        if (!prune_disp_exceeded) return;
        prune_disp_exceeded = false;
        if (n_ghost > 0) throw ApiError{MHIP_ERR_STATE, "an atom moved more than half the ghost margin since the ghost plan: re-plan earlier (mhip_plan_disp2_dev)"};
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        // If that keeps happening before the outer list has paid for itself (fast light atoms, small time step), the dual list
        // is a loss: fall back to a fresh search at every rebuild step.
        if (step_n - last_outer_step <= 2 * (int64_t)every) { if (++early_outer >= 3) { if (debug_on) std::fprintf(stderr, "[mhip] dual list off (outer list outrun 3x)\n"); dual_disabled = true; setup_grid(); choose_blocking(); stale = true; } }
        else early_outer = 0;
        rebuild(step_n);
    }

    // Neighbours at the first step of a run (find_neighbors with current_neighbors = nothing, simulators.jl:564).  The reference
    // searches afresh there; a fresh search changes results only where list MEMBERSHIP matters (an interaction without a cutoff
    // inside r_list).  Lists that carry a skin are kept if the displacement checks say they still cover every cutoff sphere: a run
    // continued in chunks then walks the same lists in the same order as the uncut run and reproduces it bit for bit.
    bool vel_check_due = false;   // velocities were replaced since the last validity check of the lists
    void start_lists(int64_t first_step) {
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        lists_after_set_state();
        if (stale || !(dual || lazy_single)) { vel_check_due = false; rebuild(first_step); return; }
        if ((check_due(first_step, every) && first_step != last_build_step) || vel_check_due) { vel_check_due = false; refresh(first_step); }
    }

    void ensure_built(int64_t step_n) {
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        resolve_track(step_n);
        lists_after_set_state();
        vel_check_due = false;   // (driven from outside there is no time step: the displacement checks of set_state are all there is)
        if (stale) rebuild(step_n);
        else if (check_due(step_n, every) && step_n != last_build_step) refresh(step_n);
    }

    // What a pruning pass leaves for the host — the pruned list's largest tile and row total, the largest displacement since the outer search — read WITHOUT draining
    // the stream inside mhip_domain_run on a ghosted sub-domain (the drain sat at exactly the step where every rank prunes, and each rank's device
    // idled while its host woke up and queued the rest of the step): the summary kernel writes into pinned words of their own behind an event; until the host finds the
    // event complete the next passes are shaped for the OUTER list's largest tile (a pruned tile is a subset of it: same results, a larger LDS carve-up for a step or
    // two), and a displacement beyond the ghost margin — which fails the run on this path anyway (after_forces) — is reported when it is read, a step or two later.
    int32_t* h_prune = nullptr; hipEvent_t ev_prune = nullptr; bool prune_pending = false; int64_t prune_pending_id = -1;
    void prune_resolve(bool block) {
        if (!prune_pending) return;
        if (!block && hipEventQuery(ev_prune) != hipSuccess) { (void)hipGetLastError(); return; }
        MHIP_HIP(hipEventSynchronize(ev_prune));
        prune_pending = false;
        if (n_filters != prune_pending_id || stale || !inner_valid) return;      // the list it described has been replaced meanwhile
        float d2; std::memcpy(&d2, &h_prune[FLAG_MAX_DISP2], sizeof(float));
        total_rows = h_prune[FLAG_TOTAL_ROWS]; max_tile_in = h_prune[FLAG_MAX_TILE];
        if (debug_on) std::fprintf(stderr, "[mhip] prune (read late): max disp %.5f nm (margin %.3f) rows %lld tile %d\n", std::sqrt((double)d2), prune_margin(), (long long)total_rows, max_tile_in);
        if (2.0 * std::sqrt((double)d2) > prune_margin() * 0.98)
            throw ApiError{MHIP_ERR_STATE, "an atom moved more than half the ghost margin since the ghost plan: re-plan earlier (mhip_plan_disp2_dev)"};
    }

    // ---------------------------------------------------------------------------------------------
    // the pair-kernel variants are compiled in forces_inst.hip (one translation unit per precision and Coulomb kind)
    void launch_forces_any(const ForceArgs<T>& A, bool energy) {
        const bool prune = A.nbr_dst != nullptr && !energy;
        const unsigned threads = (unsigned)(BI * JS);
        if (lds_force > (size_t)MAX_LDS_BYTES) throw ApiError{MHIP_ERR_CAPACITY, "atom tile does not fit the 160 KiB LDS"};
        switch (coulm) {
        case MHIP_COUL_NONE: launch_forces_tc<T, MHIP_COUL_NONE>(A, ljm, energy, minimg, segmented, prune, lds_force, threads, stream); break;
        case MHIP_COUL_PLAIN: launch_forces_tc<T, MHIP_COUL_PLAIN>(A, ljm, energy, minimg, segmented, prune, lds_force, threads, stream); break;
        case MHIP_COUL_REACTION_FIELD: launch_forces_tc<T, MHIP_COUL_REACTION_FIELD>(A, ljm, energy, minimg, segmented, prune, lds_force, threads, stream); break;
        default:
            if (I.approx_erfc) launch_forces_tc<T, MHIP_COUL_EWALD_DIRECT>(A, ljm, energy, minimg, segmented, prune, lds_force, threads, stream);
            else launch_forces_tc<T, COUL_EWALD_EXACT>(A, ljm, energy, minimg, segmented, prune, lds_force, threads, stream);
            break;
        }
    }
    // pairwise forces of the current coordinates into frc[cur] (overwrites); energy → red_part[0..n_blocks)
    void launch_pair_kernel(bool energy, int part = 0, bool allow_gs = false) {
        gs_used = false;
        prune_resolve(!(xf_direct && n_ghost > 0));      // figures a pruning pass left behind an event: taken if they have arrived (outside the ghosted run loop: waited for)
        ForceArgs<T> A;
        A.G = G; A.I = I; A.n_owned = n_owned; A.BI = BI; A.BI_shift = ilog2(BI); A.JS = JS; A.T_cap = T_cap; A.T_lds = tile_lds; A.R_cap = R_cap;
        A.n_blocks = n_blocks; A.blocks_per_xcd = cdiv(n_blocks, 8);
        A.orig = nullptr; A.S = StochP<T>{};      // (LANG variants only)
        A.pos = pos[cur].p; A.lj = lj[cur].p; A.tile_idx = tile_idx.p; A.tile_cnt = tile_cnt.p;
        // dual pair list: a force pass whose inner list is stale walks the OUTER list (always a valid superset — the cutoff is
        // applied per pair) and, if it is a plain force call, prunes it into the inner list on the way
        if constexpr (std::is_same<T, float>::value) {
            if (dual && !inner_valid && !energy && allow_gs && part == 0 && !frc_override && adopt_env && gs_groups() > 0 && cnt_outer_valid && margin_zero && !(skin_in < skin)
                && last_outer_step == pass_step && n_ghost == 0 && !host_prune) adopt_outer_list();
        }
        const bool use_inner = dual && inner_valid;
        const bool prune = dual && !inner_valid && !energy;
        const int GS = gs_groups();
        // a plain pass over an inner list that has its group-split form (made right behind the prune that wrote it)
        if constexpr (std::is_same<T, float>::value) {
            if (allow_gs && GS > 0 && use_inner && !energy && part == 0 && gs_list_id == n_filters && !frc_override) {
                const int q_lds = (max_tile_in + GS - 1) / GS + 1;      // (max_tile_in = max_tile while the outer list stands in)
                if (gs_lds_bytes(q_lds, BI, JS / GS) <= (size_t)MAX_LDS_BYTES / GS) {
                    frc_parts.reserve((size_t)(GS - 1) * cap);
                    GsArgs Z;
                    Z.G = G; Z.I = I; Z.n_owned = n_owned; Z.BI = BI; Z.BI_shift = ilog2(BI); Z.JS = JS; Z.GS = GS; Z.lgGS = ilog2(GS); Z.R_cap = GS * R_cap;      // (the group-split list's own row capacity: RegroupArgs::R_cap_dst)
                     Z.T_cap = T_cap; Z.Q_lds = q_lds;
                    Z.n_blocks = n_blocks; Z.spread = std::max(1, n_blocks / GS + 5);
                    Z.pos = pos[cur].p; Z.lj = lj[cur].p; Z.tile_idx = inner_is_outer ? tile_idx.p : tile_idx_in.p; Z.tile_cnt = inner_is_outer ? tile_cnt.p : tile_cnt_in.p; Z.nbr = nbr_gs.p; Z.wave_rows = rows_gs.p; Z.blk_center = blk_center.p;
                    Z.frc = frc[cur].p; Z.parts = frc_parts.p; Z.part_stride = cap;
                    Z.item_of = gs_item.p;
                    Z.dbg = stamps_begin((size_t)n_blocks * GS * 4 * 8);
                    last_pass_tile = max_tile_in;
                    prof.begin(0, stream);
                    tr("k_forces_gs");
                    spread_fused = false;
                    if (fuse_spread_next || fuse_terms_next) {      // … with the charge spreading (PME) and the bonded terms of the step as further workgroups of the same launch
                        bonded.ensure_roles(stream, cap);
                        const bool with_spread = fuse_spread_next;
                        const int order = with_spread ? pme.order : 5;
                        // inside vv_run: the Σ m v partials of the launch before become one partial in an extra workgroup of this launch (the step's last launch reads four words)
                        const bool gs_cm_fin = (in_vv_fused || in_lang_fused) && !energy && cm_pending == 2 && n_cm_step > 1 && n_cm_step <= 65536;
                        if (gs_cm_fin) cm_fin_buf.reserve(4);
                        const size_t lds = (with_spread ? std::max(gs_lds_bytes(q_lds, BI, JS / GS), std::min<size_t>((size_t)MAX_LDS_BYTES / GS, spread_head_bytes_f32(order) + (size_t)PME_BOX_BYTES)) : gs_lds_bytes(q_lds, BI, JS / GS)) & ~(size_t)15;
                        const int n_spread = with_spread ? (int)std::min<int64_t>(cdiv(n_owned, (int64_t)64), 4096) : 0;
                        launch_pair_spread_bonded(Z, n_blocks * GS, coulm, minimg, order, n_owned, reinterpret_cast<float*>(pme.rgrid.p), reinterpret_cast<const PmeP<float>&>(pme.P), n_spread,
                                                  reinterpret_cast<const BondedArgs<float>&>(static_cast<const BondedArgs<T>&>(bonded.slot_args(G, I, pos[cur].p, inv.p))), cdiv(bonded.n_blocks(), 4), lds, stream,
                                                  gs_cm_fin ? cm_src() : (const double*)nullptr, n_cm_step, gs_cm_fin ? cm_fin_buf.p : (double*)nullptr);
                        if (gs_cm_fin) { cm_ext = cm_fin_buf.p; n_cm_step = 1; }
                        spread_fused = with_spread; terms_fused = !with_spread;
                    } else launch_forces_gs(Z, coulm, minimg, stream);
                    fuse_spread_next = fuse_terms_next = false;
                    prof.end(0, stream);
                    MHIP_HIP(hipGetLastError());
                    ++n_force_calls; ++n_gs_passes; gs_used = true;
                    if (Z.dbg && (n_gs_passes % stamps_every()) == 0) stamps_dump("", (size_t)n_blocks * GS * 4 * 8);      // (stamp builds: tools/gs_times.py)
                    return;
                }
            }
        }
        const size_t prune_extra = prune ? prune_lds_bytes(std::min(T_cap, max_tile + 1), BI * JS) + 16 : 0;   // a tile that fills the LDS is segmented a little earlier
        carve_force_lds(use_inner ? max_tile_in : max_tile, prune_extra);
        last_pass_tile = use_inner ? max_tile_in : max_tile;
        A.T_lds = tile_lds;
        if (use_inner && !inner_is_outer) { A.tile_idx = tile_idx_in.p; A.tile_cnt = tile_cnt_in.p; }
        A.nbr = (use_inner && !inner_is_outer) ? nbr_in.p : nbr.p; A.wave_rows = (use_inner && !inner_is_outer) ? wave_rows_in.p : wave_rows.p;
        A.nbr_dst = nullptr; A.rows_dst = nullptr; A.pos_snap = nullptr; A.blk_disp2 = nullptr; A.r_prune2 = r_prune2;
        A.tile_idx_dst = nullptr; A.tile_cnt_dst = nullptr; A.mark_off = 0; A.snap_dst = nullptr; A.any_special = n_special > 0 ? 1 : 0; A.eshift = eshift;
        A.cnt_dst = nullptr;
        // the packed fp32 one-type loop keeps the tile as three arrays SOA_STRIDE dwords apart
        const bool fast_f32 = std::is_same<T, float>::value && ljm == LJ_DIST_UNIFORM && coulm == MHIP_COUL_NONE && !energy && !minimg && !segmented && n_special == 0;
        A.soa = 0;
        if (fast_f32 && eshift == ESHIFT_SCALED && I.lj_c12 != T(0))
            for (int k = 2; k >= 0; --k) if ((use_inner ? max_tile_in : max_tile) + 1 < SOA_STRIDES[k]) A.soa = SOA_STRIDES[k];   // the smallest stride that holds tile + sentinel
        if (A.soa) lds_force = std::max((size_t)3 * A.soa * sizeof(float) + 64, (size_t)JS * 4 * BI * sizeof(T) + 32);   // x[], y[], z[] instead of the generic 16-byte records
        A.part = 0; A.blk_ghost = nullptr;
        if (part != 0 && !prune) {   // blocks without / with ghost atoms in their tile (flags of the tile this pass stages)
            DBuf<int32_t>& fl = use_inner ? blk_ghost_in : blk_ghost; bool& ok = use_inner ? ghost_flags_in_ok : ghost_flags_ok;
            if (!ok) {
                fl.reserve(n_blocks);
                hipLaunchKernelGGL(k_block_ghost_flags, dim3(n_blocks), dim3(WAVE), 0, stream, n_blocks, T_cap, n_owned, (const int32_t*)A.tile_idx, (const int32_t*)A.tile_cnt, fl.p);
                ok = true;
            }
            A.part = part; A.blk_ghost = fl.p;
        }
        if (prune) {
            wave_rows_in.reserve((size_t)n_blocks * JS * (BI / WAVE)); nbr_in.reserve((size_t)n_blocks * JS * R_cap * BI); blk_disp2.reserve((size_t)n_blocks * (BI / WAVE));
            A.nbr_dst = nbr_in.p; A.rows_dst = wave_rows_in.p; A.pos_snap = pos_snap.p; A.blk_disp2 = blk_disp2.p;
            pos_snap_in.reserve(cap);
            // the snapshot the next displacement checks compare with: the owned atoms' by the kernel itself, ghosts by a copy
            A.snap_dst = pos_snap_in.p;
            if (n_ghost > 0) MHIP_HIP(hipMemcpyAsync(pos_snap_in.p + n_owned, pos[cur].p + n_owned, (size_t)n_ghost * sizeof(T4), hipMemcpyDeviceToDevice, stream));
            last_prune_step = pass_step;
            tile_idx_in.reserve((size_t)n_blocks * T_cap); tile_cnt_in.reserve(n_blocks);
            A.tile_idx_dst = tile_idx_in.p; A.tile_cnt_dst = tile_cnt_in.p;
            inner_is_outer = false;
            if (GS > 0) { cnt_in.reserve((size_t)n_blocks * JS * BI); A.cnt_dst = cnt_in.p; }   // (k_regroup wants the real entries per (sub-list, lane))
            A.mark_off = (int)((lds_force + 15) & ~(size_t)15);
            if (A.soa && A.mark_off != prune_mark_ct(A.soa)) throw ApiError{MHIP_ERR_STATE, "internal: the packed pruning pass expects its renumbering table at " + std::to_string(prune_mark_ct(A.soa)) + ", the carve-up put it at " + std::to_string(A.mark_off)};
            lds_force = (size_t)A.mark_off + prune_lds_bytes(tile_lds, BI * JS);   // + renumbering table + scan scratch + wave boxes
            if (lds_force > (size_t)MAX_LDS_BYTES) throw ApiError{MHIP_ERR_CAPACITY, "prune pass LDS carve-up exceeds 160 KiB"};
        }
        A.blk_center = blk_center.p; A.frc = frc_override ? frc_override : frc[cur].p; A.pe_part = red_part.p;
        // inside vv_run: the Σ m v partials of the integrator launch before this pass become one partial here (kernels.h, cm_finalize_in_block)
        A.cm_fin_in = nullptr; A.cm_fin_n = 0; A.cm_fin_out = nullptr;
        A.vel = nullptr; A.pos_next = nullptr; A.dt = T(0); A.dt2 = T(0); A.cm_in = nullptr; A.cm_n = 0; A.cm_pub = nullptr; A.step_seq = 0; A.cm_out = nullptr; A.trk_part = nullptr; A.snap_a = nullptr; A.snap_b = nullptr;
        const bool cm_fin = (in_vv_fused || in_lang_fused) && !energy && n_ghost == 0 && part == 0 && cm_pending == 2 && n_cm_step > 1 && n_cm_step <= 65536;      // (the energy variants do not carry the sum)
        if (cm_fin) { cm_fin_buf.reserve(4); A.cm_fin_in = cm_src(); A.cm_fin_n = n_cm_step; A.cm_fin_out = cm_fin_buf.p; }
        else if (cm_fin_solo_src && !energy && n_ghost == 0 && part == 0 && !cm_fin_solo_done) {      // (mhip_domain_run on one brick: halo_mid's partials)
            cm_fin_buf.reserve(4); A.cm_fin_in = cm_fin_solo_src; A.cm_fin_n = n_cm_step; A.cm_fin_out = cm_fin_buf.p; cm_fin_solo_done = true;
        }
        A.level_pairs = (prune && JS == 2) ? 1 : 0;      // (an atom's two sub-lists levelled before they are padded: kernels.h)
        A.dbg = (!prune && !energy) ? stamps_begin((size_t)n_blocks * 16 * 8) : nullptr;
        // the fused step: this pass also integrates (see step_req)
        step_done = false;
        bool do_step = false;
        if constexpr (std::is_same<T, float>::value) {
            do_step = step_req.on && fuse_step_env && fast_f32 && A.soa != 0 && use_inner && !prune && part == 0 && !frc_override && (n_ghost == 0 || halo_req.on) && cm_pending != 1;
            if (halo_req.on && !do_step) throw ApiError{MHIP_ERR_STATE, "internal: the fused ghosted step was asked for a pass that cannot integrate"};
            if (do_step) {
                pos_alt.reserve(cap); cm_blk.reserve(2 * 4 * (size_t)n_blocks + 8);
                if (!cm_pub.p) { cm_pub.reserve(4); MHIP_HIP(hipMemsetAsync(cm_pub.p, 0, 4 * sizeof(unsigned long long), stream)); }      // (launch numbers start at 1)
                A.vel = vel[cur].p; A.pos_next = pos_alt.p; A.dt = T(step_req.dt); A.dt2 = T(step_req.dt) / T(2);
                A.cm_in = cm_pending == 2 ? cm_src() : (const double*)nullptr; A.cm_n = n_cm_step; A.cm_pub = cm_pub.p; A.step_seq = ++step_seq;
                A.cm_out = step_req.cm ? cm_blk.p + (size_t)step_half * 4 * n_blocks : (double*)nullptr;
                A.trk_part = nullptr; A.snap_a = pos_snap_in.p; A.snap_b = pos_snap.p;
                if (step_req.measure) { trk_part.reserve(3 * (size_t)std::max(n_blocks, 1024)); trk_out.reserve(4); A.trk_part = trk_part.p; }
                A.cm_fin_in = nullptr; A.cm_fin_n = 0; A.cm_fin_out = nullptr;      // (the head workgroup of the launch sums the partials instead)
                if (halo_req.on) hx_fill(A);
                if (step_req.lang) { if (halo_req.on) throw ApiError{MHIP_ERR_STATE, "internal: Langevin update asked of a ghosted step"}; A.orig = orig[cur].p; A.S = *step_req.lang; }
            }
        }
        prof.begin(prune ? 4 : 0, stream);   // stage 4 = force passes that also prune the outer list
        tr(prune ? "k_forces (prune)" : (energy ? "k_forces (energy)" : (do_step ? (halo_req.on ? "k_forces (fused ghosted step)" : "k_forces (fused step)") : "k_forces")));
        if constexpr (std::is_same<T, float>::value) {
            if (do_step) {
                if (halo_req.on && xf.shared_device && halo_waiter_env) {
                    // Several ranks on ONE device: the peers need the same compute units to produce what this launch would wait for, and a grid of resident,
                    // spinning workgroups can leave their kernels no room (a search kernel's 100 KB of LDS next to two spinning blocks per unit: a 2 s stall,
                    // then the time-out).  A one-workgroup launch waits instead; the pass behind it finds every word in place.
                    XferWait W{}; W.mine = reinterpret_cast<const XferHeader*>(xf.region); W.parity = (int)(xf.seq & 1u); W.seq = xf.seq; W.peers = xf.d_peers.p; W.n_peers = xf.n_peers; W.err = xf.err.p; W.ticks = xf_ticks();
                    hipLaunchKernelGGL(k_xfer_wait_all, dim3(1), dim3(64), 0, stream, W);
                }
                launch_forces_uniform_f32(A, false, false, lds_force, (unsigned)(BI * JS), stream, true, halo_req.on, step_req.lang != nullptr);
                if (halo_req.on) ++xf.seq;      // (the launch's last wave announces exchange xf.seq at the peers)
                std::swap(pos[cur].p, pos_alt.p); std::swap(pos[cur].n, pos_alt.n);      // the epilogues wrote the drifted coordinates into the other buffer: it is the current one now
                step_done = true; ++n_fused_steps; step_parts = n_blocks;
            } else launch_forces_any(A, energy);
        } else launch_forces_any(A, energy);
        if (cm_fin && !do_step) { cm_ext = cm_fin_buf.p; n_cm_step = 1; }
        tr("after k_forces");
        if constexpr (std::is_same<T, float>::value) {
            if (prune && GS > 0) {      // the list this prune wrote, dealt to the groups (it stays as it is for every other kind of pass)
                nbr_gs.reserve((size_t)n_blocks * JS * GS * R_cap * BI); rows_gs.reserve((size_t)n_blocks * JS * (BI / WAVE));
                RegroupArgs R{BI, ilog2(BI), JS, GS, ilog2(GS), R_cap, (const uint2*)nbr_in.p, (const uint16_t*)cnt_in.p, (const int32_t*)tile_cnt_in.p, nbr_gs.p, rows_gs.p,
                              (int)std::min<size_t>((size_t)MAX_LDS_BYTES - (size_t)JS * BI * 16 - 64, (size_t)JS * R_cap * BI * 8), GS * R_cap};
                tr("k_regroup");
                launch_regroup(R, n_blocks, stream);
                gs_balance();
                gs_list_id = n_filters + 1;      // (n_filters counts this prune below)
            }
        }
        prof.end(prune ? 4 : 0, stream);
        MHIP_HIP(hipGetLastError());
        ++n_force_calls;
        if (A.dbg && (n_force_calls % stamps_every()) == 0) stamps_report();
        if (prune) {   // validity of the pruned list: nobody moved more than half the margin since the outer search
            // one single-block launch that leaves its figures in pinned host memory (no zeroing launch, no copy launch) …
            const bool late = xf_direct && n_ghost > 0 && prune_late_env;      // inside mhip_domain_run on a ghosted sub-domain: read behind an event (prune_resolve)
            if (late && !h_prune) { MHIP_HIP(hipHostMalloc((void**)&h_prune, N_FLAGS * sizeof(int32_t))); MHIP_HIP(hipEventCreateWithFlags(&ev_prune, hipEventDisableTiming)); }
            int32_t* const h_dst = late ? h_prune : h_flags;
            hipLaunchKernelGGL(k_prune_summary, dim3(1), dim3(1024), 0, stream, n_blocks, n_blocks * JS * (BI / WAVE), n_blocks * (BI / WAVE), R_cap,
                               (const int32_t*)tile_cnt_in.p, wave_rows_in.p, (const float*)blk_disp2.p, flags.p, h_dst);
            if (n_ghost > 0) {   // … the blocks recorded the displacement of the owned atoms; the ghosts' comes on top
                hipLaunchKernelGGL(k_max_disp<T>, dim3(std::min(cdiv(n_ghost, 256), 1024)), dim3(256), 0, stream, n_ghost, (const T4*)pos[cur].p + n_owned, (const T4*)pos_snap.p + n_owned,
                                   reinterpret_cast<unsigned int*>(flags.p + FLAG_MAX_DISP2), G);
                MHIP_HIP(hipMemcpyAsync(h_dst, flags.p, N_FLAGS * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            }
            ++n_filters; ghost_flags_in_ok = false; next_check_step = -1; hx.tile_ok = false; hx.trk_step = -1;
            inner_valid = true;
            if (late) {
                MHIP_HIP(hipEventRecord(ev_prune, stream));
                prune_pending = true; prune_pending_id = n_filters; prune_disp_exceeded = false;
                max_tile_in = max_tile;      // (an upper bound until the figures are read: the pruned tile of a block is a subset of its outer tile)
            } else {
                MHIP_HIP(hipStreamSynchronize(stream));
                prune_pending = false;
                float d2; std::memcpy(&d2, &h_flags[FLAG_MAX_DISP2], sizeof(float));
                total_rows = h_flags[FLAG_TOTAL_ROWS]; max_tile_in = h_flags[FLAG_MAX_TILE];
                prune_disp_exceeded = 2.0 * std::sqrt((double)d2) > prune_margin() * 0.98;
                if (debug_on) std::fprintf(stderr, "[mhip] prune: max disp %.5f nm (margin %.3f) rows %lld exceeded %d calls %lld\n", std::sqrt((double)d2), prune_margin(), (long long)total_rows, (int)prune_disp_exceeded, (long long)n_force_calls);
            }
        }
    }

    DBuf<double> cm_fin_buf;
    const double* cm_fin_solo_src = nullptr; bool cm_fin_solo_done = false;
    // The fused step of the large one-type fluids inside mhip_vv_run (kernels.h, k_forces STEP): a plain pair pass whose epilogue is the integrator launch — second kick
    // of this step, first kick + drift of the next, into the other position buffer (swapped in behind the launch) — with Σ m v summed and published by the grid's first
    // workgroup.  Asked for by vv_run (step_req), carried out by launch_pair_kernel when the pass is a packed plain one; every other pass keeps pair pass + k_vv_mid.
    // (gcv: the same request for a small system's step, whose last force launch — interpolation + bonded sums — can integrate: step_fused.h, k_gather_collect_vv)
    struct StepReq { bool on = false, gcv = false, cm = false, measure = false; double dt = 0; const StochP<T>* lang = nullptr; } step_req;      // (lang: gcv with the Langevin-middle update, mhip_langevin_run)
    bool step_done = false; int step_half = 0; uint32_t step_seq = 0; int64_t n_fused_steps = 0;
    int step_parts = 0;      // per-block partials (Σ m v in cm_blk's current half, maxima in trk_part) the fused step left behind
    const bool fuse_gcv_env = env_int("MOLLYHIP_FUSE_GATHER_VV", 1) != 0;
    DBuf<T4> pos_alt; DBuf<double> cm_blk; DBuf<unsigned long long> cm_pub;
    const bool fuse_step_env = env_int("MOLLYHIP_FUSE_STEP", 1) != 0;
    // 0: no one-workgroup waiter in front of a fused ghosted step even when ranks share the device — the blocks' own (bounded) waits are then what orders the step,
    // as on separate devices.  Safe only while the ranks' grids together fit the device (the small test systems); tests/test_gpu_domain.py runs it to exercise those waits
    const bool halo_waiter_env = env_int("MOLLYHIP_HALO_WAITER", 1) != 0;
    const bool prune_late_env = env_int("MOLLYHIP_PRUNE_LATE", 1) != 0;      // 0: a pruning pass of a ghosted run drains the stream for its summary, as on a single domain (tests compare the two)
    // Would a plain pass now be the packed one-type loop over a valid inner list — the only pass that can integrate?  (launch_pair_kernel's own conditions, asked
    // BEFORE the step is put together: the domain loop leaves the unpack, integrator and pack launches out only when the pass will do their work)
    bool packed_step_possible() {
        if constexpr (!std::is_same<T, float>::value) return false;
        if (!fuse_step_env || !dual || !inner_valid || stale || ljm != LJ_DIST_UNIFORM || coulm != MHIP_COUL_NONE || minimg || n_special != 0 || eshift != ESHIFT_SCALED || I.lj_c12 == T(0)) return false;
        prune_resolve(false);
        if (prune_pending && max_tile_in + 1 >= SOA_STRIDES[2]) prune_resolve(true);      // (the outer list's bound does not fit the packed loop: the pruned list's own figure is needed now)
        if (max_tile_in + 1 >= SOA_STRIDES[2]) return false;
        carve_force_lds(max_tile_in);
        return !segmented;
    }

    // ---- the fused step of a ghosted sub-domain (kernels.h HaloStep; tables: halo_step.h) -------------------------------------------------------------------
    struct Hx {
        DBuf<int32_t> order, flags, tsrc, ghost_row, cm_row, snd_start, snd_cnt, blk_send; DBuf<HaloSend> snd; DBuf<float*> cm_dst; DBuf<uint32_t*> ann;
        bool plan_ok = false, tile_ok = false;      // per ghost plan + sort | per prune
        int64_t trk_step = -1;                      // the step whose coordinates the last fused launch measured against the snapshots (trk_part), −1: none
        int64_t n_steps = 0;
        void release() { order.release(); flags.release(); tsrc.release(); ghost_row.release(); cm_row.release(); snd_start.release(); snd_cnt.release(); blk_send.release(); snd.release(); cm_dst.release(); ann.release(); }
    } hx;
    struct HaloReq { bool on = false, cm_in = false; } halo_req;
    static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void hx_build() {
        if constexpr (std::is_same<T, float>::value) {
            if (!hx.plan_ok) {
                if (debug_on) { std::fprintf(stderr, "[mhip %d] %.3f hx_build: draining the stream first\n", xf.rank, now_ms()); MHIP_HIP(hipStreamSynchronize(stream)); std::fprintf(stderr, "[mhip %d] %.3f hx_build: drained\n", xf.rank, now_ms()); }
                tr("k_hx_* (per plan)");
                hx.ghost_row.reserve(std::max<int64_t>(n_ghost, 1)); hx.cm_row.reserve(XFER_MAX_RANKS * 4);
                if (hp.n_recv_rows > 0) hipLaunchKernelGGL(k_hx_rows, dim3(cdiv(hp.n_recv_rows, 256)), dim3(256), 0, stream, hp.n_recv_rows, hp.recv_dst, hp.first_ghost, (const int32_t*)inv.p, n_owned, hx.ghost_row.p, hx.cm_row.p);
                hx.snd_cnt.reserve(n_owned + 1); hx.snd_start.reserve(n_owned + 1);
                MHIP_HIP(hipMemsetAsync(hx.snd_cnt.p, 0, (size_t)(n_owned + 1) * sizeof(int32_t), stream));
                if (hp.n_send_rows > 0) hipLaunchKernelGGL(k_hx_send_count, dim3(cdiv(hp.n_send_rows, 256)), dim3(256), 0, stream, hp.n_send_rows, hp.send_idx, (const int32_t*)inv.p, hx.snd_cnt.p);
                size_t tb = 0; MHIP_HIP(exclusive_sum_i32(nullptr, tb, hx.snd_cnt.p, hx.snd_start.p, (int)(n_owned + 1), stream));
                if (tb + 256 > cub_tmp.n) { MHIP_HIP(hipStreamSynchronize(stream)); cub_tmp.reserve(tb + 256); }
                tb = cub_tmp.n; MHIP_HIP(exclusive_sum_i32(cub_tmp.p, tb, hx.snd_cnt.p, hx.snd_start.p, (int)(n_owned + 1), stream));
                MHIP_HIP(hipMemsetAsync(hx.snd_cnt.p, 0, (size_t)(n_owned + 1) * sizeof(int32_t), stream));
                hx.snd.reserve(std::max<int64_t>(hp.n_send_rows, 1)); hx.cm_dst.reserve(std::max(hp.n_send_cm, 1));
                if (hp.n_send_rows > 0) hipLaunchKernelGGL(k_hx_send_fill, dim3(cdiv(hp.n_send_rows, 256)), dim3(256), 0, stream, hp.n_send_rows, hp.send_idx, (const int32_t*)inv.p, (const float*)hp.send_shift,
                                                           (const int32_t*)xf.row_peer.p, (const int32_t*)xf.row_dst.p, xf.peers, (const int32_t*)hx.snd_start.p, hx.snd_cnt.p, hx.snd.p);
                if (hp.n_send_cm > 0) hipLaunchKernelGGL(k_hx_cm_dst, dim3(cdiv(hp.n_send_cm, 64)), dim3(64), 0, stream, hp.n_send_cm, hp.send_cm_pos, (const int32_t*)xf.row_peer.p, (const int32_t*)xf.row_dst.p, xf.peers, hx.cm_dst.p);
                hx.blk_send.reserve(n_blocks);
                hipLaunchKernelGGL(k_hx_blk_send, dim3(n_blocks), dim3(64), 0, stream, n_blocks, BI, n_owned, (const int32_t*)hx.snd_start.p, hx.blk_send.p);
                // the peers' sequence words for my rows, as addresses (parity 0)
                std::vector<uint32_t*> ann((size_t)std::max(xf.n_peers, 1), nullptr);
                for (int q = 0; q < xf.n_peers; ++q) ann[q] = reinterpret_cast<uint32_t*>(xf.peers.region[xf.peer_rank[q]] + offsetof(XferHeader, seq_in)) + xf.rank;
                hx.ann.reserve(ann.size());
                MHIP_HIP(hipMemcpyAsync(hx.ann.p, ann.data(), ann.size() * sizeof(uint32_t*), hipMemcpyHostToDevice, stream));
                MHIP_HIP(hipStreamSynchronize(stream));      // (ann is a host temporary; once per ghost plan)
                if (debug_on) std::fprintf(stderr, "[mhip %d] %.3f hx_build: tables made\n", xf.rank, now_ms());
                hx.plan_ok = true; hx.tile_ok = false;
            }
            if (!hx.tile_ok) {
                tr("k_hx_* (per prune)");
                const int bpx = cdiv(n_blocks, 8);
                hx.tsrc.reserve((size_t)n_blocks * T_cap); hx.flags.reserve(n_blocks); hx.order.reserve((size_t)bpx * 8);
                hipLaunchKernelGGL(k_hx_tile, dim3(n_blocks), dim3(256), 0, stream, n_blocks, T_cap, n_owned, (const int32_t*)(inner_is_outer ? tile_idx.p : tile_idx_in.p), (const int32_t*)(inner_is_outer ? tile_cnt.p : tile_cnt_in.p),
                                   (const int32_t*)hx.ghost_row.p, (const int32_t*)hx.blk_send.p, hx.tsrc.p, hx.flags.p);
                hipLaunchKernelGGL(k_hx_order, dim3(8), dim3(256), 0, stream, n_blocks, bpx, (const int32_t*)hx.flags.p, hx.order.p);
                hx.tile_ok = true;
            }
            MHIP_HIP(hipGetLastError());
        }
    }
    void hx_fill(ForceArgs<T>& A) {
        if constexpr (std::is_same<T, float>::value) {
            hx_build();
            HaloStep& H = A.H;
            const int par = (int)(xf.seq & 1u);
            H.order = hx.order.p; H.flags = hx.flags.p; H.tsrc = hx.tsrc.p;
            H.rows = xf_rows(par); H.seq_in = reinterpret_cast<const XferHeader*>(xf.region)->seq_in[par];
            H.peers = xf.d_peers.p; H.n_peers = xf.n_peers; H.seq_wait = xf.seq; H.err = xf.err.p; H.ticks = xf_ticks();
            H.snd_start = hx.snd_start.p; H.snd = hx.snd.p; H.half_stride = (int64_t)xf.rows_cap * 3;
            H.parity_send = (int)((xf.seq + 1u) & 1u); H.seq_send = xf.seq + 1u;
            H.ann = hx.ann.p; H.done = xf.done.p; H.n_done = (unsigned int)(n_blocks * (BI / WAVE));
            H.cm_row = hx.cm_row.p; H.cm_dst = hx.cm_dst.p; H.cm_rows = hp.cm_rows; H.cm_all = cm_all.p;
            A.cm_in = halo_req.cm_in ? (const double*)cm_all.p : (const double*)nullptr;      // (a flag here: v_cm comes from cm_all[0] and the peers' rows, halo_cm_publish)
            A.cm_n = 0;
            ++hx.n_steps;
        }
    }
    // Can step_n of mhip_domain_run be ONE launch?  Peers reached through the mapped regions, a plan with the fused message layout, nothing but a plain packed pass due.
    bool halo_fused_ok(int64_t step_n) {
        if (!(xf_direct && xf.n_peers > 0 && n_ghost > 0 && hp_set && xf.routes) || replan_now) return false;
        if (hp.cm_rows != 3 || hp.n_cm_peers != xf.n_peers || (int)xf.peer_rank.size() != xf.n_peers) return false;
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        if (check_due(step_n, every) && step_n != last_build_step && !(host_prune && inner_valid)) return false;      // (refresh() of such a step only books it)
        return packed_step_possible();
    }
    void halo_fused(int64_t step_n, double dt, bool cm, bool measure) {
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        cur_dt = dt;
        if (check_due(step_n, every) && step_n != last_build_step && dual) refresh(step_n);
        step_req.on = true; step_req.gcv = false; step_req.cm = cm; step_req.measure = measure; step_req.dt = dt;
        halo_req.on = true; halo_req.cm_in = halo_cm_in;
        step_done = false;
        struct Off { StepReq& s; HaloReq& h; ~Off() { s.on = false; h.on = false; } } off{step_req, halo_req};
        step_forces(step_n);
        if (!step_done) throw ApiError{MHIP_ERR_STATE, "internal: the fused ghosted step did not launch"};
        step_done = false;
        if (cm) step_half ^= 1;
        halo_cm_in = cm; frc_valid = false; pend_a = nullptr; cm_pending = 0; cm_ext = nullptr;
        hx.trk_step = measure ? step_n + 1 : -1;
        MHIP_HIP(hipGetLastError());
    }

    // the (block, group) items of the group-split pass handed to its workgroups so that every compute unit gets a like share of rows (forces_gs.hip, k_gs_balance)
    DBuf<uint16_t> gs_item; int cu_count = 0;
    void gs_balance() {
        if (!cu_count) { int dev = 0; MHIP_HIP(hipGetDevice(&dev)); hipDeviceProp_t pr; MHIP_HIP(hipGetDeviceProperties(&pr, dev)); cu_count = std::max(1, pr.multiProcessorCount); }
        const int GS = gs_groups();
        gs_item.reserve((size_t)n_blocks * GS);
        launch_gs_balance((const int32_t*)rows_gs.p, n_blocks, JS, GS, BI / WAVE, cu_count, gs_item.p, stream);
    }

    // the outer list, searched at this step with the radius the inner list would be pruned to, becomes the inner list (see inner_is_outer)
    void adopt_outer_list() {
        const int GS = gs_groups();
        prof.begin(4, stream);
        pos_snap_in.reserve(cap);
        MHIP_HIP(hipMemcpyAsync(pos_snap_in.p, pos[cur].p, (size_t)n_tot * sizeof(T4), hipMemcpyDeviceToDevice, stream));
        nbr_gs.reserve((size_t)n_blocks * JS * GS * R_cap * BI); rows_gs.reserve((size_t)n_blocks * JS * (BI / WAVE));
        RegroupArgs R{BI, ilog2(BI), JS, GS, ilog2(GS), R_cap, (const uint2*)nbr.p, (const uint16_t*)cnt_outer.p, (const int32_t*)tile_cnt.p, nbr_gs.p, rows_gs.p,
                      (int)std::min<size_t>((size_t)MAX_LDS_BYTES - (size_t)JS * BI * 16 - 64, (size_t)JS * R_cap * BI * 8), GS * R_cap};
        R.dbg = stamps_begin((size_t)n_blocks * 16 * 8);
        tr("k_regroup (outer list)");
        launch_regroup(R, n_blocks, stream);
        if (R.dbg) stamps_dump(".regroup", (size_t)n_blocks * 8);
        gs_balance();
        prof.end(4, stream);
        MHIP_HIP(hipGetLastError());
        inner_is_outer = true; max_tile_in = max_tile; last_prune_step = pass_step;
        ++n_filters; ++n_adopted; gs_list_id = n_filters;
        ghost_flags_in_ok = false; next_check_step = -1; hx.tile_ok = false;
        inner_valid = true; prune_disp_exceeded = false;
        if (debug_on) std::fprintf(stderr, "[mhip] outer list adopted as the inner list (no prune): rows %lld calls %lld\n", (long long)total_rows, (long long)n_force_calls);
    }

    // Time stamps inside the block kernels — only in libraries built with -DMHIP_STAMPS=1 (common.h), where MOLLYHIP_DBG_TIMES=n arms them (every n-th pass
    // is reported / dumped to $MOLLYHIP_DBG_DUMP[.build|.regroup], tools/gs_times.py, tools/build_times.py).  The product build carries none of it: its kernels get a null pointer.
#if MHIP_STAMPS
    DBuf<unsigned long long> dbg_buf;
    static int stamps_every() { static const int n = env_int("MOLLYHIP_DBG_TIMES", 0); return std::max(n, 1); }
    unsigned long long* stamps_begin(size_t words) {
        static const int on = env_int("MOLLYHIP_DBG_TIMES", 0);
        if (!on) return nullptr;
        dbg_buf.reserve(words); MHIP_HIP(hipMemsetAsync(dbg_buf.p, 0, words * sizeof(unsigned long long), stream));
        return dbg_buf.p;
    }
    void stamps_dump(const char* suffix, size_t words) {      // → the file $MOLLYHIP_DBG_DUMP + suffix
        std::vector<unsigned long long> h(words);
        MHIP_HIP(hipStreamSynchronize(stream));
        MHIP_HIP(hipMemcpy(h.data(), dbg_buf.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (const char* path = std::getenv("MOLLYHIP_DBG_DUMP")) { if (FILE* f = std::fopen((std::string(path) + suffix).c_str(), "wb")) { std::fwrite(h.data(), sizeof(unsigned long long), h.size(), f); std::fclose(f); } }
    }
    void stamps_report() {
        const int nw = BI * JS / WAVE;
        std::vector<unsigned long long> h((size_t)n_blocks * nw * 8);
        MHIP_HIP(hipStreamSynchronize(stream));
        MHIP_HIP(hipMemcpy(h.data(), dbg_buf.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long w0 = ~0ull, w3 = 0; double sum[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, clk = 0; int n = 0;
        std::vector<double> loop_us;
        for (size_t q = 0; q < (size_t)n_blocks * nw; ++q) {
            const unsigned long long* d = &h[q * 8];
            if (!d[7]) continue;
            w0 = std::min(w0, d[4]); w3 = std::max(w3, d[7]);
            for (int k = 0; k < 3; ++k) { const double us = (double)(d[5 + k] - d[4 + k]) * 0.01; sum[k] += us; mx[k] = std::max(mx[k], us); }
            loop_us.push_back((double)(d[6] - d[5]) * 0.01);
            if (d[7] > d[4]) clk += (double)(d[3] - d[0]) / ((double)(d[7] - d[4]) * 10.0);   // shader cycles per ns
            ++n;
        }
        std::sort(loop_us.begin(), loop_us.end());
        std::fprintf(stderr, "[mhip dbg] pair kernel: %d waves, first entry -> last exit %.2f us | per wave mean (max) us: staging %.2f (%.2f) row walk %.2f (%.2f) reduce+store %.2f (%.2f) | row walk p10 %.2f p50 %.2f p90 %.2f | shader clock %.3f GHz\n",
                     n, (double)(w3 - w0) * 0.01, sum[0] / n, mx[0], sum[1] / n, mx[1], sum[2] / n, mx[2], loop_us[n / 10], loop_us[n / 2], loop_us[(size_t)n * 9 / 10], clk / n);
    }
#else
    static int stamps_every() { return 1; }
    unsigned long long* stamps_begin(size_t) { return nullptr; }
    void stamps_dump(const char*, size_t) {}
    void stamps_report() {}
#endif
    // How far atoms may have moved since the outer search for a prune to be trustworthy: the outer list holds every pair within
    // r_list + outer_margin of then, the prune wants every pair within rc_max + skin_in of now.
    double prune_margin() const { return outer_margin + (skin - skin_in); }

    const bool inner_skin_fixed = env_int("MOLLYHIP_INNER_SKIN_FIXED", 0) != 0;

    double read_sum(int n_part) {
        hipLaunchKernelGGL(k_sum_double, dim3(1), dim3(256), 0, stream, n_part, (const double*)red_part.p, red_out.p);
        MHIP_HIP(hipMemcpyAsync(h_red, red_out.p, sizeof(double), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        return h_red[0];
    }

    void export_frc(int accumulate, void* f_xyz, int mem_kind) {
        const size_t cnt = 3 * (size_t)n_owned;
        if (mem_kind == MHIP_MEM_DEVICE) {
            hipLaunchKernelGGL(k_export_forces<T>, dim3(cdiv(n_owned, 256)), dim3(256), 0, stream, n_owned, (const int32_t*)orig[cur].p, (const T4*)frc[cur].p, (T*)f_xyz, accumulate);
        } else {
            stage_a.reserve(cnt);
            if (accumulate) MHIP_HIP(hipMemcpyAsync(stage_a.p, f_xyz, cnt * sizeof(T), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(k_export_forces<T>, dim3(cdiv(n_owned, 256)), dim3(256), 0, stream, n_owned, (const int32_t*)orig[cur].p, (const T4*)frc[cur].p, stage_a.p, accumulate);
            MHIP_HIP(hipMemcpyAsync(f_xyz, stage_a.p, cnt * sizeof(T), hipMemcpyDeviceToHost, stream));
            MHIP_HIP(hipStreamSynchronize(stream));
        }
        MHIP_HIP(hipGetLastError());
    }

    // all forces of one MD step: the pair kernel overwrites frc[cur]; the bonded terms and the PME reciprocal part ride in the same launch where the shapes
    // allow (forces_gs.hip), else follow on the same stream; a second force array a fused launch leaves behind (pend_a) is added by the consumer
    // (second kick, or fold_side_forces)
    void step_forces(int64_t step_n) {
        pass_step = step_n;
        if (pme.on() && n_ghost > 0) throw ApiError{MHIP_ERR_UNSUPPORTED, "PME runs on a single domain (SURVEY §8(e): 6mrr-size systems are replicas only)"};
        // (group-split passes leave partial forces that the per-atom sums of the bonded slots fold in: only where such a launch follows)
        const bool gs_ok = bonded.any() && n_ghost == 0;
        const bool small_fused = bonded.any() && pme.on() && n_ghost == 0;
        // (with the stage timers on, every job keeps its own launch: a stage's time is then that job's — bench.py's profiling pass, never its timed region)
        fuse_spread_next = gs_ok && small_fused && pme.order >= 4 && pme.order <= 6 && !prof.on;
        fuse_terms_next = gs_ok && !small_fused && !pme.on() && !prof.on;      // (no PME: the bonded terms alone ride with the pair groups)
        spread_fused = terms_fused = false;
        launch_pair_kernel(false, interior_done ? 2 : 0, gs_ok);   // (the blocks without ghosts may have run already, while the ghosts were on the wire)
        fuse_spread_next = fuse_terms_next = false;
        interior_done = false;
        bool redo = false;
        if (prune_disp_exceeded) {   // the outer list could not vouch for this pass: search again and redo it on the fresh list
            after_forces(step_n);
            launch_pair_kernel(false);
            redo = true;
        }
        if (gs_used) bonded.fold(frc_parts.p, gs_groups() - 1, cap);      // the next collect launch adds the groups' partial forces
        pend_a = nullptr;
        // small systems: charge spreading next to the bonded terms, force interpolation next to the bonded sums (step_fused.h)
        if (small_fused) {
            frc_side.reserve(cap);
            // … and, on a mid-run step of vv_run, the integrator in that last launch (v_cm of the step before as ONE partial, or none pending)
            GcvArgs<T> V; const GcvArgs<T>* vp = nullptr;
            if (step_req.gcv && fuse_gcv_env && !redo && (cm_pending == 0 || (cm_pending == 2 && n_cm_step == 1)) && pme.order >= 4 && pme.order <= 6) {
                const int nb = (int)Pme<T>::atom_blocks(n_owned);
                cm_blk.reserve(2 * 4 * (size_t)nb + 8);
                if (step_req.measure) { trk_part.reserve(3 * (size_t)std::max(nb, 1024)); trk_out.reserve(4); }
                std::memset(&V, 0, sizeof(V));
                V.vel = vel[cur].p; V.dt = T(step_req.dt); V.dt2 = T(step_req.dt) / T(2); V.G = G;
                V.cm_in = cm_pending == 2 ? cm_src() : (const double*)nullptr;
                V.cm_out = step_req.cm ? cm_blk.p + (size_t)step_half * 4 * nb : (double*)nullptr;
                V.snap_a = pos_snap_in.p; V.snap_b = pos_snap.p; V.trk_part = step_req.measure ? trk_part.p : (float*)nullptr;
                if (step_req.lang) V.S = *step_req.lang;
                vp = &V; step_parts = nb;
            }
            prof.begin(6, stream);
            launch_pme_bonded_fused<T>(stream, pme, bonded, G, I, n_owned, cap, pos[cur].p, inv.p, orig[cur].p, frc[cur].p, frc_side.p, spread_fused, vp, vp && step_req.lang);
            prof.end(6, stream);
            if (vp) { step_done = true; ++n_fused_steps; pend_a = nullptr; frc_valid = false; return; }
            pend_a = frc_side.p;
            frc_valid = true;
            return;
        }
        if (bonded.any()) { prof.begin(5, stream); bonded.launch_forces(stream, G, I, pos[cur].p, inv.p, frc[cur].p, orig[cur].p, n_owned, cap, terms_fused); prof.end(5, stream); }
        launch_pme_forces();
        frc_valid = true;
    }
    // frc[cur] += the second force array, for consumers other than the second kick
    void fold_side_forces() {
        if (!pend_a) return;
        hipLaunchKernelGGL(k_add_forces<T>, dim3(std::min(cdiv(n_owned, 256), 1024)), dim3(256), 0, stream, n_owned, frc[cur].p, pend_a);
        pend_a = nullptr;
    }

    void launch_pme_forces() {
        if (!pme.on()) return;
        if (n_ghost > 0) throw ApiError{MHIP_ERR_UNSUPPORTED, "PME runs on a single domain (SURVEY §8(e): 6mrr-size systems are replicas only)"};
        prof.begin(6, stream);
        pme.run(stream, n_owned, pos[cur].p, frc[cur].p, nullptr);
        prof.end(6, stream);
    }

  public:
    // ---------------------------------------------------------------------------------------------
    void set_stream(void* s) override {
        MHIP_HIP(hipStreamSynchronize(stream));
        if (own_stream) { (void)hipStreamDestroy(stream); own_stream = false; }
        stream = (hipStream_t)s;
    }
    void synchronize() override { MHIP_HIP(hipStreamSynchronize(stream)); }
    void set_profiling(bool on) override {
        prof.resolve(stream);
        if (on) { prof.reset(); prof.reserve(0, 2048); prof.reserve(2, 2048); for (int st : {1, 3, 4, 5, 6}) prof.reserve(st, 256); }
        prof.on = on;
    }

    void set_atom_counts(int64_t no, int64_t ng) override {
        if (no <= 0 || ng < 0 || no + ng > cap) throw ApiError{MHIP_ERR_INVALID, "atom counts exceed the context capacity"};
        if (ng > 0 && tri_mode) throw ApiError{MHIP_ERR_UNSUPPORTED, "TriclinicBoundary is single-domain"};
        n_owned = no; n_ghost = ng; n_tot = no + ng;
        // new local atom set: restart from the identity order
        hipLaunchKernelGGL(k_iota2, dim3(std::min(cdiv(n_tot, 256), 1024)), dim3(256), 0, stream, n_tot, orig[cur].p, inv.p);
        MHIP_HIP(hipMemsetAsync(frc[cur].p, 0, n_tot * sizeof(T4), stream));
        MHIP_HIP(hipGetLastError());
        stale = true; cm_pending = 0; cm_ext = nullptr; frc_valid = false; hp_set = false; halo_cm_in = false; trk_issued = false; hx.plan_ok = false; hx.trk_step = -1;
        // the search radius depends on whether there are ghosts and on the ghost margin, the blocking on the size class: a re-plan
        // that changes neither keeps the grid, its Hilbert table and the (already adapted) capacities
        const int size_class = n_owned >= 100000 ? 2 : (n_owned >= 40000 ? 1 : 0);
        const long long key = (n_ghost > 0 ? 1 : 0) | (dual_disabled ? 2 : 0) | (size_class << 2) | (margin_halvings << 4) | (margin_zero ? 128 : 0) | ((long long)std::llround(ghost_margin * 1e6) << 8);
        if (key != grid_key) { setup_grid(); choose_blocking(); grid_key = key; }
    }

    void set_atoms(const void* q, const void* sg, const void* ep, const void* ms, const void* lam, int mem_kind) override {
        flush_cm();
        frc_run_total = false; frc_before_set_state = false;   // new charges / σ / ϵ / masses: neither a finished run's forces nor those held before a set_state stand
        DBuf<T> s3, s4, s5;
        const T* dq = to_device(q, n_tot, mem_kind, stage_a);
        const T* ds = to_device(sg, n_tot, mem_kind, stage_b);
        const T* de = to_device(ep, n_tot, mem_kind, s3);
        const T* dm = to_device(ms, n_tot, mem_kind, s4);
        const T* dl = to_device(lam, n_tot, mem_kind, s5);
        hipLaunchKernelGGL(k_scatter_params<T>, dim3(cdiv(n_tot, 256)), dim3(256), 0, stream, n_tot, (const int32_t*)inv.p, dq, ds, de, dm, dl, pos[cur].p, vel[cur].p, lj[cur].p);
        MHIP_HIP(hipGetLastError());
        MHIP_HIP(hipStreamSynchronize(stream));
        // one atom type (every σ, ϵ equal and non-zero, no λ = 0) + DistanceCutoff → uniform-LJ kernel variant
        ljm = ljm_base;
        if (ljm_base == LJ_DIST && ds && de && !tri_mode) {   // (the one-type kernels know cubic boxes only)
            // decided on the device (a sub-domain hands its parameters over at every re-plan): flag ≠ 0 if some σ, ϵ differs from atom 0's or a λ is 0
            MHIP_HIP(hipMemsetAsync(flags.p, 0, N_FLAGS * sizeof(int32_t), stream));
            hipLaunchKernelGGL(k_uniform_check<T>, dim3(std::min(cdiv(n_tot, 256), 1024)), dim3(256), 0, stream, n_tot, ds, de, dl, flags.p);
            T h2[2] = {T(0), T(0)};
            MHIP_HIP(hipMemcpyAsync(h_flags, flags.p, N_FLAGS * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            MHIP_HIP(hipMemcpyAsync(&h2[0], ds, sizeof(T), hipMemcpyDeviceToHost, stream));
            MHIP_HIP(hipMemcpyAsync(&h2[1], de, sizeof(T), hipMemcpyDeviceToHost, stream));
            MHIP_HIP(hipStreamSynchronize(stream));
            apply_uniform(h_flags[FLAG_NAN] == 0, h2[0], h2[1]);
        }
        if (want_eshift() != eshift) stale = true;   // the lists in use are in the other entry format
        pc_valid = false;   // Σq, Σq² of the PME self / net-charge terms are read back only when an energy asks for them
        s3.release(); s4.release(); s5.release();
        params_set = true; frc_valid = false;
    }

    void set_exceptions(const int32_t* ei, const int32_t* ej, int64_t ne, const int32_t* si, const int32_t* sj, int64_t ns) override {
        frc_run_total = false; frc_before_set_state = false;
        // one CSR over caller indices holding both kinds; excluded beats special (neighbors.jl:410-411 tests eligible first)
        std::vector<std::vector<uint32_t>> adj(cap);
        int span = 0;
        auto add = [&](const int32_t* a, const int32_t* b, int64_t m, uint32_t flag) {
            for (int64_t k = 0; k < m; ++k) {
                if (a[k] < 0 || b[k] < 0 || a[k] >= cap || b[k] >= cap || a[k] == b[k]) throw ApiError{MHIP_ERR_INVALID, "exception pair index out of range"};
                for (int dir = 0; dir < 2; ++dir) {
                    uint32_t me = dir ? b[k] : a[k], other = dir ? a[k] : b[k];
                    bool found = false;
                    for (auto& e : adj[me]) if ((e & XL_INDEX) == other) { e |= flag; found = true; }
                    if (!found) adj[me].push_back(other | flag);
                }
                span = std::max(span, std::abs(a[k] - b[k]));
            }
        };
        add(ei, ej, ne, XL_EXCLUDED);
        add(si, sj, ns, XL_SPECIAL);
        std::vector<int32_t> st(cap + 1, 0);
        for (int64_t i = 0; i < cap; ++i) st[i + 1] = st[i] + (int32_t)adj[i].size();
        std::vector<uint32_t> ls((size_t)st[cap]);
        for (int64_t i = 0; i < cap; ++i) std::copy(adj[i].begin(), adj[i].end(), ls.begin() + st[i]);
        MHIP_HIP(hipStreamSynchronize(stream));
        has_exc = (ne + ns) > 0; n_special = ns;
        xl_span = span;
        xl_start.reserve(cap + 1); xl_list.reserve(std::max<size_t>(ls.size(), 1));
        MHIP_HIP(hipMemcpy(xl_start.p, st.data(), (cap + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
        if (!ls.empty()) MHIP_HIP(hipMemcpy(xl_list.p, ls.data(), ls.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        stale = true;   // ≙ cache invalidation after append_excluded_pairs! (test/gpu_consistency.jl:494-527)
    }

    // (new terms change the total force: what a finished run left in frc[cur] is not the next run's first force any more)
    void set_bonds(int64_t n, const int32_t* i, const int32_t* j, const void* k, const void* r0) override { frc_run_total = false; bonded.set_bonds(cap, n, i, j, (const T*)k, (const T*)r0); }
    void set_angles(int64_t n, const int32_t* i, const int32_t* j, const int32_t* k, const void* kth, const void* th0) override { frc_run_total = false; bonded.set_angles(cap, n, i, j, k, (const T*)kth, (const T*)th0); }
    void set_torsions(int64_t n, const int32_t* i, const int32_t* j, const int32_t* k, const int32_t* l, const int32_t* per, const void* ph, const void* kt) override {
        frc_run_total = false;
        bonded.set_torsions(cap, n, i, j, k, l, per, (const T*)ph, (const T*)kt);
    }
    void set_ewald_exclusions(int64_t n, const int32_t* i, const int32_t* j) override { frc_run_total = false; bonded.set_ewx(cap, n, i, j); }

    // set_state raises these words on the device when a coordinate / a velocity really differs from what the engine held
    DBuf<int32_t> state_changed; bool state_pending = false;
    bool frc_before_set_state = false;   // were the forces current when the (possibly identical) coordinates came in?
    void set_state(const void* xyz, const void* v, int mem_kind) override {
        flush_cm();
        const T* dx = to_device(xyz, 3 * (size_t)n_tot, mem_kind, stage_a);
        const T* dv = to_device(v, 3 * (size_t)n_owned, mem_kind, stage_b);
        state_changed.reserve(2);
        if (!state_pending) { MHIP_HIP(hipMemsetAsync(state_changed.p, 0, 2 * sizeof(int32_t), stream)); frc_before_set_state = frc_valid; }
        state_pending = true;
        hipLaunchKernelGGL(k_scatter_state<T>, dim3(cdiv(n_tot, 256)), dim3(256), 0, stream, n_tot, n_owned, (const int32_t*)inv.p, dx, dv, pos[cur].p, vel[cur].p, G, state_changed.p);
        MHIP_HIP(hipGetLastError());
        if (mem_kind == MHIP_MEM_HOST) MHIP_HIP(hipStreamSynchronize(stream));
        if (xyz) {
            // New coordinates, same atoms: the sorted order, the tiles and the pair lists stay (≙ the reference's GPU path, which re-reads
            // moved coordinates at every call and redoes its sort / tile search only at the step cadence, ext/MollyCUDAExt.jl:774-783).
            // Whether the lists still cover every cutoff sphere is decided by displacement at the next force call (lists_after_set_state);
            // lists that have no skin to spend are rebuilt at once, as before.
            if (!stale && (dual || lazy_single)) coords_moved = true; else stale = true;
            frc_valid = false; state_set = true;
        }
    }

    // after set_state handed over moved coordinates: keep the lists if nobody outran their margins, else prune / search again
    void lists_after_set_state() {
        bool vel_new = false;
        if (state_pending) {   // what did set_state really change?  (a state handed back as it was — a run continued in chunks — changes nothing)
            state_pending = false;
            int32_t h[2] = {1, 1};
            MHIP_HIP(hipMemcpyAsync(h, state_changed.p, sizeof(h), hipMemcpyDeviceToHost, stream));
            MHIP_HIP(hipStreamSynchronize(stream));
            if (!h[0]) { coords_moved = false; if (frc_before_set_state && !stale) frc_valid = true; }   // the same coordinates again: what was computed for them stands
            frc_before_set_state = false;
            vel_new = h[1] != 0;
            if (h[0] || h[1]) trk_issued = false;      // a measurement in flight describes the state that was replaced
        }
        if (vel_new && !stale && (dual || lazy_single) && !coords_moved) {
            // New velocities on old coordinates (re-thermalisation, a temperature ramp, a replica-exchange swap): the inner skin was
            // sized for the speeds of the run before.  Look at the new ones now — the drift bound of the checks — instead of at the
            // next cadence step.
            vel_check_due = true;
        }
        if (!coords_moved) return;
        coords_moved = false;
        if (stale) return;
        vel_check_due = vel_check_due || vel_new;
        if (dual) {
            const double d_outer = std::sqrt((double)max_disp2_since(pos_snap));
            if (!(2.0 * d_outer <= prune_margin() * 0.98)) { stale = true; return; }
            if (inner_valid) {
                const double d_in = std::sqrt((double)max_disp2_since(pos_snap_in));
                if (!(2.0 * d_in <= skin_in * 0.98)) inner_valid = false;   // the next force pass re-prunes the outer list
            }
        } else {   // lazy_single: one list of radius r_list, snapshot of its build in pos_snap_in
            const double d = std::sqrt((double)max_disp2_since(pos_snap_in));
            if (!(2.0 * d <= skin * 0.98)) stale = true;
            else if (d > 0) export_needs_search = true;    // fine for forces; mhip_export_neighbors wants the list of the new coordinates
        }
        if (stale || !inner_valid) ++n_set_state_refresh;
    }

    void get_state(void* xyz, void* v, int mem_kind) override {
        flush_cm();
        auto one = [&](void* out, int64_t n, const T4* src) {
            if (!out) return;
            T* d = (T*)out;
            if (mem_kind == MHIP_MEM_HOST) { stage_a.reserve(3 * (size_t)n); d = stage_a.p; }
            hipLaunchKernelGGL(k_gather_state<T>, dim3(cdiv(n, 256)), dim3(256), 0, stream, n, (const int32_t*)inv.p, src, d);
            if (mem_kind == MHIP_MEM_HOST) { MHIP_HIP(hipMemcpyAsync(out, d, 3 * (size_t)n * sizeof(T), hipMemcpyDeviceToHost, stream)); MHIP_HIP(hipStreamSynchronize(stream)); }
        };
        one(xyz, n_tot, pos[cur].p);
        one(v, n_owned, vel[cur].p);
        MHIP_HIP(hipGetLastError());
    }

    void forces(int64_t step_n, int accumulate, void* f_xyz, int mem_kind) override {
        cur_dt = 0;   // driven from outside: no time step to bound the drift with
        ensure_built(step_n);
        pass_step = step_n;
        launch_pair_kernel(false);
        frc_valid = false;   // frc holds the pairwise part only
        if (prune_disp_exceeded) { after_forces(step_n); launch_pair_kernel(false); }
        export_frc(accumulate, f_xyz, mem_kind);
    }

    void specific_forces(int accumulate, void* f_xyz, int mem_kind) override {
        if (!state_set) throw ApiError{MHIP_ERR_STATE, "set_state must be called before specific_forces"};
        MHIP_HIP(hipMemsetAsync(frc[cur].p, 0, (size_t)n_tot * sizeof(T4), stream));
        bonded.launch_forces(stream, G, I, pos[cur].p, inv.p, frc[cur].p, orig[cur].p, n_owned, cap);
        frc_valid = false;
        export_frc(accumulate, f_xyz, mem_kind);
    }

    // The energy variant of the pair kernel also writes forces: they go to a scratch array (kept by the context: no allocation per
    // call on the logging path), the forces the integrator carries stay where they are.
    DBuf<T4> frc_scratch; DBuf<unsigned long long> nl_counter;
    T4* frc_override = nullptr;
    void energy_pass() {
        frc_scratch.reserve(cap);
        frc_override = frc_scratch.p;
        try { launch_pair_kernel(true); } catch (...) { frc_override = nullptr; throw; }
        frc_override = nullptr;
    }
    double potential_energy(int64_t step_n) override {
        ensure_built(step_n);
        energy_pass();
        return read_sum(n_blocks);
    }

    // Σ over the pair list of dr ⊗ f (force.jl:848-852, 877-880), ADDED to out9 (row-major 3x3, host doubles).  Runs the energy variant
    // of the pair kernel, whose forces are discarded like those of potential_energy.
    void pairwise_virial(int64_t step_n, double* out9) override {
        ensure_built(step_n);
        energy_pass();
        double v[6];   // six component sums in one launch (one block each), one readback
        hipLaunchKernelGGL(k_sum_double, dim3(6), dim3(256), 0, stream, n_blocks, (const double*)red_part.p + (size_t)n_blocks, red_out.p);
        MHIP_HIP(hipMemcpyAsync(h_red, red_out.p, 6 * sizeof(double), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        for (int c = 0; c < 6; ++c) v[c] = h_red[c];
        out9[0] += v[0]; out9[4] += v[1]; out9[8] += v[2];
        out9[1] += v[3]; out9[3] += v[3]; out9[2] += v[4]; out9[6] += v[4]; out9[5] += v[5]; out9[7] += v[5];
    }

    double sum_run(const double* part, int n) {
        hipLaunchKernelGGL(k_sum_double, dim3(1), dim3(256), 0, stream, n, part, red_out.p);
        MHIP_HIP(hipMemcpyAsync(h_red, red_out.p, sizeof(double), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        return h_red[0];
    }
    // Σ over the specific interactions of r ⊗ f (force.jl:991-1060), ADDED to out9
    void specific_virial(double* out9) override {
        if (!state_set) throw ApiError{MHIP_ERR_STATE, "set_state must be called before specific_virial"};
        const int nb = bonded.launch_virial(stream, G, I, pos[cur].p, inv.p, red_part);
        if (!nb) return;
        for (int c = 0; c < 9; ++c) out9[c] += sum_run((const double*)red_part.p + (size_t)c * nb, nb);
    }
    // reciprocal-space PME virial (recip_conv_inner! ewald.jl:701-723, halved :747-750) + the net-charge term charge_E·I (:925-927), ADDED to out9
    void general_virial(double* out9) override {
        if (!state_set) throw ApiError{MHIP_ERR_STATE, "set_state must be called before general_virial"};
        if (!pme.on()) return;
        const double e = general_potential_energy();           // fills red_part with the 7 component-major runs (and Σq)
        (void)e;
        const int nb = pme.conv_blocks();
        double v[6];
        for (int c = 0; c < 6; ++c) v[c] = 0.5 * sum_run((const double*)red_part.p + (size_t)(c + 1) * nb, nb);
        const double charge_E = pme.charge_factor * pc_sum * pc_sum;
        out9[0] += v[0] + charge_E; out9[4] += v[1] + charge_E; out9[8] += v[2] + charge_E;
        out9[1] += v[3]; out9[3] += v[3]; out9[2] += v[4]; out9[6] += v[4]; out9[5] += v[5]; out9[7] += v[5];
    }

    double specific_potential_energy() override {
        if (!state_set) throw ApiError{MHIP_ERR_STATE, "set_state must be called before specific_potential_energy"};
        int n_part = bonded.launch_energy(stream, G, I, pos[cur].p, inv.p, red_part);
        if (n_part == 0) return 0.0;
        return read_sum(n_part);
    }

    // TriclinicBoundary(v1, v2, v3; approx_images) (spatial.jl:131-220).  cfg.box must hold (v1.x, v2.y, v3.z).  Single domain,
    // systems that fit one tile (every block sees every atom; all distances by the exact in-loop minimum image).
    void set_triclinic(const double* bv9, int32_t approx_images) override {
        if (!(bv9[0] > 0) || bv9[1] != 0 || bv9[2] != 0) throw ApiError{MHIP_ERR_INVALID, "first basis vector must be along the x-axis (no y or z component) and have a positive x component"};
        if (!(bv9[4] > 0) || bv9[5] != 0) throw ApiError{MHIP_ERR_INVALID, "second basis vector must be in the xy plane (no z component) and have a positive y component"};
        if (!(bv9[8] > 0)) throw ApiError{MHIP_ERR_INVALID, "third basis vector must have a positive z component"};
        for (int d = 0; d < 3; ++d) if (!cfg.periodic[d]) throw ApiError{MHIP_ERR_UNSUPPORTED, "TriclinicBoundary is periodic on all three axes"};
        if (n_ghost > 0) throw ApiError{MHIP_ERR_UNSUPPORTED, "TriclinicBoundary: single domain"};
        if (std::fabs(bv9[0] - cfg.box[0]) > 1e-12 * bv9[0] || std::fabs(bv9[4] - cfg.box[1]) > 1e-12 * bv9[4] || std::fabs(bv9[8] - cfg.box[2]) > 1e-12 * bv9[8])
            throw ApiError{MHIP_ERR_INVALID, "the context's box must be (v1.x, v2.y, v3.z) of the triclinic basis"};
        MHIP_HIP(hipStreamSynchronize(stream));
        std::memcpy(tri_bv, bv9, sizeof(tri_bv));
        tri_mode = approx_images ? 1 : 2;
        ljm = ljm_base;                          // the one-type LJ kernels are cubic-only
        if (pme.on()) pme.setup(pme_order_, pme_mesh_, pme_alpha_, cfg.inter.coul_ke, pme_eps_r_, cfg.box, cfg.periodic, tri_bv);   // (PME set before the boundary: its recip_box again)
        setup_grid(); choose_blocking();
        stale = true; frc_valid = false; state_set = false;
    }
    // The boundary of a LIVE context replaced (≙ `sys.boundary = scale_boundary(…)`: scale_coords!, spatial.jl:1184-1218, as the barostats of coupling.jl call it —
    // the reference's force and energy entry points read sys.boundary on every call, ext/MollyCUDAExt.jl:845, 936).  box3: the new side lengths (a TriclinicBoundary:
    // v1.x, v2.y, v3.z); bv9: its basis, required exactly when the context is triclinic (the image mode stays).  Atoms, parameters, exception and bonded lists, the
    // PME mesh and α, the launch shape and the velocities are kept; the cell grid, the capacities and the reciprocal box are made again, every list is
    // dropped, and the coordinates must be handed over again (they were scaled with the box): mhip_set_state.
    void set_box(const double* box3, const double* bv9) override {
        if (n_ghost > 0 || dom.ready || xf.world > 1) throw ApiError{MHIP_ERR_UNSUPPORTED, "set_box: single domain (the bricks of a decomposition are cut from the box: plan again)"};
        for (int d = 0; d < 3; ++d) if (!(box3[d] > 0) || std::isinf(box3[d]) || std::isnan(box3[d])) throw ApiError{MHIP_ERR_INVALID, "box side lengths must be positive and finite"};
        if ((tri_mode != 0) != (bv9 != nullptr)) throw ApiError{MHIP_ERR_INVALID, tri_mode ? "set_box: the context has a TriclinicBoundary, its basis vectors are needed" : "set_box: basis vectors given for a context without a TriclinicBoundary"};
        if (bv9) {
            if (!(bv9[0] > 0) || bv9[1] != 0 || bv9[2] != 0 || !(bv9[4] > 0) || bv9[5] != 0 || !(bv9[8] > 0)) throw ApiError{MHIP_ERR_INVALID, "set_box: v1 along x, v2 in the xy plane, v3.z > 0 (spatial.jl:173-186)"};
            if (std::fabs(bv9[0] - box3[0]) > 1e-12 * bv9[0] || std::fabs(bv9[4] - box3[1]) > 1e-12 * bv9[4] || std::fabs(bv9[8] - box3[2]) > 1e-12 * bv9[8])
                throw ApiError{MHIP_ERR_INVALID, "set_box: the box must be (v1.x, v2.y, v3.z) of the triclinic basis"};
        }
        flush_cm();
        MHIP_HIP(hipStreamSynchronize(stream));
        const mhip_config old_cfg = cfg; double old_bv[9]; std::memcpy(old_bv, tri_bv, sizeof(old_bv));
        for (int d = 0; d < 3; ++d) cfg.box[d] = box3[d];
        if (bv9) std::memcpy(tri_bv, bv9, sizeof(tri_bv));
        try {
            pme.rebox(pme_alpha_, cfg.inter.coul_ke, pme_eps_r_, cfg.box, tri_mode ? tri_bv : nullptr);      // (order, mesh, α stay: only what depends on the box lengths is made again, nothing reallocated)
            setup_grid(); choose_blocking();
            size_t tb2 = 0; MHIP_HIP(exclusive_sum_i32(nullptr, tb2, cell_cnt.p, cell_start.p, 2 * G.ncell + 1, stream));
            if (tb2 + 256 > cub_tmp.n) cub_tmp.reserve(tb2 + 256);
        } catch (...) {      // (a box the engine cannot take — r_list beyond half a side with exact images off, a mesh the PME refuses: the context stays what it was)
            cfg = old_cfg; std::memcpy(tri_bv, old_bv, sizeof(tri_bv));
            pme.rebox(pme_alpha_, cfg.inter.coul_ke, pme_eps_r_, cfg.box, tri_mode ? tri_bv : nullptr);
            setup_grid(); choose_blocking(); stale = true; frc_valid = false; state_set = false;
            throw;
        }
        stale = true; frc_valid = false; frc_run_total = false; frc_before_set_state = false; state_set = false;      // (Σq, Σq² of the PME's constant terms do not depend on the box: pc_valid stays)
        ++n_box_changes;
    }
    int64_t n_box_changes = 0;
    int32_t pme_order_ = 0, pme_mesh_[3] = {0, 0, 0}; double pme_alpha_ = 0, pme_eps_r_ = 1;      // (as last set: a TriclinicBoundary set afterwards rebuilds the reciprocal box)
    void set_pme(int32_t order, const int32_t* mesh, double alpha, double eps_r) override {
        frc_run_total = false;
        MHIP_HIP(hipStreamSynchronize(stream));
        int32_t none[3] = {0, 0, 0};
        pme_order_ = order; pme_alpha_ = alpha; pme_eps_r_ = eps_r; for (int d = 0; d < 3; ++d) pme_mesh_[d] = order ? mesh[d] : 0;
        pme.setup(order, order ? mesh : none, alpha, cfg.inter.coul_ke, eps_r, cfg.box, cfg.periodic, tri_mode ? tri_bv : nullptr);
        frc_valid = false;
    }
    // ≙ AtomsCalculators.forces! of the general interaction (force.jl:792-795): forces added to / written into f_xyz
    void general_forces(int accumulate, void* f_xyz, int mem_kind) override {
        if (!state_set) throw ApiError{MHIP_ERR_STATE, "set_state must be called before general_forces"};
        MHIP_HIP(hipMemsetAsync(frc[cur].p, 0, (size_t)n_tot * sizeof(T4), stream));
        launch_pme_forces();
        frc_valid = false;
        export_frc(accumulate, f_xyz, mem_kind);
    }
    double general_potential_energy() override {
        if (!state_set) throw ApiError{MHIP_ERR_STATE, "set_state must be called before general_potential_energy"};
        if (!pme.on()) return 0.0;
        if (n_ghost > 0) throw ApiError{MHIP_ERR_UNSUPPORTED, "PME runs on a single domain"};
        if (!pc_valid) {   // Σq and Σq² (ewald.jl:917-924) from the charges kept in pos.w
            std::vector<T4> hp(n_owned);
            MHIP_HIP(hipStreamSynchronize(stream));
            MHIP_HIP(hipMemcpy(hp.data(), pos[cur].p, n_owned * sizeof(T4), hipMemcpyDeviceToHost));
            pc_sum = 0; pc_abs2_sum = 0;
            for (int64_t i = 0; i < n_owned; ++i) { pc_sum += (double)hp[i].w; pc_abs2_sum += (double)hp[i].w * (double)hp[i].w; }
            pc_valid = true;
        }
        const int nb = pme.conv_blocks();
        red_part.reserve(7 * (size_t)nb);
        pme.run(stream, n_owned, pos[cur].p, (T4*)nullptr, red_part.p);
        return 0.5 * read_sum(nb) + pme.self_factor * pc_abs2_sum + pme.charge_factor * pc_sum * pc_sum;   // ewald.jl:917-928
    }

    double kinetic_energy() override {
        flush_cm();
        int nb = cdiv(n_owned, 256);
        red_part.reserve(nb);
        hipLaunchKernelGGL(k_ke_partials<T>, dim3(nb), dim3(256), 0, stream, n_owned, (const T4*)vel[cur].p, red_part.p);
        return read_sum(nb);
    }

    void cm_partials_now() {
        int nb = cdiv(n_owned, 256);
        red_part.reserve(4 * (size_t)nb);
        hipLaunchKernelGGL(k_cm_partials<T>, dim3(nb), dim3(256), 0, stream, n_owned, (const T4*)vel[cur].p, (const T*)nullptr, red_part.p);   // callers flush_cm() first
    }

    void remove_cm() override {
        flush_cm();
        cm_partials_now();
        hipLaunchKernelGGL(k_cm_finalize<T>, dim3(1), dim3(256), 0, stream, cdiv(n_owned, 256), (const double*)red_part.p, red_out.p, vcm.p);
        cm_pending = 1; flush_cm();
        MHIP_HIP(hipGetLastError());
    }

    void cm_momentum(double* out4) override {
        flush_cm();
        cm_partials_now();
        hipLaunchKernelGGL(k_cm_finalize<T>, dim3(1), dim3(256), 0, stream, cdiv(n_owned, 256), (const double*)red_part.p, red_out.p, (T*)nullptr);
        MHIP_HIP(hipMemcpyAsync(h_red, red_out.p, 4 * sizeof(double), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        for (int c = 0; c < 4; ++c) out4[c] = h_red[c];
    }

    void shift_velocities(const double* dv3) override {
        flush_cm();
        T h[3] = {T(dv3[0]), T(dv3[1]), T(dv3[2])};
        MHIP_HIP(hipMemcpyAsync(vcm.p, h, 3 * sizeof(T), hipMemcpyHostToDevice, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        cm_pending = 1; flush_cm();
    }

    void cm_momentum_dev(double* out4_dev) override {
        flush_cm();
        cm_partials_now();
        hipLaunchKernelGGL(k_cm_finalize<T>, dim3(1), dim3(256), 0, stream, cdiv(n_owned, 256), (const double*)red_part.p, out4_dev, (T*)nullptr);
        MHIP_HIP(hipGetLastError());
    }
    void remove_cm_dev(const double* total4_dev) override {
        flush_cm();
        // {ΣPx, ΣPy, ΣPz, ΣM} is read as ONE partial by the next k_vv1 (or any state read): total4_dev must stay untouched until then
        cm_ext = total4_dev; n_cm_step = 1; cm_pending = 2;
    }

    void check_finite() override {
        MHIP_HIP(hipMemsetAsync(flags.p + FLAG_NAN, 0, sizeof(int32_t), stream));
        hipLaunchKernelGGL(k_check_finite<T>, dim3(cdiv(n_owned, 256)), dim3(256), 0, stream, n_owned, (const T4*)pos[cur].p, (const T4*)vel[cur].p, (const T4*)frc[cur].p, flags.p);
        MHIP_HIP(hipMemcpyAsync(h_flags, flags.p, N_FLAGS * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        if (h_flags[FLAG_NAN]) throw ApiError{MHIP_ERR_NAN, "NaN or overflow found in coordinates, velocities or forces"};
    }

    // simulators.jl:561-571: wrap (done by set_state / the integrator), neighbours, forces at first_step
    // frc[cur] holds the TOTAL force of the coordinates the engine holds, in the current order (left there by the last step of a run
    // that had no side arrays pending).  A run that continues from exactly that state — the chunked simulate! of test/simulation.jl:16-57,
    // a benchmark's consecutive windows — finds the forces of its first step already there: the reference recomputes them
    // (simulators.jl:564-571), which yields the same numbers from the same lists in the same order.
    bool frc_run_total = false;
    const bool reuse_run_forces = env_int("MOLLYHIP_REUSE_RUN_FORCES", 1) != 0;
    int64_t n_reused_starts = 0;
    void vv_init(int64_t first_step) override {
        if (!state_set) throw ApiError{MHIP_ERR_STATE, "set_state must be called before vv_run"};
        lists_after_set_state();           // (a set_state that brought the SAME coordinates leaves the forces valid)
        const bool had = frc_run_total && frc_valid && reuse_run_forces && !stale && !coords_moved;
        const int64_t reb0 = n_outer, fil0 = n_filters; const bool inner0 = inner_valid;   // (searches re-sort, prunes change the list walked)
        frc_run_total = false;
        start_lists(first_step);
        if (had && n_outer == reb0 && n_filters == fil0 && inner_valid == inner0 && (inner_valid || !dual) && !prune_disp_exceeded) { ++n_reused_starts; return; }
        step_forces(first_step);
        fold_side_forces();
    }
    void vv_stage1(double dt) override {
        if (!frc_valid) throw ApiError{MHIP_ERR_STATE, "vv_stage1 needs forces from vv_init / vv_stage2"};
        cur_dt = dt;
        tr("k_vv1");
        prof.begin(2, stream);
        hipLaunchKernelGGL(k_vv1<T>, dim3(std::min(cdiv(n_owned, 256), 1024)), dim3(256), 0, stream, n_owned, pos[cur].p, vel[cur].p, (const T4*)frc[cur].p, T(dt), T(dt) / T(2),
                           cm_pending == 1 ? (const T*)vcm.p : (const T*)nullptr, cm_pending == 2 ? cm_src() : (const double*)nullptr, n_cm_step, G);
        prof.end(2, stream);
        cm_pending = 0; cm_ext = nullptr;
    }
    void stage2_impl(int64_t step_n, double dt, bool cm, double* cm_parts_ext = nullptr, int n_parts_ext = 0) {
        step_forces(step_n);
        // with an external partial buffer every one of its n_parts_ext slots gets a block (blocks without atoms write zeros)
        const int nb = cm_parts_ext ? n_parts_ext : std::min(cdiv(n_owned, 256), 1024);
        tr("k_vv2");
        prof.begin(2, stream);
        if (cm) {
            hipLaunchKernelGGL((k_vv2<T, true>), dim3(nb), dim3(256), 0, stream, n_owned, vel[cur].p, frc[cur].p, T(dt) / T(2), cm_parts_ext ? cm_parts_ext : cm_step.p, pend_a);
            cm_pending = 2; n_cm_step = nb;   // the next k_vv1 (or any flush) re-sums the partials: no finalize launch
        } else {
            hipLaunchKernelGGL((k_vv2<T, false>), dim3(nb), dim3(256), 0, stream, n_owned, vel[cur].p, frc[cur].p, T(dt) / T(2), (double*)nullptr, pend_a);
        }
        prof.end(2, stream);
        pend_a = nullptr;    // the kick wrote the total back into frc[cur]
    }
    // the neighbour cadence of a stepwise-driven run: as in vv_run.  A ghosted sub-domain without the dual list is re-planned
    // (set_atom_counts / set_state → stale) by the host at every rebuild step instead.
    void stage2_cadenced(int64_t step_n, double dt, bool cm, double* cm_parts_ext = nullptr, int n_parts_ext = 0) {
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        resolve_track(step_n);
        const bool due = check_due(step_n, every) && step_n != last_build_step && (n_ghost == 0 || dual);
        if (due && dual) refresh(step_n);
        stage2_impl(step_n, dt, cm, cm_parts_ext, n_parts_ext);
        if (due && !dual) refresh(step_n);
    }
    void vv_stage2(int64_t step_n, double dt) override {
        stage2_cadenced(step_n, dt, false);
        MHIP_HIP(hipGetLastError());
    }

    // the same with this rank's Σ m v left as n_parts per-block partials {Px, Py, Pz, M} in cm_parts_dev (no finalize launch): the
    // host all-reduces the whole array and hands it back through remove_cm_parts_dev, where the next first kick re-sums it
    void halo_end_parts(int64_t step_n, double dt, int64_t first, int64_t n, const void* in_dev, double* cm_parts_dev, int32_t n_parts) override {
        if (cm_parts_dev && (n_parts < 1 || n_parts > 1024)) throw ApiError{MHIP_ERR_INVALID, "n_parts must be 1..1024"};
        scatter_coords(first, n, in_dev);
        stage2_cadenced(step_n, dt, cm_parts_dev != nullptr, cm_parts_dev, n_parts);
        if (cm_parts_dev) { cm_pending = 0; cm_ext = nullptr; }
        MHIP_HIP(hipGetLastError());
    }
    void remove_cm_parts_dev(const double* total_parts_dev, int32_t n_parts) override {
        if (n_parts < 1 || n_parts > 1024) throw ApiError{MHIP_ERR_INVALID, "n_parts must be 1..1024"};
        flush_cm();
        cm_ext = total_parts_dev; n_cm_step = n_parts; cm_pending = 2;
    }

    void set_ghost_margin(double m) override {
        if (!(m >= 0) || std::isinf(m)) throw ApiError{MHIP_ERR_INVALID, "ghost margin must be finite and >= 0"};
        ghost_margin = m; stale = true; host_prune = m > 0;
        if (n_ghost > 0) { setup_grid(); choose_blocking(); }
    }
    // max |x − x_plan|² over owned and ghost atoms since the outer search of the current ghost plan, as a float in device memory
    // (ready for a MAX all-reduce over the ranks); +inf when this sub-domain has to be re-planned at every rebuild step anyway
    // out[0] = max |x − x_plan|² over owned and ghost atoms since the outer search of the current ghost plan, out[1] = the same since
    // the last prune of the inner list — floats in device memory, ready for a MAX all-reduce over the ranks.  +inf in out[0] when this
    // sub-domain has to be re-planned at every rebuild step anyway.
    void plan_disp2_dev(float* out_dev) override {
        const uint32_t inf_bits = 0x7f800000u;
        if (!dual || stale) { MHIP_HIP(hipMemsetD32Async((hipDeviceptr_t)out_dev, (int)inf_bits, 2, stream)); return; }
        MHIP_HIP(hipMemsetAsync(out_dev, 0, 2 * sizeof(float), stream));
        const dim3 g(std::min(cdiv(n_tot, 256), 1024));
        hipLaunchKernelGGL(k_max_disp<T>, g, dim3(256), 0, stream, n_tot, (const T4*)pos[cur].p, (const T4*)pos_snap.p, reinterpret_cast<unsigned int*>(out_dev), G);
        if (inner_valid) hipLaunchKernelGGL(k_max_disp<T>, g, dim3(256), 0, stream, n_tot, (const T4*)pos[cur].p, (const T4*)pos_snap_in.p, reinterpret_cast<unsigned int*>(out_dev) + 1, G);
        else MHIP_HIP(hipMemsetD32Async((hipDeviceptr_t)(out_dev + 1), (int)inf_bits, 1, stream));
        MHIP_HIP(hipGetLastError());
    }
    void request_prune() override { inner_valid = false; }
    // {max |x − x_plan|², max |x − x_prune|², max |v|² of the owned atoms} as floats in device memory, ready for ONE MAX all-reduce over
    // the ranks; mhip_plan_decide takes the reduced triple.  +inf in [0]: this sub-domain is re-planned at every rebuild step anyway.
    void plan_state_dev(float* out_dev) override {
        const uint32_t inf_bits = 0x7f800000u;
        MHIP_HIP(hipMemsetAsync(out_dev, 0, 3 * sizeof(float), stream));
        if (!dual || stale) { MHIP_HIP(hipMemsetD32Async((hipDeviceptr_t)out_dev, (int)inf_bits, 2, stream)); return; }
        const dim3 g(std::min(cdiv(n_owned, 1024), 512));
        unsigned int* w = reinterpret_cast<unsigned int*>(out_dev);
        // owned atoms only: every ghost is an owned atom of some rank, and the MAX runs over all of them (the call may come before
        // this step's ghost coordinates are in)
        hipLaunchKernelGGL(k_max_disp<T>, g, dim3(1024), 0, stream, n_owned, (const T4*)pos[cur].p, (const T4*)pos_snap.p, w, G, (const T4*)vel[cur].p, n_owned, w + 2,
                           inner_valid ? (const T4*)pos_snap_in.p : (const T4*)nullptr, inner_valid ? w + 1 : (unsigned int*)nullptr);
        if (!inner_valid) MHIP_HIP(hipMemsetD32Async((hipDeviceptr_t)(out_dev + 1), (int)inf_bits, 1, stream));
        MHIP_HIP(hipGetLastError());
    }
    // The collective decision at the rebuild cadence, made by every rank from the same reduced numbers with the criteria of the
    // single-domain engine (refresh): 0 = the inner list lives on, 1 = the next force pass re-prunes the outer list, 2 = the ghost
    // plan cannot vouch for a prune any more (or there is no margin): re-plan.  The inner list is pruned with the tight inner skin
    // (rc + skin_in, grown when the fastest atoms would outrun a third of it between two checks) as in mhip_vv_run.
    int plan_decide(int64_t step_n, const float* red3, int32_t* check_in) override { return plan_decide_late(step_n, red3, check_in, 0); }
    // late: the decision is applied `late` steps after the measurement (mhip_domain_run reads the reduced numbers one step later, so
    // that nothing waits for them): the drift bounds then reach that much further
    int plan_decide_late(int64_t step_n, const float* red3, int32_t* check_in, int late) {
        if (check_in) *check_in = 0;
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        if (!engine_sched) {   // first use: from now on the inner list may be tighter than r_list
            engine_sched = true;
            skin_in = std::min(skin, std::max(1, env_int("MOLLYHIP_INNER_SKIN_PM", 100)) * 1e-3);
            const T rp = T(rc_max_ + skin_in);
            r_prune2 = (skin_in < skin) ? rp * rp : r_in2;
        }
        if (!dual || stale || std::isinf(red3[0])) return 2;
        prev_vmax = last_vmax; last_vmax = std::sqrt((double)red3[2]);
        ++n_disp_checks;
        bool reprune = !inner_valid || std::isinf(red3[1]);
        if (!reprune) {
            // (measured at step_n, applied `late` steps later; the list is walked up to the pass of step_n + every — the next check's own step — as in resolve_track)
            const double d = std::sqrt((double)red3[1]), ahead = drift_ahead(d, step_n - last_prune_step, every + std::max(late - 1, 0));
            adapt_inner_skin(ahead);
            reprune = !inner_valid || 2.0 * (d + ahead) > skin_in * 0.98;
            // not good for a whole interval, but for k steps: the host looks again then (it owns the step loop)
            if (reprune && inner_valid && check_in)
                if (const int k = steps_within(d, 0.49 * skin_in, step_n - last_prune_step, every, true)) { if (k > late + 1) { *check_in = k; return 0; } }
        }
        if (!reprune) return 0;
        // the prune runs inside the NEXT force pass, one step from now (+ late): leave it that much headroom
        if (2.0 * (std::sqrt((double)red3[0]) + last_vmax * cur_dt * 1.25 * (1 + late)) > prune_margin() * 0.98) return 2;
        inner_valid = false;
        return 1;
    }

    // ---- one MD step of a ghosted sub-domain in ONE call after the ghost exchange (fused integrator, Σ m v on the ghost message) --------
    mhip_halo_plan hp{}; bool hp_set = false, halo_cm_in = false; int cm_half = 0;
    DBuf<double> cm_all;         // [0] this rank's {ΣPx, ΣPy, ΣPz, ΣM} of the step before (k_halo_pack), [1 + p] peer p's (k_halo_unpack)
    void set_halo_plan(const mhip_halo_plan* p) override {
        if (!p) { hp_set = false; return; }
        if (p->n_recv_rows < 0 || p->n_send_rows < 0 || p->n_cm_peers < 0 || p->n_cm_peers > 26 || p->cm_rows < 0 || p->cm_rows > 4 || p->n_send_cm < 0)
            throw ApiError{MHIP_ERR_INVALID, "halo plan: counts out of range"};
        if ((p->n_recv_rows > 0 && (!p->recv || !p->recv_dst)) || (p->n_send_rows > 0 && (!p->send || !p->send_idx || !p->send_shift)) || (p->n_send_cm > 0 && !p->send_cm_pos))
            throw ApiError{MHIP_ERR_INVALID, "halo plan: null buffer"};
        if (p->cm_rows > 0 && p->cm_rows * 3 * (int)sizeof(T) < 32) throw ApiError{MHIP_ERR_INVALID, "halo plan: cm_rows rows cannot hold four doubles"};
        hp = *p; hp_set = true; halo_cm_in = false; xf.routes = false; hx.plan_ok = false;
        cm_all.reserve(4 * 28);
        MHIP_HIP(hipMemsetAsync(cm_all.p, 0, 4 * 28 * sizeof(double), stream));
    }
    void halo_pack(bool with_cm) {
        if (hp.n_send_rows <= 0 && !with_cm) return;
        // (after an in-engine re-plan the tables are the engine's own and there is no staging buffer: only mhip_domain_run's direct stores can carry them)
        if (!(xf_direct && xf.n_peers > 0) && (hp.n_send_rows > 0 || with_cm) && !hp.send) throw ApiError{MHIP_ERR_STATE, "the current ghost plan was made inside the engine (mhip_set_domain) and has no send buffer: step it with mhip_domain_run, or hand a plan with buffers to mhip_set_halo_plan"};
        tr("k_halo_pack");
        XferSend X{};        // inside mhip_domain_run with peers: the rows go straight into the peers' regions, exchange number ++seq
        if (xf_direct && xf.n_peers > 0) {
            ++xf.seq;
            X.row_peer = xf.row_peer.p; X.row_dst = xf.row_dst.p; X.P = xf.peers; X.rows_cap = xf.rows_cap; X.parity = (int)(xf.seq & 1u); X.seq = xf.seq; X.my_rank = xf.rank;
            X.peers = xf.d_peers.p; X.n_peers = xf.n_peers; X.done = xf.done.p;
        }
        hipLaunchKernelGGL(k_halo_pack<T>, dim3(cdiv(std::max<int64_t>(hp.n_send_rows, 1), 256) + 1), dim3(256), 0, stream, hp.n_send_rows, hp.send_idx, (const T*)hp.send_shift, (const int32_t*)inv.p,
                           (const T4*)pos[cur].p, (T*)hp.send, with_cm ? (const double*)cm_step.p : (const double*)nullptr, n_cm_step, hp.send_cm_pos, hp.n_send_cm, std::max(hp.cm_rows, 1), cm_all.p, X);
        MHIP_HIP(hipGetLastError());
    }
    // first kick + drift + pack: after vv_init, after a step that stopped behind its second kick, after a re-plan
    void halo_start(double dt) override {
        if (!hp_set) throw ApiError{MHIP_ERR_STATE, "mhip_set_halo_plan first"};
        vv_stage1(dt);
        halo_pack(false);
        halo_cm_in = false;
    }
    // flags bit 0: this step removes the centre-of-mass motion; bit 1: stop behind the second kick (the step's Σ m v goes to
    // cm_parts_dev as n_parts per-block partials for an all-reduce, nothing is packed) — at the rebuild cadence and at the end of a run
    void halo_mid(int64_t step_n, double dt, int32_t flags, double* cm_parts_dev, int32_t n_parts) override {
        if (!hp_set) throw ApiError{MHIP_ERR_STATE, "mhip_set_halo_plan first"};
        const bool cm = (flags & 1) != 0, last = (flags & 2) != 0;
        if (cm && last && (!cm_parts_dev || n_parts < 1 || n_parts > 1024)) throw ApiError{MHIP_ERR_INVALID, "n_parts must be 1..1024"};
        if (cm && !last && hp.cm_rows <= 0 && hp.n_cm_peers > 0) throw ApiError{MHIP_ERR_INVALID, "halo plan carries no centre-of-mass rows"};
        if (hp.n_recv_rows > 0) {
            if (hp.first_ghost < 0 || hp.first_ghost > n_tot) throw ApiError{MHIP_ERR_INVALID, "halo plan: ghost range out of bounds"};
            if (!(xf_direct && xf.n_peers > 0) && !hp.recv) throw ApiError{MHIP_ERR_STATE, "the current ghost plan was made inside the engine (mhip_set_domain) and has no receive buffer: step it with mhip_domain_run, or hand a plan with buffers to mhip_set_halo_plan"};
            tr("k_halo_unpack");
            XferWait W{};    // inside mhip_domain_run with peers: wait for exchange xf.seq, read my region's half
            if (xf_direct && xf.n_peers > 0) { W.mine = reinterpret_cast<const XferHeader*>(xf.region); W.parity = (int)(xf.seq & 1u); W.seq = xf.seq; W.peers = xf.d_peers.p; W.n_peers = xf.n_peers; W.err = xf.err.p; W.ticks = xf_ticks(); }
            hipLaunchKernelGGL(k_halo_unpack<T>, dim3(cdiv(hp.n_recv_rows, 256)), dim3(256), 0, stream, hp.n_recv_rows, W.n_peers > 0 ? (const T*)xf_rows(W.parity) : (const T*)hp.recv, hp.recv_dst, hp.first_ghost, (const int32_t*)inv.p,
                               pos[cur].p, cm_all.p, std::max(hp.cm_rows, 1), W);
        }
        cur_dt = dt;
        if (replan_now) { replan_now = false; device_replan(step_n); }            // (mhip_domain_run: ownership, ghosts and the outer list redone here, in front of the step's force pass)
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        const bool due = check_due(step_n, every) && step_n != last_build_step && (n_ghost == 0 || dual);
        if (due && dual) refresh(step_n);
        const bool solo = hp.n_cm_peers == 0 && hp.n_send_rows == 0;              // no peers: the partials of the launch before are the whole sum
        // … which workgroup 0 of the pair pass in between adds up into ONE partial, as inside mhip_vv_run (ForceArgs::cm_fin_in): the integrator's blocks
        // then do not each re-sum hundreds of partials first
        cm_fin_solo_src = (solo && halo_cm_in && n_ghost == 0 && n_cm_step > 1 && n_cm_step <= 4096) ? (const double*)cm_step.p + (size_t)(cm_half ^ 1) * 4 * 1024 : (const double*)nullptr;
        cm_fin_solo_done = false;
        step_forces(step_n);
        cm_fin_solo_src = nullptr;
        if (cm_pending) flush_cm();                                               // (a removal registered through the stepwise entry points)
        const double* cm_in = halo_cm_in ? (solo ? (cm_fin_solo_done ? (const double*)cm_fin_buf.p : (const double*)cm_step.p + (size_t)(cm_half ^ 1) * 4 * 1024) : (const double*)cm_all.p) : (const double*)nullptr;
        const int n_in = solo ? (cm_fin_solo_done ? 1 : n_cm_step) : 1 + hp.n_cm_peers;
        // (block count: mhip_vv_run's — fewer, longer blocks at these sizes, see there)
        const int nb = (cm && last) ? n_parts : std::min(cdiv(n_owned, 256), solo ? (int)std::max<int64_t>(256, std::min<int64_t>(512, n_owned / 2048)) : 1024);
        double* cm_out = cm ? (last ? cm_parts_dev : cm_step.p + (size_t)(solo ? cm_half : 0) * 4 * 1024) : (double*)nullptr;
        prof.begin(2, stream);
        tr("k_vv_mid");
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, stream, n_owned, pos[cur].p, vel[cur].p, (const T4*)frc[cur].p, T(dt), T(dt) / T(2),
                               cm_in, n_in, cm_out, (const T4*)pend_a, G, (const T4*)nullptr, (const T4*)nullptr, (float*)nullptr);
        };
        if (last) { if (cm) go(k_vv_mid<T, true, true>); else go(k_vv_mid<T, false, true>); }
        else { if (cm) go(k_vv_mid<T, true, false>); else go(k_vv_mid<T, false, false>); }
        prof.end(2, stream);
        pend_a = nullptr; cm_pending = 0; cm_ext = nullptr;
        if (due && !dual) refresh(step_n);
        if (!last) {
            n_cm_step = nb;
            if (solo) cm_half ^= 1; else halo_pack(cm);
            halo_cm_in = cm;
            frc_valid = false;                                                    // frc[cur] belongs to the coordinates before the drift
        } else halo_cm_in = false;
        MHIP_HIP(hipGetLastError());
    }
    // ---- the ghost exchange inside the engine (halo_xfer.h): peer stores into IPC-mapped receive regions, the step loop in C++ --------
    struct Xfer {
        unsigned char* region = nullptr; int64_t rows_cap = 0; int world = 0, rank = 0;
        XferPeers peers{}; bool opened[XFER_MAX_RANKS] = {}; int64_t peer_cap[XFER_MAX_RANKS] = {};   // peer_cap: rows per half of each peer's region (its header says)
        bool routes = false; int n_peers = 0; std::vector<int32_t> peer_rank;
        uint64_t dev_key = 0; bool shared_device = false;      // some peer runs on THIS device (several ranks on one GPU: the test set-up)
        DBuf<int32_t> row_peer, row_dst, d_peers; DBuf<unsigned int> done; DBuf<int32_t> err; DBuf<float> mine3, red3;
        uint32_t seq = 0, plan_seq = 0;
        float* h_red3 = nullptr; int32_t* h_err = nullptr; hipEvent_t ev_plan = nullptr;
        bool plan_pending = false; int64_t plan_step = -1, next_check = -1, plan_prune_id = -1, plan_outer_id = -1;      // (ids: the running counts of prunes / outer searches when the check was issued)
        RpPlanPtrs plan{}; uint32_t rp_seq = 0;      // every rank's plan area; number of the last re-plan made inside the engine (replan.h)
    } xf;
    bool xf_direct = false;      // inside mhip_domain_run: k_halo_pack stores into the peers' regions, k_halo_unpack waits for theirs
    T* xf_rows(int parity) const { return reinterpret_cast<T*>(xf.region + XFER_ROWS_OFF) + (size_t)parity * xf.rows_cap * 3; }
    void xf_release() {
        for (int r = 0; r < XFER_MAX_RANKS; ++r) if (xf.opened[r] && xf.peers.region[r]) { (void)hipIpcCloseMemHandle(xf.peers.region[r]); xf.opened[r] = false; }
        if (xf.region) (void)hipFree(xf.region);
        xf.region = nullptr;
        xf.row_peer.release(); xf.row_dst.release(); xf.d_peers.release(); xf.done.release(); xf.err.release(); xf.mine3.release(); xf.red3.release();
        if (xf.h_red3) (void)hipHostFree(xf.h_red3); if (xf.h_err) (void)hipHostFree(xf.h_err); if (xf.ev_plan) (void)hipEventDestroy(xf.ev_plan);
        xf.h_red3 = nullptr; xf.h_err = nullptr; xf.ev_plan = nullptr;
    }
    // this rank's receive region: two halves of rows_cap rows of 3 reals behind the header, fine-grained device memory (peers write it,
    // this device polls it); its IPC handle goes to every peer
    void halo_region(int64_t rows_cap, int32_t world, int32_t my_rank, void* handle_out) override {
        if (rows_cap <= 0 || world < 1 || world > XFER_MAX_RANKS || my_rank < 0 || my_rank >= world) throw ApiError{MHIP_ERR_INVALID, "halo region: rows / world / rank out of range"};
        MHIP_HIP(hipStreamSynchronize(stream));
        if (!xf.region || xf.rows_cap < rows_cap || xf.world != world || xf.rank != my_rank) {
            if (xf.region) xf_release();
            const size_t bytes = xfer_region_bytes<T>(rows_cap);      // header | two row halves | plan area (replan.h)
            MHIP_HIP(hipExtMallocWithFlags((void**)&xf.region, bytes, hipDeviceMallocFinegrained));
            MHIP_HIP(hipMemset(xf.region, 0, bytes));
            MHIP_HIP(hipMemcpy(xf.region + offsetof(XferHeader, rows_cap), &rows_cap, sizeof(int64_t), hipMemcpyHostToDevice));
            {
                hipDeviceProp_t pr; MHIP_HIP(hipGetDeviceProperties(&pr, device));
                xf.dev_key = (((uint64_t)(uint32_t)pr.pciDomainID << 32) | ((uint64_t)(uint32_t)pr.pciBusID << 16) | (uint64_t)(uint32_t)pr.pciDeviceID) + 1u;
                MHIP_HIP(hipMemcpy(xf.region + offsetof(XferHeader, dev_key), &xf.dev_key, sizeof(uint64_t), hipMemcpyHostToDevice));
                xf.shared_device = false;
            }
            xf.rows_cap = rows_cap; xf.world = world; xf.rank = my_rank; xf.seq = 0; xf.plan_seq = 0; xf.routes = false;
            for (int r = 0; r < XFER_MAX_RANKS; ++r) xf.peers.region[r] = nullptr;
            xf.peers.region[my_rank] = xf.region; xf.peer_cap[my_rank] = rows_cap;
            for (int r = 0; r < XFER_MAX_RANKS; ++r) xf.plan.area[r] = nullptr;
            xf.plan.area[my_rank] = xf.region + xfer_plan_off<T>(rows_cap);
            xf.done.reserve(1); xf.err.reserve(4); xf.mine3.reserve(4); xf.red3.reserve(4);
            MHIP_HIP(hipMemset(xf.done.p, 0, sizeof(unsigned int))); MHIP_HIP(hipMemset(xf.err.p, 0, 4 * sizeof(int32_t)));
            if (!xf.h_red3) MHIP_HIP(hipHostMalloc((void**)&xf.h_red3, 4 * sizeof(float)));
            if (!xf.h_err) MHIP_HIP(hipHostMalloc((void**)&xf.h_err, 4 * sizeof(int32_t)));
            if (!xf.ev_plan) MHIP_HIP(hipEventCreateWithFlags(&xf.ev_plan, hipEventDisableTiming));
        }
        if (handle_out) {
            hipIpcMemHandle_t h;
            MHIP_HIP(hipIpcGetMemHandle(&h, xf.region));
            static_assert(sizeof(hipIpcMemHandle_t) == MHIP_IPC_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
            std::memcpy(handle_out, &h, sizeof(h));
        }
    }
    void halo_open_peer(int32_t rank, const void* handle) override {
        if (!xf.region) throw ApiError{MHIP_ERR_STATE, "mhip_halo_region first"};
        if (rank < 0 || rank >= xf.world || !handle) throw ApiError{MHIP_ERR_INVALID, "halo peer: rank out of range or null handle"};
        if (rank == xf.rank) return;
        if (xf.opened[rank]) { (void)hipIpcCloseMemHandle(xf.peers.region[rank]); xf.opened[rank] = false; }
        hipIpcMemHandle_t h; std::memcpy(&h, handle, sizeof(h));
        void* base = nullptr;
        MHIP_HIP(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
        xf.peers.region[rank] = (unsigned char*)base; xf.opened[rank] = true;
        // the peer may have been created with another capacity than this rank: its own header is the authority on what fits there
        MHIP_HIP(hipMemcpy(&xf.peer_cap[rank], (unsigned char*)base + offsetof(XferHeader, rows_cap), sizeof(int64_t), hipMemcpyDeviceToHost));
        xf.plan.area[rank] = (unsigned char*)base + xfer_plan_off<T>(xf.peer_cap[rank]);
        uint64_t key = 0;
        MHIP_HIP(hipMemcpy(&key, (unsigned char*)base + offsetof(XferHeader, dev_key), sizeof(uint64_t), hipMemcpyDeviceToHost));
        if (key != 0 && key == xf.dev_key) xf.shared_device = true;
    }
    // where the rows of the current ghost plan travel: consecutive segments of the send buffer → (peer, first row in the peer's half)
    void set_halo_routes(const mhip_halo_routes* rt) override {
        if (!rt) { xf.routes = false; return; }
        if (!hp_set) throw ApiError{MHIP_ERR_STATE, "mhip_set_halo_plan first"};
        if (rt->n_peers < 0 || rt->n_peers >= XFER_MAX_RANKS) throw ApiError{MHIP_ERR_INVALID, "halo routes: peer count out of range"};
        if (rt->n_peers > 0 && !xf.region) throw ApiError{MHIP_ERR_STATE, "mhip_halo_region first"};
        std::vector<int32_t> rp, rd; rp.reserve((size_t)hp.n_send_rows); rd.reserve((size_t)hp.n_send_rows);
        int64_t recv_total = 0;
        xf.peer_rank.assign(rt->peer_rank, rt->peer_rank + rt->n_peers);
        for (int q = 0; q < rt->n_peers; ++q) {
            const int r = rt->peer_rank[q];
            if (r < 0 || r >= xf.world || r == xf.rank || !xf.peers.region[r]) throw ApiError{MHIP_ERR_INVALID, "halo routes: peer not opened (mhip_halo_open_peer)"};
            if (rt->send_rows[q] < 0 || rt->dst_row[q] < 0 || rt->dst_row[q] + rt->send_rows[q] > xf.peer_cap[r]) throw ApiError{MHIP_ERR_CAPACITY, "halo routes: segment does not fit the peer's region"};
            for (int64_t k = 0; k < rt->send_rows[q]; ++k) { rp.push_back(r); rd.push_back((int32_t)(rt->dst_row[q] + k)); }
            recv_total += rt->recv_rows[q];
        }
        if ((int64_t)rp.size() != hp.n_send_rows || recv_total != hp.n_recv_rows || recv_total > xf.rows_cap) throw ApiError{MHIP_ERR_INVALID, "halo routes do not match the halo plan"};
        xf.n_peers = rt->n_peers;
        MHIP_HIP(hipStreamSynchronize(stream));
        xf.row_peer.reserve(std::max<size_t>(rp.size(), 1)); xf.row_dst.reserve(std::max<size_t>(rd.size(), 1)); xf.d_peers.reserve(std::max(rt->n_peers, 1));
        if (!rp.empty()) { MHIP_HIP(hipMemcpy(xf.row_peer.p, rp.data(), rp.size() * sizeof(int32_t), hipMemcpyHostToDevice)); MHIP_HIP(hipMemcpy(xf.row_dst.p, rd.data(), rd.size() * sizeof(int32_t), hipMemcpyHostToDevice)); }
        if (rt->n_peers) MHIP_HIP(hipMemcpy(xf.d_peers.p, rt->peer_rank, rt->n_peers * sizeof(int32_t), hipMemcpyHostToDevice));
        xf.routes = true; xf.plan_pending = false; xf.next_check = -1; hx.plan_ok = false;
    }
    // One round of the collective number exchange over the mapped regions, before any step depends on it: every rank stores a token
    // into every rank's table and waits (bounded) for all of theirs.  0 = some rank's store did not become visible here within 2 s —
    // the host then keeps its own loop with torch.distributed collectives (every rank must call this at the same point).
    int halo_selftest() override {
        if (!xf.region) throw ApiError{MHIP_ERR_STATE, "mhip_halo_region first"};
        for (int r = 0; r < xf.world; ++r) if (!xf.peers.region[r]) throw ApiError{MHIP_ERR_STATE, "mhip_halo_open_peer for every rank first"};
        const float token[4] = {(float)(xf.rank + 1), 0.f, 0.f, 0.f};
        MHIP_HIP(hipMemcpyAsync(xf.mine3.p, token, 3 * sizeof(float), hipMemcpyHostToDevice, stream));
        ++xf.plan_seq;
        hipLaunchKernelGGL(k_plan_push, dim3(1), dim3(64), 0, stream, (const float*)xf.mine3.p, xf.peers, xf.world, xf.rank, (int)(xf.plan_seq & 1u), xf.plan_seq);
        hipLaunchKernelGGL(k_plan_reduce, dim3(1), dim3(64), 0, stream, reinterpret_cast<const XferHeader*>(xf.region), xf.world, (int)(xf.plan_seq & 1u), xf.plan_seq, xf.red3.p, xf.h_red3, xf.err.p, xf_ticks());
        MHIP_HIP(hipMemcpyAsync(xf.h_err, xf.err.p, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        const bool ok = *xf.h_err == 0 && xf.h_red3[0] == (float)xf.world;      // the MAX of the tokens is the highest rank's
        if (*xf.h_err) MHIP_HIP(hipMemset(xf.err.p, 0, 4 * sizeof(int32_t)));
        return ok ? 1 : 0;
    }
    // bound of every in-kernel wait for a peer, in ticks of the 100 MHz wall clock (MOLLYHIP_XFER_TIMEOUT_MS; 2 s unless set: a peer may be
    // held up on its host by a capacity-retry rebuild, a profiler or first-use code loading)
    static unsigned long long xf_ticks() {
        static const unsigned long long t = (unsigned long long)std::max(1, env_int("MOLLYHIP_XFER_TIMEOUT_MS", 2000)) * 100000ull;
        return t;
    }
    // what a wait that gave up recorded (halo_xfer.h, xfer_wait): which kernel waited for which rank, for which sequence number, and what it last saw
    static std::string wait_report(const int32_t* e) {
        if (!e[1]) return "";
        static const char* const what[] = {"?", "the ghost rows (k_halo_unpack)", "the validity triple (k_plan_reduce)", "the migration counts", "the ghost counts", "the migrating atoms", "the new ghosts",
                                           "the ghost rows (fused step, a block with ghosts)", "the centre-of-mass rows (fused step, head workgroup)", "the peer's read of the half before (fused step, a sending block without ghosts)"};
        const int who = e[1] - 1, k = (who >> 8) & 0xff;
        return std::string(": waited for ") + what[k < 10 ? k : 0] + " of rank " + std::to_string(who & 0xff) + ", sequence number " + std::to_string((uint32_t)e[2]) + ", last seen " + std::to_string((uint32_t)e[3]);
    }
    void xf_check_errors() {
        if (!xf.region) return;
        MHIP_HIP(hipMemcpyAsync(xf.h_err, xf.err.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        if (*xf.h_err) {
            const std::string rep = wait_report(xf.h_err);
            MHIP_HIP(hipMemset(xf.err.p, 0, 4 * sizeof(int32_t)));
            throw ApiError{MHIP_ERR_STATE, "ghost exchange timed out: a peer's rows (or its validity triple) did not arrive in time (MOLLYHIP_XFER_TIMEOUT_MS)" + rep + "; this rank is at exchange " + std::to_string(xf.seq)};
        }
    }
    // the collective validity check of the pair lists, issued at step s and read one step later (nothing waits for it)
    void xf_issue_plan_check(int64_t s) {
        if (hx.trk_step == s && dual && inner_valid && !stale) {      // measured by the fused launch that made these coordinates (per-block maxima, kernels.h STEP epilogue)
            hipLaunchKernelGGL(k_track_reduce, dim3(1), dim3(256), 0, stream, step_parts, (const float*)trk_part.p, xf.mine3.p, (float*)nullptr, 1);
            MHIP_HIP(hipGetLastError());
        } else plan_state_dev(xf.mine3.p);
        hx.trk_step = -1;
        ++xf.plan_seq;
        if (xf.world > 1) {
            hipLaunchKernelGGL(k_plan_push, dim3(1), dim3(64), 0, stream, (const float*)xf.mine3.p, xf.peers, xf.world, xf.rank, (int)(xf.plan_seq & 1u), xf.plan_seq);
            hipLaunchKernelGGL(k_plan_reduce, dim3(1), dim3(64), 0, stream, reinterpret_cast<const XferHeader*>(xf.region), xf.world, (int)(xf.plan_seq & 1u), xf.plan_seq, xf.red3.p, xf.h_red3, xf.err.p, xf_ticks());
        } else MHIP_HIP(hipMemcpyAsync(xf.h_red3, xf.mine3.p, 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipEventRecord(xf.ev_plan, stream));
        xf.plan_pending = true; xf.plan_step = s; xf.plan_prune_id = n_filters; xf.plan_outer_id = n_outer;
    }
    // A run of ghosted steps in ONE call (≙ DomainRun.run of domain.py, fused form): returns after n_steps (reason 0) or behind the second
    // kick of the step after which ownership and ghosts have to be re-planned (reason 1); *steps_done steps were taken.  When the last
    // step taken removes the centre-of-mass motion, cm_parts_dev holds its n_parts partials for the all-reduce over the ranks
    // (mhip_remove_cm_parts_dev), as after mhip_vv_halo_mid with the stop flag.  counters[0..2] += checks, prunes arranged, re-plans asked.
    void domain_run(int64_t first_step, int64_t n_steps, double dt, int32_t remove_cm_every, double* cm_parts_dev, int32_t n_parts, int64_t* steps_done, int32_t* reason, int64_t* counters) override {
        if (!hp_set) throw ApiError{MHIP_ERR_STATE, "mhip_set_halo_plan first"};
        const bool solo = hp.n_cm_peers == 0 && hp.n_send_rows == 0;
        if (!solo && !xf.routes) throw ApiError{MHIP_ERR_STATE, "mhip_set_halo_routes first"};
        if (!xf.h_red3) { MHIP_HIP(hipHostMalloc((void**)&xf.h_red3, 4 * sizeof(float))); MHIP_HIP(hipEventCreateWithFlags(&xf.ev_plan, hipEventDisableTiming)); xf.mine3.reserve(4); xf.world = std::max(xf.world, 1); }
        *steps_done = 0; *reason = 0;
        if (n_steps <= 0) return;
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        const int64_t last = first_step + n_steps;
        InRun guard_in_run(in_run);
        if (solo && n_ghost == 0 && xf.world <= 1) {
            // One brick: nobody to exchange with, no ghosts, a re-plan IS an outer-list rebuild — the step loop of mhip_vv_run (fused steps, checks taken inside the
            // launches), with the lists kept by the engine's own criteria.  The run's last Σ m v goes to cm_parts_dev for the caller's (one-rank) sum.
            if (host_prune) {      // (set_ghost_margin handed the prune decisions to a host that has no peers to agree with)
                host_prune = false; engine_sched = true;
                skin_in = std::min(skin, std::max(skin_in_adapted, std::max(1, env_int("MOLLYHIP_INNER_SKIN_PM", 100)) * 1e-3));
                const T rp = T(rc_max_ + skin_in);
                r_prune2 = (skin_in < skin) ? rp * rp : r_in2;
                inner_valid = false;
            }
            cur_dt = dt;
            const int64_t c0 = n_disp_checks, f0 = n_filters, o0 = n_outer;
            xf.plan_pending = false;
            vv_loop(first_step, n_steps, dt, remove_cm_every, cm_parts_dev, n_parts);
            flush_cm();
            *steps_done = n_steps;
            // (checks: those read so far + the one a launch has measured and the next call's first step will read)
            if (counters) { counters[0] += n_disp_checks - c0 + (trk_issued ? 1 : 0); counters[1] += n_filters - f0; counters[2] += n_outer - o0; }
            dom.n_replans += n_outer - o0;
            MHIP_HIP(hipGetLastError());
            MHIP_HIP(hipStreamSynchronize(stream));
            return;
        }
        struct Direct { bool& f; explicit Direct(bool& b) : f(b) { f = true; } ~Direct() { f = false; } } guard_direct(xf_direct);   // packs send, unpacks wait
        halo_start(dt);
        // a check carried over from the previous call belongs to that call's last step; a caller that continues somewhere else gets a
        // fresh one at its first step instead
        if (xf.plan_pending && xf.plan_step != first_step) { xf.plan_pending = false; xf.next_check = first_step + 1; }
        for (int64_t s = first_step + 1; s <= last; ++s) {
            const bool cm = remove_cm_every != 0 && s % remove_cm_every == 0;
            bool replan = false;
            if (xf.plan_pending && s > xf.plan_step) {          // the numbers of the check issued a step ago have long arrived
                MHIP_HIP(hipEventSynchronize(xf.ev_plan));
                xf.plan_pending = false;
                // an exchange timed out somewhere before this check: the steps since ran on stale ghost rows and the triple may be partial
                // (ranks could decide differently) — give the chunk up here instead of queueing the rest of it
                if (xf.world > 1 && xf.h_red3[3] != 0.f) { xf_check_errors(); throw ApiError{MHIP_ERR_STATE, "ghost exchange timed out"}; }
                int32_t check_in = 0;
                const float red[3] = {xf.h_red3[0], xf.h_red3[1], xf.h_red3[2]};
                // A check issued at a step whose own force pass went on to prune (or search) measured lists that no longer exist — a check asked for k steps
                // after a cadence step is read exactly at the next cadence step, where the next check is issued before the pass that carries out ITS prune:
                // acting on it pruned again one step later (the lj1m domain loop pruned every 20 steps where mhip_vv_run prunes every 25).  Every rank counts
                // the same prunes and searches, so every rank drops the same measurements; the fresh list is looked at at the next cadence step.
                const bool lists_replaced = n_filters != xf.plan_prune_id || n_outer != xf.plan_outer_id;
                const int action = lists_replaced ? 0 : (std::isinf(red[0]) ? 2 : plan_decide_late(xf.plan_step, red, &check_in, (int)(s - xf.plan_step)));
                if (counters) { counters[1] += action == 1; counters[2] += action == 2; }
                xf.next_check = check_in > 0 ? xf.plan_step + check_in : -1;
                replan = action == 2;
            }
            if (!replan && !xf.plan_pending) {      // (a check issued at the last step of this call is read at the first step of the next one)
                if (ghost_margin <= 0 && n_ghost > 0) replan = s % every == 0;      // no margin: ownership and ghosts are redone at every rebuild step
                else if (s % every == 0 || (xf.next_check >= 0 && s >= xf.next_check)) { xf.next_check = -1; xf_issue_plan_check(s); if (counters) counters[0] += 1; }
            }
            // the re-plan inside the engine (replan.h): the step goes on — unpack, re-plan + search, forces (pruning), integrator, pack with the new plan
            const bool replan_here = replan && dev_replan_ok();
            if (replan_here) { replan = false; replan_now = true; }
            const bool stop = replan || s == last;
            if (!stop && !replan_here && halo_fused_ok(s)) {
                // ONE launch: the blocks that need ghosts wait for the peers' rows themselves, every block integrates and sends in its epilogue (kernels.h, HaloStep).
                // The maxima of the next step's validity check ride along when one is due.
                const bool measure = ghost_margin > 0 && ((s + 1) % every == 0 || (xf.next_check >= 0 && s + 1 >= xf.next_check));
                if (debug_on) std::fprintf(stderr, "[mhip %d] %.3f step %lld fused (reads exchange %u) measure %d\n", xf.rank, now_ms(), (long long)s, xf.seq, (int)measure);
                halo_fused(s, dt, cm, measure);
                ++*steps_done;
                continue;
            }
            if (debug_on) std::fprintf(stderr, "[mhip %d] %.3f step %lld separate launches (reads exchange %u) replan %d stop %d inner_valid %d\n", xf.rank, now_ms(), (long long)s, xf.seq, (int)replan_here, (int)stop, (int)inner_valid);
            if (n_ghost > 0 && xf.n_peers > 0 && !replan_here) (void)halo_interior(s);          // the blocks that need no ghost, while the peers' rows arrive
            halo_mid(s, dt, (cm ? 1 : 0) | (stop ? 2 : 0), (cm && stop) ? cm_parts_dev : nullptr, (cm && stop) ? n_parts : 0);   // waits + unpacks … packs + sends
            ++*steps_done;
            // (a check issued at the LAST step of this call stays pending: the first step of the next call reads it, one step late like any
            // other — dropping it here left the inner list unvouched for until the next multiple of `every`, up to 2·every steps after the
            // check before.  A re-plan makes it moot; set_halo_routes clears it then.)
            if (stop) { *reason = replan ? 1 : 0; if (replan) xf.plan_pending = false; break; }
        }
        xf_check_errors();      // (one stream sync per call: a chunk is ≈ 100 steps)
        prune_resolve(true);    // (a pruning pass of the chunk's last steps: its figures, and its verdict on the ghost margin, belong to this call)
    }


    // ---- the re-plan inside the engine (replan.h; SURVEY §8(e) "Migration"): ownership, ghosts and the per-step message tables are redone on the
    // device when the collective check says so, in the middle of mhip_domain_run's step — behind the unpack of the step's ghost rows (which also
    // carry the peers' Σ m v of the step before), in front of its force pass, which then prunes the freshly searched outer list exactly as a
    // single domain's pass does behind an outer search.  The host reads one table (counts, error word) and runs the search.
    struct Dom {
        bool ready = false; int world = 1, me = 0; int64_t n_replans = 0, n_migrated = 0; double plan_ms = 0, search_ms = 0;      // host wall time inside the re-plans: planning (launches + the one sync) / the search behind it
        DBuf<int64_t> gid[2]; int gcur = 0;
        DBuf<RpTab> tab; RpTab* h_tab = nullptr; T* h_lj0 = nullptr;
        DBuf<uint32_t> mask; DBuf<int32_t> blk_cnt, blk_off, err, ranks;
        DBuf<int32_t> send_idx, send_cm_pos, recv_dst; DBuf<T> send_shift;
        int64_t tables_cap = 0;
    } dom;
    ReplanGeom<T> dom_g{};
    const bool dev_replan_env = env_int("MOLLYHIP_DEVICE_REPLAN", 1) != 0;
    bool replan_now = false;
    T uni_s0 = T(0), uni_e0 = T(0);      // σ, ϵ of the one atom type the uniform-LJ constants were made for (set_atoms)

    void dom_release() {
        dom.gid[0].release(); dom.gid[1].release(); dom.tab.release(); dom.mask.release(); dom.blk_cnt.release(); dom.blk_off.release(); dom.err.release(); dom.ranks.release();
        dom.send_idx.release(); dom.send_cm_pos.release(); dom.recv_dst.release(); dom.send_shift.release();
        if (dom.h_tab) (void)hipHostFree(dom.h_tab); if (dom.h_lj0) (void)hipHostFree(dom.h_lj0);
        dom.h_tab = nullptr; dom.h_lj0 = nullptr; dom.ready = false;
    }

    // ≙ BrickGrid of molly.jl_amd/domain.py: the bricks, this rank's neighbour directions sorted by (peer rank, direction vector), the periodic shift of
    // each, the face thresholds in T — every number formed the way the host planner forms it, so that both planners select the same atoms
    void set_domain(const mhip_domain_geometry* gm, const int64_t* gids_dev) override {
        if (!gm) { dom.ready = false; return; }
        if (caller_indexed_topology()) throw ApiError{MHIP_ERR_UNSUPPORTED, "the in-engine re-plan moves atoms between ranks and resets the caller order: contexts with bonded terms, exception lists, special pairs or PME keep the host planner"};
        const int gx = gm->grid[0], gy = gm->grid[1], gz = gm->grid[2];
        if (gx < 1 || gy < 1 || gz < 1 || (int64_t)gx * gy * gz > XFER_MAX_RANKS) throw ApiError{MHIP_ERR_INVALID, "domain geometry: 1 .. 64 bricks"};
        const int world = gx * gy * gz, me = gm->rank;
        if (me < 0 || me >= world) throw ApiError{MHIP_ERR_INVALID, "domain geometry: rank out of range"};
        if (!(gm->r_ghost > 0) && world > 1) throw ApiError{MHIP_ERR_INVALID, "domain geometry: ghost reach must be positive"};
        const int grid[3] = {gx, gy, gz}, coord[3] = {me % gx, (me / gx) % gy, me / (gx * gy)};
        ReplanGeom<T> g{};
        g.world = world; g.me = me; g.cm_rows = sizeof(T) == 4 ? 3 : 2;
        for (int d = 0; d < 3; ++d) {
            if (!(gm->box[d] > 0)) throw ApiError{MHIP_ERR_INVALID, "domain geometry: box sides must be positive"};
            const double brick = gm->box[d] / grid[d], lo = brick * coord[d], hi = lo + brick;
            g.grid[d] = grid[d]; g.box[d] = T(gm->box[d]); g.brick[d] = T(brick); g.cut[d] = grid[d] > 1 ? 1 : 0;
            g.near_lo[d] = T(lo + gm->r_ghost); g.near_hi[d] = T(hi - gm->r_ghost);
            if (grid[d] > 1 && brick < gm->r_ghost) throw ApiError{MHIP_ERR_INVALID, "domain geometry: a brick is narrower than the ghost reach"};
            if ((grid[d] > 1) == (cfg.periodic[d] != 0)) throw ApiError{MHIP_ERR_INVALID, "domain geometry: cut axes must be open in the context, uncut ones periodic"};
        }
        struct Dir { int peer; int v[3]; double sh[3]; };
        std::vector<Dir> dirs;
        for (int a = -1; a <= 1; ++a) for (int b = -1; b <= 1; ++b) for (int c = -1; c <= 1; ++c) {
            const int v[3] = {a, b, c};
            if ((!a && !b && !c) || (a && gx == 1) || (b && gy == 1) || (c && gz == 1)) continue;
            Dir D{}; int pc[3];
            for (int d = 0; d < 3; ++d) {
                int q = coord[d] + v[d]; double sh = 0;
                if (q < 0) { q += grid[d]; sh = +gm->box[d]; } else if (q >= grid[d]) { q -= grid[d]; sh = -gm->box[d]; }
                pc[d] = q; D.v[d] = v[d]; D.sh[d] = sh;
            }
            D.peer = (pc[2] * gy + pc[1]) * gx + pc[0];
            dirs.push_back(D);
        }
        std::sort(dirs.begin(), dirs.end(), [](const Dir& x, const Dir& y) { if (x.peer != y.peer) return x.peer < y.peer; for (int d = 0; d < 3; ++d) if (x.v[d] != y.v[d]) return x.v[d] < y.v[d]; return false; });
        if ((int)dirs.size() > RP_MAX_DIRS) throw ApiError{MHIP_ERR_INVALID, "domain geometry: more than 26 directions"};
        std::vector<char> is_peer(world, 0);
        g.n_dirs = (int)dirs.size();
        for (int k = 0; k < g.n_dirs; ++k) {
            if (dirs[k].peer == me) throw ApiError{MHIP_ERR_UNSUPPORTED, "domain geometry: a brick that neighbours itself (one brick on a cut axis)"};
            is_peer[dirs[k].peer] = 1; g.dir_peer[k] = dirs[k].peer;
            for (int d = 0; d < 3; ++d) { g.dvec[k][d] = (signed char)dirs[k].v[d]; g.dir_shift[k][d] = T(dirs[k].sh[d]); }
        }
        // the fused per-step message (Σ m v on the ghost rows) needs every other rank as a peer: 1, 2, 4, 8 bricks
        for (int r = 0; r < world; ++r) if (r != me && !is_peer[r]) throw ApiError{MHIP_ERR_UNSUPPORTED, "the in-engine re-plan covers decompositions in which every other rank is a neighbour (1, 2, 4 or 8 bricks)"};
        MHIP_HIP(hipStreamSynchronize(stream));
        dom_g = g; dom.world = world; dom.me = me;
        dom.gid[0].reserve(cap); dom.gid[1].reserve(cap); dom.gcur = 0;
        if (gids_dev) MHIP_HIP(hipMemcpyAsync(dom.gid[0].p, gids_dev, (size_t)n_owned * sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
        else { std::vector<int64_t> io((size_t)n_owned); for (int64_t i = 0; i < n_owned; ++i) io[i] = i; MHIP_HIP(hipMemcpy(dom.gid[0].p, io.data(), io.size() * sizeof(int64_t), hipMemcpyHostToDevice)); }
        dom.tab.reserve(1); dom.err.reserve(4); dom.ranks.reserve(XFER_MAX_RANKS);
        MHIP_HIP(hipMemsetAsync(dom.err.p, 0, 4 * sizeof(int32_t), stream));
        if (!dom.h_tab) MHIP_HIP(hipHostMalloc((void**)&dom.h_tab, sizeof(RpTab)));
        if (!dom.h_lj0) MHIP_HIP(hipHostMalloc((void**)&dom.h_lj0, 2 * sizeof(T)));
        std::vector<int32_t> others; for (int r = 0; r < world; ++r) if (r != me) others.push_back(r);
        if (!others.empty()) MHIP_HIP(hipMemcpyAsync(dom.ranks.p, others.data(), others.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        dom.ready = true;
    }
    void domain_info(int64_t* out8) override {
        out8[0] = n_owned; out8[1] = n_ghost; out8[2] = dom.n_replans; out8[3] = dom.n_migrated; out8[4] = (int64_t)std::llround(dom.plan_ms * 1e3); out8[5] = (int64_t)std::llround(dom.search_ms * 1e3); out8[6] = out8[7] = 0;
    }
    void domain_export(int64_t* gid_dev, void* par4_dev) override {
        if (!dom.ready) throw ApiError{MHIP_ERR_STATE, "mhip_set_domain first"};
        hipLaunchKernelGGL(k_rp_export<T>, dim3(cdiv(n_owned, 256)), dim3(256), 0, stream, n_owned, (const int32_t*)inv.p, (const T4*)pos[cur].p, (const T4*)vel[cur].p, (const T2*)lj[cur].p,
                           (const int64_t*)dom.gid[dom.gcur].p, gid_dev, (T*)par4_dev);
        MHIP_HIP(hipGetLastError());
    }
    // caller order is reset by a device re-plan (k_rp_mig_send / k_rp_compact move x, v, q, σ, ϵ, m and the global id): anything indexed by caller order —
    // bonded terms, exception lists, special pairs, the PME charge mesh's exclusions — would be scrambled, so such contexts keep the host planner
    bool caller_indexed_topology() const { return bonded.any() || has_exc || n_special > 0 || pme.on(); }
    bool dev_replan_ok() const {
        if (!dom.ready || !dev_replan_env || !hp_set || caller_indexed_topology()) return false;
        if (dom.world == 1) return n_ghost == 0;
        if (!xf.region || !xf.routes || xf.n_peers != dom.world - 1 || xf.world != dom.world || hp.cm_rows != dom_g.cm_rows) return false;
        for (int r = 0; r < dom.world; ++r) if (!xf.peers.region[r] || !xf.plan.area[r]) return false;
        return true;
    }
    // the one-type LJ decision of set_atoms, from σ and ϵ of atom 0 and the "some atom differs" word of the device check
    void apply_uniform(bool all_equal, T s0, T e0) {
        ljm = ljm_base;
        if (ljm_base == LJ_DIST && !tri_mode && all_equal && s0 != T(0) && e0 != T(0)) {
            T sm = (s0 + s0) / T(2), em = std::sqrt(e0 * e0);   // the mixing rules applied to equal values
            I.lj_s2 = sm * sm; I.lj_24e = T(24) * em; I.lj_4e = T(4) * em;
            const double s6 = std::pow((double)sm, 6), c6 = 24.0 * (double)em * s6, c12 = 48.0 * (double)em * s6 * s6;
            const bool normal = c12 > 1e-30 && c12 < 1e30 && c6 > 1e-30 && c6 < 1e30;   // else: the generic loop, which works on σ²/r²
            I.lj_c6 = normal ? T(c6) : T(0); I.lj_c12 = normal ? T(c12) : T(0);
            ljm = LJ_DIST_UNIFORM;
        }
        uni_s0 = s0; uni_e0 = e0;
    }
    void device_replan(int64_t step_n) {
        ++dom.n_replans;
        const auto t_begin = std::chrono::steady_clock::now();
        auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        if (dom.world == 1) { rebuild(step_n); dom.search_ms += ms_since(t_begin); return; }      // one brick: nobody to hand atoms to, no ghosts — an outer-list rebuild, as in mhip_vv_run
        const ReplanGeom<T>& g = dom_g;
        const int world = g.world, me = g.me, cr = g.cm_rows;
        const int o = cur, n = 1 - cur, go = dom.gcur, gn = 1 - dom.gcur;
        const int nb = cdiv(cap, 256);
        if (dom.tables_cap < xf.rows_cap) {
            dom.send_idx.reserve(xf.rows_cap); dom.send_shift.reserve(3 * (size_t)xf.rows_cap); dom.recv_dst.reserve(xf.rows_cap); dom.send_cm_pos.reserve((size_t)XFER_MAX_RANKS * 4);
            xf.row_peer.reserve(xf.rows_cap); xf.row_dst.reserve(xf.rows_cap);
            dom.tables_cap = xf.rows_cap;
        }
        dom.mask.reserve(cap); dom.blk_cnt.reserve((size_t)RP_MAX_DIRS * nb); dom.blk_off.reserve((size_t)RP_MAX_DIRS * nb);
        const uint32_t seq = ++xf.rp_seq;
        const XferHeader* mine = reinterpret_cast<const XferHeader*>(xf.region);
        const unsigned long long ticks = xf_ticks();
        const int n_old = (int)n_owned;
        MHIP_HIP(hipMemsetAsync(dom.tab.p, 0, sizeof(RpTab), stream));
        // A. migration
        tr("k_rp_owner_keys");
        hipLaunchKernelGGL(k_rp_owner_keys<T>, dim3(cdiv(n_owned, 256)), dim3(256), 0, stream, n_owned, (const T4*)pos[o].p, (const int32_t*)inv.p, g, key_in.p, idx_in.p, dom.tab.p);
        size_t tb = cub_tmp.n;
        MHIP_HIP(sort_pairs_u32(cub_tmp.p, tb, key_in.p, key_out.p, idx_in.p, perm.p, n_old, ilog2(world + 1) + 1, stream));
        tr("k_rp_exchange<0>");
        hipLaunchKernelGGL(k_rp_exchange<0>, dim3(1), dim3(64), 0, stream, xf.peers, mine, world, me, seq, dom.tab.p, cr, n_old, (int)cap, (int)xf.rows_cap, dom.err.p, ticks);
        tr("k_rp_mig_send");
        hipLaunchKernelGGL(k_rp_mig_send<T>, dim3(64), dim3(256), 0, stream, (const RpTab*)dom.tab.p, (const uint32_t*)key_out.p, (const int32_t*)perm.p, (const int32_t*)inv.p, (const T4*)pos[o].p,
                           (const T4*)vel[o].p, (const T2*)lj[o].p, (const int64_t*)dom.gid[go].p, g, xf.plan, xf.peers, (const int32_t*)dom.ranks.p, world - 1, seq, xf.done.p);
        hipLaunchKernelGGL(k_rp_wait_rows<0>, dim3(1), dim3(64), 0, stream, mine, world, me, seq, dom.tab.p, dom.err.p, ticks);
        tr("k_rp_compact");
        hipLaunchKernelGGL(k_rp_compact<T>, dim3(std::min(cdiv(cap, 256), 1024)), dim3(256), 0, stream, (const RpTab*)dom.tab.p, (const int32_t*)perm.p, (const int32_t*)inv.p, (const T4*)pos[o].p,
                           (const T4*)vel[o].p, (const T2*)lj[o].p, (const int64_t*)dom.gid[go].p, g, (const unsigned char*)xf.plan.area[me], pos[n].p, vel[n].p, lj[n].p, dom.gid[gn].p);
        // B. ghost plan
        tr("k_rp_ghost_count");
        hipLaunchKernelGGL(k_rp_ghost_count<T>, dim3(nb), dim3(256), 0, stream, (const RpTab*)dom.tab.p, (const T4*)pos[n].p, g, dom.mask.p, dom.blk_cnt.p);
        hipLaunchKernelGGL(k_rp_ghost_scan<T>, dim3(1), dim3(1024), 0, stream, dom.tab.p, g, nb, (const int32_t*)dom.blk_cnt.p, dom.blk_off.p);
        tr("k_rp_exchange<1>");
        hipLaunchKernelGGL(k_rp_exchange<1>, dim3(1), dim3(64), 0, stream, xf.peers, mine, world, me, seq, dom.tab.p, cr, n_old, (int)cap, (int)xf.rows_cap, dom.err.p, ticks);
        tr("k_rp_ghost_send");
        hipLaunchKernelGGL(k_rp_ghost_send<T>, dim3(nb + 1), dim3(256), 0, stream, (const RpTab*)dom.tab.p, (const T4*)pos[n].p, (const T4*)vel[n].p, (const T2*)lj[n].p, (const uint32_t*)dom.mask.p,
                           (const int32_t*)dom.blk_off.p, nb, g, dom.send_idx.p, dom.send_shift.p, dom.send_cm_pos.p, xf.row_peer.p, xf.row_dst.p, xf.plan, xf.peers, (const int32_t*)dom.ranks.p, world - 1, seq, xf.done.p);
        hipLaunchKernelGGL(k_rp_wait_rows<1>, dim3(1), dim3(64), 0, stream, mine, world, me, seq, dom.tab.p, dom.err.p, ticks);
        tr("k_rp_ghost_recv");
        hipLaunchKernelGGL(k_rp_ghost_recv<T>, dim3(std::min(cdiv(cap, 256), 1024)), dim3(256), 0, stream, dom.tab.p, g, (const unsigned char*)xf.plan.area[me], pos[n].p, vel[n].p, lj[n].p, dom.recv_dst.p);
        MHIP_HIP(hipGetLastError());
        MHIP_HIP(hipMemcpyAsync(dom.h_tab, dom.tab.p, sizeof(RpTab), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipMemcpyAsync(dom.h_lj0, lj[n].p, sizeof(T2), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        const RpTab& t = *dom.h_tab;
        if (t.err) {
            int32_t e4[4] = {0, 0, 0, 0};
            MHIP_HIP(hipMemcpy(e4, dom.err.p, sizeof(e4), hipMemcpyDeviceToHost));
            MHIP_HIP(hipMemset(dom.err.p, 0, 4 * sizeof(int32_t)));
            std::string tabs = " [rank " + std::to_string(me) + " at step " + std::to_string(step_n) + ": " + std::to_string(n_old) + " owned before, " + std::to_string(t.n_stay) + " stay, " + std::to_string(t.n_leave) + " leave, " +
                               std::to_string(t.n_arrive) + " arrive, " + std::to_string(t.n_send) + " ghost rows out, " + std::to_string(t.n_ghost) + " ghosts in, capacity " + std::to_string(cap) + " atoms / " + std::to_string(xf.rows_cap) + " rows; to / from each rank:";
            for (int r = 0; r < world; ++r) tabs += " " + std::to_string(t.leave_cnt[r]) + "/" + std::to_string(t.arr_from[r]) + "|" + std::to_string(t.send_to[r]) + "/" + std::to_string(t.gh_from[r]);
            tabs += "]";
            if (t.err & RP_ERR_TIMEOUT) throw ApiError{MHIP_ERR_STATE, "re-plan: a peer's counts or rows did not arrive in time (MOLLYHIP_XFER_TIMEOUT_MS)" + wait_report(e4) + tabs};
            throw ApiError{MHIP_ERR_CAPACITY, std::string("re-plan: ") + ((t.err & RP_ERR_ATOMS) ? "a sub-domain's atoms + ghosts exceed its context capacity" : (t.err & RP_ERR_PLAN_AREA) ? "the migrating atoms exceed a plan area"
                                             : "the ghost rows exceed a receive region") + " (create the contexts with more room)" + tabs};
        }
        if (debug_on) std::fprintf(stderr, "[mhip %d] re-plan at step %lld: %d stay, %d leave, %d arrive; %d ghost rows out, %d ghosts in\n", me, (long long)step_n, t.n_stay, t.n_leave, t.n_arrive, t.n_send, t.n_ghost);
        // commit: the new local set in its identity order
        cur = n; dom.gcur = gn; dom.n_migrated += t.n_arrive;
        n_owned = t.n_owned; n_ghost = t.n_ghost; n_tot = n_owned + n_ghost;
        hipLaunchKernelGGL(k_iota2, dim3(std::min(cdiv(n_tot, 256), 1024)), dim3(256), 0, stream, n_tot, orig[cur].p, inv.p);
        MHIP_HIP(hipMemsetAsync(frc[cur].p, 0, n_tot * sizeof(T4), stream));
        stale = true; frc_valid = false; trk_issued = false; interior_done = false; frc_run_total = false; coords_moved = false; state_pending = false;
        const int size_class = n_owned >= 100000 ? 2 : (n_owned >= 40000 ? 1 : 0);
        const long long key = (n_ghost > 0 ? 1 : 0) | (dual_disabled ? 2 : 0) | (size_class << 2) | (margin_halvings << 4) | (margin_zero ? 128 : 0) | ((long long)std::llround(ghost_margin * 1e6) << 8);
        if (key != grid_key) { setup_grid(); choose_blocking(); grid_key = key; }
        const bool all_equal = t.uni_bad == 0;
        if ((ljm == LJ_DIST_UNIFORM) != (ljm_base == LJ_DIST && all_equal && dom.h_lj0[0] != T(0) && dom.h_lj0[1] != T(0)) || dom.h_lj0[0] != uni_s0 || dom.h_lj0[1] != uni_e0) apply_uniform(all_equal, dom.h_lj0[0], dom.h_lj0[1]);
        // the per-step message of the new plan (mhip_halo_plan, mhip_halo_routes): tables made by the kernels above
        hp.first_ghost = n_owned; hp.n_recv_rows = t.n_ghost + (world - 1) * cr; hp.recv = nullptr; hp.recv_dst = dom.recv_dst.p;
        hp.n_cm_peers = world - 1; hp.cm_rows = cr; hp.send_idx = dom.send_idx.p; hp.send_shift = dom.send_shift.p;
        hp.n_send_rows = t.n_send + (world - 1) * cr; hp.send = nullptr; hp.send_cm_pos = dom.send_cm_pos.p; hp.n_send_cm = (world - 1) * cr;
        xf.plan_pending = false; xf.next_check = -1; hx.plan_ok = false; hx.trk_step = -1;
        dom.plan_ms += ms_since(t_begin);
        const auto t_search = std::chrono::steady_clock::now();
        rebuild(step_n);
        dom.search_ms += ms_since(t_search);
    }

    // one MD step of a ghosted sub-domain in two calls around the ghost exchange
    void halo_begin(double dt, const int32_t* idx_dev, const void* shift_dev, int64_t n, void* out_dev) override {
        vv_stage1(dt);
        gather_coords(idx_dev, shift_dev, n, out_dev);
    }
    // Between halo_begin and halo_end, while the ghost coordinates travel: the pair forces of the blocks whose tile holds no ghost.
    // Only on steps whose force pass is a plain one over lists that are known to be good (no search, no prune, no cadence decision
    // pending); returns 1 if it launched, 0 if halo_end will do the whole pass.
    int halo_interior(int64_t step_n) override {
        interior_done = false;
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        if (n_ghost == 0 || stale || !frc_valid_for_split() || (step_n % every == 0 && step_n != last_build_step)) return 0;
        launch_pair_kernel(false, 1);
        interior_done = true;
        return 1;
    }
    bool frc_valid_for_split() const { return dual ? inner_valid : !lazy_single; }
    void halo_end(int64_t step_n, double dt, int64_t first, int64_t n, const void* in_dev, double* cm_out4_dev) override {
        scatter_coords(first, n, in_dev);
        stage2_cadenced(step_n, dt, cm_out4_dev != nullptr);
        if (cm_out4_dev) {   // this rank's {ΣPx, ΣPy, ΣPz, ΣM} for the all-reduce; nothing pending locally: the TOTAL comes back via remove_cm_dev
            hipLaunchKernelGGL(k_cm_finalize<T>, dim3(1), dim3(256), 0, stream, n_cm_step, (const double*)cm_step.p, cm_out4_dev, (T*)nullptr);
            cm_pending = 0;
        }
        MHIP_HIP(hipGetLastError());
    }
    void rebuild_now(int64_t step_n) override { flush_cm(); resolve_track(step_n); lists_after_set_state(); if (stale) rebuild(step_n); else refresh(step_n); }

    void vv_run(int64_t first_step, int64_t n_steps, double dt, int remove_cm_every) override {
        if (!state_set || !params_set) throw ApiError{MHIP_ERR_STATE, "set_atoms and set_state must be called before vv_run"};
        if (n_ghost > 0) throw ApiError{MHIP_ERR_STATE, "vv_run is single-domain; drive ghosted domains with vv_stage1/vv_stage2"};
        cur_dt = dt;
        InRun guard_in_run(in_run);
        if (first_step == 0 && remove_cm_every != 0) remove_cm();                 // simulators.jl:563
        vv_init(first_step);                                                      // :564-571
        vv_loop(first_step, n_steps, dt, remove_cm_every, nullptr, 0);
        flush_cm();
        MHIP_HIP(hipGetLastError());
        MHIP_HIP(hipStreamSynchronize(stream));
    }
    // the step loop of a single domain: forces of first_step are in place.  cm_parts_last (nullable): where the LAST step leaves its Σ m v partials (n_parts_last
    // blocks) instead of registering their removal with the context — mhip_domain_run on one brick, whose caller sums them over the (one) rank.
    void vv_loop(int64_t first_step, int64_t n_steps, double dt, int remove_cm_every, double* cm_parts_last, int n_parts_last) {
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        // fused stepping: first kick + drift once, then ONE integrator launch between consecutive force passes (k_vv_mid), the
        // plain second kick at the end.  A thermostat needs v_n between the kicks: the two-launch form then.
        const bool fused = !(andersen_prob > 0);
        InRun guard_fused(in_vv_fused); in_vv_fused = fused;
        const bool pre = dual;                                                    // without the dual list: the reference's order
        const int64_t last = first_step + n_steps;
        int half = 0;
        if (fused && n_steps > 0) vv_stage1(dt);
        for (int64_t step = first_step + 1; step <= last; ++step) {
            if (!fused) vv_stage1(dt);                                            // :594-609
            // find_neighbors at step % n_steps == 0 (:645, neighbors.jl:396) builds the list from the coordinates of THIS step; it is
            // scheduled before the force pass so that, with the dual pair list, that pass can prune the outer list on the way.
            // Forces are unaffected: the pass walks a superset of the old list and every interaction has a cutoff <= r_list.
            if (trk_issued && step > trk_step) resolve_track(step);
            if (pre && check_due(step, every)) refresh(step);
            const bool cm = remove_cm_every != 0 && step % remove_cm_every == 0;
            if (!fused) {
                stage2_impl(step, dt, cm);                                        // :612-628
                apply_coupling(step);                                             // :630
                if (!pre && check_due(step, every)) refresh(step);
                continue;
            }
            // the validity check of step + 1 is measured where its coordinates are made: by this step's integrator launch — or by the pair pass itself when it integrates
            const bool measure = step != last && async_ok() && !trk_issued && check_due(step + 1, every);
            step_req.on = step != last && !bonded.any() && !pme.on(); step_req.cm = cm;      // (one kernel either way: its stage time is its own, so the stage timers leave it fused)
            step_req.gcv = step != last && bonded.any() && pme.on() && (pre || !check_due(step, every));      // (a re-sort behind the pass would want the total force array)
            step_req.measure = measure; step_req.dt = dt;
            step_done = false;
            step_forces(step);
            step_req.on = step_req.gcv = false;
            if (step_done) {      // the pair pass integrated on the way (k_forces STEP): no integrator launch for this step
                step_done = false;
                if (measure) {
                    if (!h_trk) MHIP_HIP(hipHostMalloc((void**)&h_trk, 4 * sizeof(float)));
                    if (!ev_trk) MHIP_HIP(hipEventCreateWithFlags(&ev_trk, hipEventDisableTiming));
                    hipLaunchKernelGGL(k_track_reduce, dim3(1), dim3(256), 0, stream, step_parts, (const float*)trk_part.p, trk_out.p, h_trk);
                    MHIP_HIP(hipEventRecord(ev_trk, stream));
                    trk_issued = true; trk_step = step + 1; trk_prev_vmax = last_vmax; trk_prune_id = n_filters; trk_outer_id = n_outer;
                }
                pend_a = nullptr; cm_pending = 0; cm_ext = nullptr;
                if (cm) { cm_pending = 2; cm_ext = cm_blk.p + (size_t)step_half * 4 * step_parts; n_cm_step = step_parts; step_half ^= 1; }
                frc_valid = false;
                continue;
            }
            if (!pre && check_due(step, every)) { fold_side_forces(); refresh(step); }   // the sort permutes vel / frc with the atoms; Σ m v does not care
            // every block re-sums the previous step's per-block Σ m v partials (32 bytes each), so fewer, longer blocks pay: 1024 blocks
            // re-read 32 MB from L2 per launch — more than the 21 MB of atoms of the 256k-atom fluid (13.0 → 10.0 µs with 256 blocks;
            // 1M atoms: 21.2 → 20.2 µs with 512, 21.8 with 256)
            const bool parts_out = step == last && cm && cm_parts_last != nullptr;
            const int nb = parts_out ? n_parts_last : std::min(cdiv(n_owned, 256), (int)std::max<int64_t>(256, std::min<int64_t>(512, n_owned / 2048)));
            const double* cm_in = cm_pending == 2 ? cm_src() : (const double*)nullptr;
            double* cm_out = cm ? (parts_out ? cm_parts_last : cm_step.p + (size_t)half * 4 * 1024) : (double*)nullptr;
            prof.begin(2, stream);
            // the speeds for a check that the next step's force pass will measure (see resolve_track); evaluated behind the pass: a prune inside it makes the lists checkable again
            const bool measure_mid = step != last && async_ok() && !trk_issued && check_due(step + 1, every);
            if (measure_mid) { trk_part.reserve(3 * (size_t)std::max(n_blocks, 1024)); trk_out.reserve(4); }
            auto go = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, stream, n_owned, pos[cur].p, vel[cur].p, (const T4*)frc[cur].p, T(dt), T(dt) / T(2),
                                   cm_in, n_cm_step, cm_out, (const T4*)pend_a, G,
                                   measure_mid ? (const T4*)pos_snap_in.p : (const T4*)nullptr, measure_mid ? (const T4*)pos_snap.p : (const T4*)nullptr, measure_mid ? trk_part.p : (float*)nullptr);
            };
            if (step == last) { if (cm) go(k_vv_mid<T, true, true>); else go(k_vv_mid<T, false, true>); }
            else { if (cm) go(k_vv_mid<T, true, false>); else go(k_vv_mid<T, false, false>); }
            prof.end(2, stream);
            if (measure_mid) {   // the check of step + 1: reduce, copy, event — read by resolve_track at step + 2
                if (!h_trk) MHIP_HIP(hipHostMalloc((void**)&h_trk, 4 * sizeof(float)));
                if (!ev_trk) MHIP_HIP(hipEventCreateWithFlags(&ev_trk, hipEventDisableTiming));
                hipLaunchKernelGGL(k_track_reduce, dim3(1), dim3(256), 0, stream, nb, (const float*)trk_part.p, trk_out.p, h_trk);   // (straight into pinned host memory)
                MHIP_HIP(hipEventRecord(ev_trk, stream));
                trk_issued = true; trk_step = step + 1; trk_prev_vmax = last_vmax; trk_prune_id = n_filters; trk_outer_id = n_outer;
            }
            if (step == last) frc_run_total = pend_a == nullptr && n_ghost == 0;   // (side arrays are added by the kick, not folded)
            pend_a = nullptr;
            cm_pending = 0; cm_ext = nullptr;
            if (cm && !parts_out) { cm_pending = 2; cm_ext = cm_out; n_cm_step = nb; half ^= 1; }
            if (step != last) frc_valid = false;                                  // frc[cur] belongs to the coordinates before the drift
        }
    }

    // ---- stochastic dynamics (kernels in stochastic.hip) ------------------------------------------------------
    StochP<T> stoch_params(double kT, uint64_t key, uint64_t ctr1) const {
        StochP<T> P{};
        P.noise_kt = std::sqrt(kT); P.key = key; P.ctr1 = ctr1; P.natoms = (uint64_t)cfg.n_atoms;
        return P;
    }
    // mode 0: one application of the Andersen thermostat with per-atom probability `prob` (coupling.jl:196-211);
    // mode 1: random_velocities! (spatial.jl:803-831).  Velocities only: positions, forces and the pair list stay valid.
    void redraw_velocities(int mode, double kT, double prob, uint64_t key, uint64_t ctr1) override {
        if (!state_set || !params_set) throw ApiError{MHIP_ERR_STATE, "set_atoms and set_state must be called before drawing velocities"};
        if (n_ghost > 0) throw ApiError{MHIP_ERR_UNSUPPORTED, "velocity draws are single-domain"};
        if (!(kT >= 0)) throw ApiError{MHIP_ERR_INVALID, "kT must be non-negative"};
        StochP<T> P = stoch_params(kT, key, ctr1);
        const double pc = std::min(std::max(prob, 0.0), std::nextafter(1.0, 0.0));          // clamp(…, 0, prevfloat(1.0))
        P.prob_u64 = (uint64_t)std::nearbyint(std::ldexp(pc, 64));                            // round(UInt64, prob·2⁶⁴) ≤ 2⁶⁴ − 2¹¹
        launch_redraw<T>(stream, mode, n_owned, vel[cur].p, orig[cur].p, P, cm_pending == 1 ? (const T*)vcm.p : (const T*)nullptr,
                         cm_pending == 2 ? cm_src() : (const double*)nullptr, n_cm_step);
        cm_pending = 0; cm_ext = nullptr;
        MHIP_HIP(hipGetLastError());
    }
    // AndersenThermostat as the coupling of vv_run / langevin_run: applied after every step's CM removal (simulators.jl:630, 1209);
    // the reference draws (ctr1, key) from the host rng per step — here they are the words of philox(step, 0; seed)
    double andersen_kT = 0, andersen_prob = 0; uint64_t andersen_seed = 0;
    void set_andersen(double kT, double prob, uint64_t seed) override { andersen_kT = kT; andersen_prob = prob; andersen_seed = seed; }
    void apply_coupling(int64_t step) {
        if (!(andersen_prob > 0)) return;
        uint32_t w[4]; philox_host((uint64_t)step, 0, andersen_seed, w);
        redraw_velocities(0, andersen_kT, andersen_prob, ((uint64_t)w[3] << 32) | w[2], ((uint64_t)w[1] << 32) | w[0]);
    }

    // simulate!(sys, ::Langevin, n_steps) (simulators.jl:1099-1220) without constraints: per step the forces of the current
    // coordinates, ONE fused update launch (kick, half drift, O-step, half drift, wrap, Σ m v partials), the neighbour cadence.
    // ctr1 advances by one per step (:1190) from ctr1_0 + (first_step's offset), so chunked continuation reproduces one long run.
    void langevin_run(int64_t first_step, int64_t n_steps, double dt, double kT, double friction, int remove_cm_every, uint64_t key, uint64_t ctr1_0) override {
        if (!state_set || !params_set) throw ApiError{MHIP_ERR_STATE, "set_atoms and set_state must be called before langevin_run"};
        if (n_ghost > 0) throw ApiError{MHIP_ERR_STATE, "langevin_run is single-domain"};
        if (!(kT >= 0) || !(friction >= 0)) throw ApiError{MHIP_ERR_INVALID, "temperature and friction must be non-negative"};
        const int every = cfg.rebuild_every > 0 ? cfg.rebuild_every : 10;
        cur_dt = dt;
        InRun guard_in_run(in_run);
        InRun guard_lang(in_lang_fused); in_lang_fused = bonded.any() && pme.on() && fuse_gcv_env;
        InRun guard_lang_async(in_lang_async); in_lang_async = (in_lang_fused || (!bonded.any() && !pme.on() && fuse_step_env)) && !(andersen_prob > 0);
        if (first_step == 0 && remove_cm_every != 0) remove_cm();                 // :1115
        start_lists(first_step);                                                  // :1116
        const double vs = std::exp(-dt * friction);                               // :1091-1092
        StochP<T> P = stoch_params(kT, key, ctr1_0);
        P.dt = T(dt); P.dt_half = T(dt) / T(2); P.vel_scale = T(vs); P.noise_kt = std::sqrt(1.0 - vs * vs) * std::sqrt(kT);
        const int nb = std::min(cdiv(n_owned, 256), 1024);
        int half = 0;
        for (int64_t step = first_step + 1; step <= first_step + n_steps; ++step) {
            // a check measured by the update launch of step s (the coordinates x_s it made) is read at the top of step s + 2, behind a whole step of queued work — the
            // scheme of mhip_vv_run, whose loop is one force pass ahead of this one (there the launch of step s makes x_(s+1)); a check left by a run before: at once
            if (trk_issued && (!in_lang_async || step > trk_step + 1)) resolve_track(step);
            const bool cm = remove_cm_every != 0 && step % remove_cm_every == 0;
            P.ctr1 = ctr1_0 + (uint64_t)(step - first_step - 1);
            // a small system's step (bonded terms + PME): its last force launch — interpolation + bonded sums — runs the update as well (step_fused.h, k_gather_collect_vv<…, LANG>),
            // every step of the run: a Langevin step is complete in itself, there is no closing half kick to keep a launch for
            const bool measure = in_lang_async && async_ok() && !trk_issued && check_due(step, every);      // the check refresh(step) below would make with a drained stream
            step_req.gcv = bonded.any() && pme.on(); step_req.on = !bonded.any() && !pme.on(); step_req.lang = &P; step_req.cm = cm; step_req.measure = measure; step_req.dt = dt;      // (on: the packed fp32 one-type pass runs the update in its epilogue, k_forces<…, STEP, ·, LANG>)
            step_done = false;
            step_forces(step);                                                    // :1173
            step_req.gcv = step_req.on = false; step_req.lang = nullptr; step_req.measure = false;
            if (step_done) {
                step_done = false;
                if (measure) {
                    if (!h_trk) MHIP_HIP(hipHostMalloc((void**)&h_trk, 4 * sizeof(float)));
                    if (!ev_trk) MHIP_HIP(hipEventCreateWithFlags(&ev_trk, hipEventDisableTiming));
                    hipLaunchKernelGGL(k_track_reduce, dim3(1), dim3(256), 0, stream, step_parts, (const float*)trk_part.p, trk_out.p, h_trk);
                    MHIP_HIP(hipEventRecord(ev_trk, stream));
                    trk_issued = true; trk_step = step; trk_prev_vmax = last_vmax; trk_prune_id = n_filters; trk_outer_id = n_outer;
                }
                pend_a = nullptr; cm_pending = 0; cm_ext = nullptr; frc_valid = false;
                if (cm) { cm_pending = 2; cm_ext = cm_blk.p + (size_t)step_half * 4 * step_parts; n_cm_step = step_parts; step_half ^= 1; }
                apply_coupling(step);
                if (check_due(step, every)) refresh(step);
                continue;
            }
            prof.begin(2, stream);
            double* cm_out = cm ? cm_step.p + (size_t)half * 4 * 1024 : (double*)nullptr;   // the other half may still be read by this launch
            launch_langevin<T>(stream, nb, n_owned, pos[cur].p, vel[cur].p, (const T4*)frc[cur].p, orig[cur].p, P,
                               cm_pending == 1 ? (const T*)vcm.p : (const T*)nullptr, cm_pending == 2 ? cm_src() : (const double*)nullptr, n_cm_step,
                               cm_out, G, (const T4*)pend_a);      // (the side array of a small system's step is added by the update itself, as k_vv_mid does: no k_add_forces launch)
            prof.end(2, stream);
            pend_a = nullptr;
            cm_pending = 0; cm_ext = nullptr; frc_valid = false;
            if (cm) { cm_pending = 2; cm_ext = cm_out; n_cm_step = nb; half ^= 1; }   // :1204-1206, subtracted by the next consumer
            apply_coupling(step);                                                 // :1208
            if (check_due(step, every)) refresh(step);                            // :1211 — the next force pass prunes the fresh outer list
        }
        flush_cm();
        MHIP_HIP(hipGetLastError());
        MHIP_HIP(hipStreamSynchronize(stream));
    }

    int64_t export_neighbors(int32_t* oi, int32_t* oj, uint8_t* osp, int64_t capacity) override { return export_list(oi, oj, osp, capacity, true); }
    // now = false (statistics): count the pairs of the list in use, never search for it
    int64_t export_list(int32_t* oi, int32_t* oj, uint8_t* osp, int64_t capacity, bool now) {
        if (now) lists_after_set_state();
        if (stale) {   // never built, or invalidated by new coordinates / exceptions: search now
            if (!now) return 0;
            if (!state_set || !params_set) throw ApiError{MHIP_ERR_STATE, "no neighbour list: call set_atoms and set_state first"};
            flush_cm(); rebuild(last_build_step == std::numeric_limits<int64_t>::min() ? 0 : last_build_step);
        }
        if (now && lazy_single && (last_build_step != last_prune_step || export_needs_search)) { flush_cm(); rebuild(last_build_step); }   // skipped rebuilds / moved coordinates: hand out the list of NOW
        const int32_t *x_tidx = tile_idx.p, *x_tcnt = tile_cnt.p, *x_rows = wave_rows.p; const uint2* x_nbr = nbr.p;
        if (dual) {
            // the reference's list at the current coordinates = the outer list filtered with the exact predicate; valid as long as
            // nobody moved more than half the margin since the outer search, else search again first
            for (int attempt = 0; attempt < 2; ++attempt) {
                launch_filter();
                MHIP_HIP(hipMemcpyAsync(h_flags, flags.p, N_FLAGS * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                MHIP_HIP(hipStreamSynchronize(stream));
                float d2; std::memcpy(&d2, &h_flags[FLAG_MAX_DISP2], sizeof(float));
                if (2.0 * std::sqrt((double)d2) <= outer_margin * 0.98 || !now) break;
                flush_cm(); rebuild(last_build_step);
                if (!dual) break;
            }
            if (dual) { x_tidx = tile_idx_x.p; x_tcnt = tile_cnt_x.p; x_rows = rows_x.p; x_nbr = nbr_x.p; }
            else { x_tidx = tile_idx.p; x_tcnt = tile_cnt.p; x_rows = wave_rows.p; x_nbr = nbr.p; }
        }
        DBuf<unsigned long long>& counter = nl_counter; counter.reserve(1);
        MHIP_HIP(hipMemsetAsync(counter.p, 0, sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(k_export_nl<T>, dim3(n_blocks), dim3(BI), 0, stream, n_blocks, BI, JS, T_cap, R_cap, n_owned, (const int32_t*)orig[cur].p, x_tidx, x_tcnt, x_nbr, x_rows,
                           (int32_t*)nullptr, (int32_t*)nullptr, (uint8_t*)nullptr, counter.p, 0ull, eshift);
        unsigned long long n = 0;
        MHIP_HIP(hipMemcpyAsync(&n, counter.p, sizeof(n), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
        if (oi && (int64_t)n <= capacity && n > 0) {
            DBuf<int32_t> di, dj; DBuf<uint8_t> ds; di.reserve(n); dj.reserve(n); ds.reserve(n);
            MHIP_HIP(hipMemsetAsync(counter.p, 0, sizeof(unsigned long long), stream));
            hipLaunchKernelGGL(k_export_nl<T>, dim3(n_blocks), dim3(BI), 0, stream, n_blocks, BI, JS, T_cap, R_cap, n_owned, (const int32_t*)orig[cur].p, x_tidx, x_tcnt, x_nbr, x_rows,
                               di.p, dj.p, ds.p, counter.p, n, eshift);
            MHIP_HIP(hipMemcpyAsync(oi, di.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            MHIP_HIP(hipMemcpyAsync(oj, dj.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            MHIP_HIP(hipMemcpyAsync(osp, ds.p, n * sizeof(uint8_t), hipMemcpyDeviceToHost, stream));
            MHIP_HIP(hipStreamSynchronize(stream));
            di.release(); dj.release(); ds.release();
        }
        return (int64_t)n;
    }

    void export_order(int32_t* out, int64_t capacity) override {
        if (capacity < n_tot) throw ApiError{MHIP_ERR_CAPACITY, "export_order: buffer too small"};
        MHIP_HIP(hipMemcpyAsync(out, orig[cur].p, n_tot * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        MHIP_HIP(hipStreamSynchronize(stream));
    }

    void get_stats(mhip_stats* s) override {
        std::memset(s, 0, sizeof(*s));
        prune_resolve(true);
        s->n_atoms = n_tot; s->n_owned = n_owned; s->n_ghost = n_ghost; s->n_rebuilds = n_rebuilds; s->n_force_calls = n_force_calls;
        s->n_blocks = n_blocks; s->block_atoms = BI; s->j_split = JS; s->minimg_mode = minimg ? 1 : 0; s->max_tile_atoms = max_tile;
        s->last_rebuild_ms = last_rebuild_ms; s->lds_bytes = (int64_t)lds_force;
        s->tile_segments = segmented ? cdiv(std::max(last_pass_tile, 1), std::max(tile_lds, 1)) : 1;
        s->n_group_split_passes = n_gs_passes; s->group_split = gs_groups(); s->n_adopted_outer_lists = (int32_t)std::min<int64_t>(n_adopted, INT32_MAX);
        s->n_fused_steps = n_fused_steps; s->n_box_changes = n_box_changes;
        s->n_list_slots = total_rows * 4 * WAVE;
        if (!stale) {
            std::vector<int32_t> tc(n_blocks);
            MHIP_HIP(hipMemcpy(tc.data(), tile_cnt.p, n_blocks * sizeof(int32_t), hipMemcpyDeviceToHost));
            int64_t t = 0; for (int v : tc) t += v; s->tile_atoms_total = s->outer_tile_atoms_total = t;
            if (dual && inner_valid && !inner_is_outer && tile_cnt_in.p) {      // the plain passes stage the pruned list's compacted tile
                MHIP_HIP(hipMemcpy(tc.data(), tile_cnt_in.p, n_blocks * sizeof(int32_t), hipMemcpyDeviceToHost));
                t = 0; for (int v : tc) t += v; s->tile_atoms_total = t;
            }
            s->n_pairs_full = 2 * export_list(nullptr, nullptr, nullptr, 0, false);
        }
        const int64_t w = sizeof(T), Rp = (coulm != MHIP_COUL_NONE ? 6 : 4) * w;
        s->algorithmic_bytes_step = n_owned * (Rp + 22 * w) + 4 * (s->n_pairs_full / 2);   // N(R_p + 22w) + 4L
        s->force_pass_bytes = n_owned * (Rp + 3 * w) + 4 * (s->n_pairs_full / 2);          // force pass: N(R_p + 3w) + 4L
        // the list's upkeep, priced by the bytes of the lists themselves (2 per slot, padding included — the format's, not the pair count's) and every per-atom array once
        s->n_outer_slots = outer_rows * 4 * WAVE;
        s->build_pass_bytes = n_tot * 4 * w + 2 * s->n_outer_slots + 4 * s->outer_tile_atoms_total;
        s->prune_pass_bytes = n_owned * (Rp + 3 * w) + 2 * s->n_outer_slots + 4 * s->outer_tile_atoms_total + 2 * s->n_list_slots + 4 * s->tile_atoms_total + n_tot * 4 * w;
        prof.resolve(stream);
        for (int k = 0; k < Prof::NS; ++k) { s->prof_ms[k] = prof.ms[k]; s->prof_calls[k] = prof.calls[k]; }
        s->n_outer_builds = n_outer; s->n_filter_passes = n_filters;
    }

    void gather_coords(const int32_t* idx_dev, const void* shift_dev, int64_t n, void* out_dev) override {
        if (n <= 0) return;
        tr("k_gather_coords");
        hipLaunchKernelGGL(k_gather_coords<T>, dim3(cdiv(n, 256)), dim3(256), 0, stream, n, idx_dev, (const T*)shift_dev, (const int32_t*)inv.p, (const T4*)pos[cur].p, (T*)out_dev);
        MHIP_HIP(hipGetLastError());
    }
    void scatter_coords(int64_t first, int64_t n, const void* in_dev) override {
        if (n <= 0) return;
        if (first < 0 || first + n > n_tot) throw ApiError{MHIP_ERR_INVALID, "scatter_coords range out of bounds"};
        tr("k_scatter_coords");
        hipLaunchKernelGGL(k_scatter_coords<T>, dim3(cdiv(n, 256)), dim3(256), 0, stream, first, n, (const T*)in_dev, (const int32_t*)inv.p, pos[cur].p);
        MHIP_HIP(hipGetLastError());
    }
};

}  // namespace mhip

// ====================================================================================================
// C ABI
// ====================================================================================================
struct mhip_ctx { std::unique_ptr<mhip::EngineBase> e; };

namespace {
template <class F> int32_t guard(mhip_ctx* ctx, F&& f) {
    std::string* err = ctx && ctx->e ? &ctx->e->err : &mhip::g_create_error;
    try { f(); err->clear(); return MHIP_OK; }
    catch (const mhip::ApiError& a) { *err = a.msg; return a.code; }
    catch (const mhip::HipErr& h) { *err = std::string(h.what) + ": " + hipGetErrorString(h.e); return MHIP_ERR_HIP; }
    catch (const std::bad_alloc&) { *err = "out of host memory"; return MHIP_ERR_HIP; }
    catch (const std::exception& x) { *err = x.what(); return MHIP_ERR_INVALID; }
}
#define NEED_CTX() if (!ctx || !ctx->e) { mhip::g_create_error = "null context"; return MHIP_ERR_INVALID; }
}  // namespace

extern "C" {

int32_t mhip_device_count(int32_t* n_out) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    if (n_out) *n_out = n;
    return MHIP_OK;
}

int32_t mhip_create(mhip_ctx** out, const mhip_config* cfg) {
    if (!out || !cfg) { mhip::g_create_error = "null argument"; return MHIP_ERR_INVALID; }
    *out = nullptr;
    mhip_ctx* c = new (std::nothrow) mhip_ctx;
    if (!c) { mhip::g_create_error = "out of host memory"; return MHIP_ERR_HIP; }
    int32_t rc = guard(nullptr, [&] {
        if (cfg->precision == 32) c->e.reset(new mhip::Engine<float>(*cfg));
        else if (cfg->precision == 64) c->e.reset(new mhip::Engine<double>(*cfg));
        else throw mhip::ApiError{MHIP_ERR_INVALID, "precision must be 32 or 64"};
    });
    if (rc != MHIP_OK) { delete c; return rc; }
    *out = c;
    return MHIP_OK;
}

int32_t mhip_destroy(mhip_ctx* ctx) { delete ctx; return MHIP_OK; }

const char* mhip_last_error(const mhip_ctx* ctx) { return (ctx && ctx->e) ? ctx->e->err.c_str() : mhip::g_create_error.c_str(); }

int32_t mhip_set_stream(mhip_ctx* ctx, void* s) { NEED_CTX(); return guard(ctx, [&] { ctx->e->set_stream(s); }); }
int32_t mhip_synchronize(mhip_ctx* ctx) { NEED_CTX(); return guard(ctx, [&] { ctx->e->synchronize(); }); }
int32_t mhip_set_profiling(mhip_ctx* ctx, int32_t on) { NEED_CTX(); return guard(ctx, [&] { ctx->e->set_profiling(on != 0); }); }
int32_t mhip_set_atom_counts(mhip_ctx* ctx, int64_t no, int64_t ng) { NEED_CTX(); return guard(ctx, [&] { ctx->e->set_atom_counts(no, ng); }); }
int32_t mhip_set_atoms(mhip_ctx* ctx, const void* q, const void* s, const void* e, const void* m, const void* l, int32_t mk) {
    NEED_CTX(); return guard(ctx, [&] { ctx->e->set_atoms(q, s, e, m, l, mk); });
}
int32_t mhip_set_exceptions(mhip_ctx* ctx, const int32_t* ei, const int32_t* ej, int64_t ne, const int32_t* si, const int32_t* sj, int64_t ns) {
    NEED_CTX(); return guard(ctx, [&] { if (ne < 0 || ns < 0) throw mhip::ApiError{MHIP_ERR_INVALID, "negative count"}; ctx->e->set_exceptions(ei, ej, ne, si, sj, ns); });
}
int32_t mhip_set_bonds(mhip_ctx* ctx, int64_t n, const int32_t* i, const int32_t* j, const void* k, const void* r0) { NEED_CTX(); return guard(ctx, [&] { ctx->e->set_bonds(n, i, j, k, r0); }); }
int32_t mhip_set_angles(mhip_ctx* ctx, int64_t n, const int32_t* i, const int32_t* j, const int32_t* k, const void* kth, const void* th0) {
    NEED_CTX(); return guard(ctx, [&] { ctx->e->set_angles(n, i, j, k, kth, th0); });
}
int32_t mhip_set_torsions(mhip_ctx* ctx, int64_t n, const int32_t* i, const int32_t* j, const int32_t* k, const int32_t* l, const int32_t* per, const void* ph, const void* kt) {
    NEED_CTX(); return guard(ctx, [&] { ctx->e->set_torsions(n, i, j, k, l, per, ph, kt); });
}
int32_t mhip_set_ewald_exclusions(mhip_ctx* ctx, int64_t n, const int32_t* i, const int32_t* j) { NEED_CTX(); return guard(ctx, [&] { ctx->e->set_ewald_exclusions(n, i, j); }); }
int32_t mhip_set_state(mhip_ctx* ctx, const void* x, const void* v, int32_t mk) { NEED_CTX(); return guard(ctx, [&] { ctx->e->set_state(x, v, mk); }); }
int32_t mhip_get_state(mhip_ctx* ctx, void* x, void* v, int32_t mk) { NEED_CTX(); return guard(ctx, [&] { ctx->e->get_state(x, v, mk); }); }
int32_t mhip_forces(mhip_ctx* ctx, int64_t step_n, int32_t acc, void* f, void* virial9, int32_t mk) {
    NEED_CTX();
    return guard(ctx, [&] {
        if (!f) throw mhip::ApiError{MHIP_ERR_INVALID, "null force buffer"};
        ctx->e->forces(step_n, acc, f, mk);
        if (virial9) ctx->e->pairwise_virial(step_n, static_cast<double*>(virial9));
    });
}
int32_t mhip_specific_forces(mhip_ctx* ctx, int32_t acc, void* f, int32_t mk) { NEED_CTX(); return guard(ctx, [&] { if (!f) throw mhip::ApiError{MHIP_ERR_INVALID, "null force buffer"}; ctx->e->specific_forces(acc, f, mk); }); }
int32_t mhip_potential_energy(mhip_ctx* ctx, int64_t step_n, double* pe) { NEED_CTX(); return guard(ctx, [&] { *pe = ctx->e->potential_energy(step_n); }); }
int32_t mhip_specific_potential_energy(mhip_ctx* ctx, double* pe) { NEED_CTX(); return guard(ctx, [&] { *pe = ctx->e->specific_potential_energy(); }); }
int32_t mhip_kinetic_energy(mhip_ctx* ctx, double* ke) { NEED_CTX(); return guard(ctx, [&] { *ke = ctx->e->kinetic_energy(); }); }
int32_t mhip_remove_cm(mhip_ctx* ctx) { NEED_CTX(); return guard(ctx, [&] { ctx->e->remove_cm(); }); }
int32_t mhip_check_finite(mhip_ctx* ctx) { NEED_CTX(); return guard(ctx, [&] { ctx->e->check_finite(); }); }
int32_t mhip_vv_run(mhip_ctx* ctx, int64_t first, int64_t n, double dt, int32_t cm) {
    NEED_CTX(); return guard(ctx, [&] { if (n < 0 || !(dt > 0)) throw mhip::ApiError{MHIP_ERR_INVALID, "n_steps must be >= 0 and dt > 0"}; ctx->e->vv_run(first, n, dt, cm); });
}
int32_t mhip_vv_init(mhip_ctx* ctx, int64_t first) { NEED_CTX(); return guard(ctx, [&] { ctx->e->vv_init(first); }); }
int32_t mhip_vv_stage1(mhip_ctx* ctx, double dt) { NEED_CTX(); return guard(ctx, [&] { ctx->e->vv_stage1(dt); }); }
int32_t mhip_vv_stage2(mhip_ctx* ctx, int64_t step_n, double dt) { NEED_CTX(); return guard(ctx, [&] { ctx->e->vv_stage2(step_n, dt); }); }
int32_t mhip_rebuild(mhip_ctx* ctx, int64_t step_n) { NEED_CTX(); return guard(ctx, [&] { ctx->e->rebuild_now(step_n); }); }
int32_t mhip_export_neighbors(mhip_ctx* ctx, int32_t* i, int32_t* j, uint8_t* sp, int64_t capacity, int64_t* n_out) {
    NEED_CTX();
    return guard(ctx, [&] {
        int64_t n = ctx->e->export_neighbors(i, j, sp, capacity);
        if (n_out) *n_out = n;
        if (i && n > capacity) throw mhip::ApiError{MHIP_ERR_CAPACITY, "export_neighbors: buffer too small"};
    });
}
int32_t mhip_export_order(mhip_ctx* ctx, int32_t* perm, int64_t capacity) { NEED_CTX(); return guard(ctx, [&] { ctx->e->export_order(perm, capacity); }); }
int32_t mhip_get_stats(mhip_ctx* ctx, mhip_stats* out) { NEED_CTX(); return guard(ctx, [&] { ctx->e->get_stats(out); }); }
int32_t mhip_gather_coords(mhip_ctx* ctx, const int32_t* idx, const void* shift, int64_t n, void* out) { NEED_CTX(); return guard(ctx, [&] { ctx->e->gather_coords(idx, shift, n, out); }); }
int32_t mhip_scatter_coords(mhip_ctx* ctx, int64_t first, int64_t n, const void* in) { NEED_CTX(); return guard(ctx, [&] { ctx->e->scatter_coords(first, n, in); }); }
int32_t mhip_cm_momentum(mhip_ctx* ctx, double* out4) { NEED_CTX(); return guard(ctx, [&] { ctx->e->cm_momentum(out4); }); }
int32_t mhip_shift_velocities(mhip_ctx* ctx, const double* dv3) { NEED_CTX(); return guard(ctx, [&] { ctx->e->shift_velocities(dv3); }); }
int32_t mhip_cm_momentum_dev(mhip_ctx* ctx, double* out4) { NEED_CTX(); return guard(ctx, [&] { ctx->e->cm_momentum_dev(out4); }); }
int32_t mhip_remove_cm_dev(mhip_ctx* ctx, const double* t4) { NEED_CTX(); return guard(ctx, [&] { ctx->e->remove_cm_dev(t4); }); }
int32_t mhip_langevin_run(mhip_ctx* ctx, int64_t first, int64_t n, double dt, double kT, double friction, int32_t cm_every, uint64_t key, uint64_t ctr1) {
    NEED_CTX(); return guard(ctx, [&] {
        if (first < 0 || n < 0 || !(dt > 0)) throw mhip::ApiError{MHIP_ERR_INVALID, "first_step and n_steps must be non-negative, dt positive"};
        ctx->e->langevin_run(first, n, dt, kT, friction, cm_every, key, ctr1); });
}
int32_t mhip_random_velocities(mhip_ctx* ctx, double kT, uint64_t key, uint64_t ctr1) { NEED_CTX(); return guard(ctx, [&] { ctx->e->redraw_velocities(1, kT, 1.0, key, ctr1); }); }
int32_t mhip_andersen(mhip_ctx* ctx, double kT, double prob, uint64_t key, uint64_t ctr1) { NEED_CTX(); return guard(ctx, [&] { ctx->e->redraw_velocities(0, kT, prob, key, ctr1); }); }
int32_t mhip_set_andersen(mhip_ctx* ctx, double kT, double prob, uint64_t seed) { NEED_CTX(); return guard(ctx, [&] { ctx->e->set_andersen(kT, prob, seed); }); }
int32_t mhip_philox4x32_10(const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4) {
    if (!ctr4 || !key2 || !out4) return MHIP_ERR_INVALID;
    uint32_t *in = nullptr, *out = nullptr, h[6] = {ctr4[0], ctr4[1], ctr4[2], ctr4[3], key2[0], key2[1]};
    if (hipMalloc((void**)&in, sizeof(h)) != hipSuccess || hipMalloc((void**)&out, 4 * sizeof(uint32_t)) != hipSuccess) { (void)hipFree(in); return MHIP_ERR_HIP; }
    bool ok = hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) { mhip::launch_philox_probe(nullptr, in, out); ok = hipMemcpy(out4, out, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess; }
    (void)hipFree(in); (void)hipFree(out);
    return ok ? MHIP_OK : MHIP_ERR_HIP;
}
int32_t mhip_specific_virial(mhip_ctx* ctx, double* out9) { NEED_CTX(); return guard(ctx, [&] { if (!out9) throw mhip::ApiError{MHIP_ERR_INVALID, "null output"}; ctx->e->specific_virial(out9); }); }
int32_t mhip_general_virial(mhip_ctx* ctx, double* out9) { NEED_CTX(); return guard(ctx, [&] { if (!out9) throw mhip::ApiError{MHIP_ERR_INVALID, "null output"}; ctx->e->general_virial(out9); }); }
int32_t mhip_set_box(mhip_ctx* ctx, const double* box3, const double* basis9) {
    NEED_CTX(); return guard(ctx, [&] { if (!box3) throw mhip::ApiError{MHIP_ERR_INVALID, "null box"}; ctx->e->set_box(box3, basis9); });
}
int32_t mhip_set_triclinic(mhip_ctx* ctx, const double* basis9, int32_t approx_images) {
    NEED_CTX(); return guard(ctx, [&] { if (!basis9) throw mhip::ApiError{MHIP_ERR_INVALID, "null basis"}; ctx->e->set_triclinic(basis9, approx_images); });
}
int32_t mhip_set_pme(mhip_ctx* ctx, int32_t order, const int32_t* mesh, double alpha, double eps_r) {
    NEED_CTX(); return guard(ctx, [&] { if (order != 0 && !mesh) throw mhip::ApiError{MHIP_ERR_INVALID, "null mesh"}; ctx->e->set_pme(order, mesh, alpha, eps_r); });
}
int32_t mhip_general_forces(mhip_ctx* ctx, int32_t acc, void* f, int32_t mk) { NEED_CTX(); return guard(ctx, [&] { if (!f) throw mhip::ApiError{MHIP_ERR_INVALID, "null force buffer"}; ctx->e->general_forces(acc, f, mk); }); }
int32_t mhip_general_potential_energy(mhip_ctx* ctx, double* pe) { NEED_CTX(); return guard(ctx, [&] { *pe = ctx->e->general_potential_energy(); }); }
int32_t mhip_vv_halo_end_parts(mhip_ctx* ctx, int64_t step_n, double dt, int64_t first, int64_t n, const void* in, double* cm_parts, int32_t n_parts) {
    NEED_CTX(); return guard(ctx, [&] { ctx->e->halo_end_parts(step_n, dt, first, n, in, cm_parts, n_parts); });
}
int32_t mhip_remove_cm_parts_dev(mhip_ctx* ctx, const double* parts, int32_t n_parts) { NEED_CTX(); return guard(ctx, [&] { if (!parts) throw mhip::ApiError{MHIP_ERR_INVALID, "null partials"}; ctx->e->remove_cm_parts_dev(parts, n_parts); }); }
int32_t mhip_set_ghost_margin(mhip_ctx* ctx, double m) { NEED_CTX(); return guard(ctx, [&] { ctx->e->set_ghost_margin(m); }); }
int32_t mhip_plan_disp2_dev(mhip_ctx* ctx, float* out) { NEED_CTX(); return guard(ctx, [&] { if (!out) throw mhip::ApiError{MHIP_ERR_INVALID, "null output"}; ctx->e->plan_disp2_dev(out); }); }
int32_t mhip_request_prune(mhip_ctx* ctx) { NEED_CTX(); return guard(ctx, [&] { ctx->e->request_prune(); }); }
int32_t mhip_plan_state_dev(mhip_ctx* ctx, float* out3) { NEED_CTX(); return guard(ctx, [&] { if (!out3) throw mhip::ApiError{MHIP_ERR_INVALID, "null output"}; ctx->e->plan_state_dev(out3); }); }
int32_t mhip_plan_decide(mhip_ctx* ctx, int64_t step_n, const float* reduced3, int32_t* action, int32_t* check_in) {
    NEED_CTX(); return guard(ctx, [&] { if (!reduced3 || !action) throw mhip::ApiError{MHIP_ERR_INVALID, "null argument"}; *action = ctx->e->plan_decide(step_n, reduced3, check_in); });
}
int32_t mhip_set_halo_plan(mhip_ctx* ctx, const mhip_halo_plan* plan) { NEED_CTX(); return guard(ctx, [&] { ctx->e->set_halo_plan(plan); }); }
int32_t mhip_vv_halo_start(mhip_ctx* ctx, double dt) { NEED_CTX(); return guard(ctx, [&] { ctx->e->halo_start(dt); }); }
int32_t mhip_vv_halo_mid(mhip_ctx* ctx, int64_t step_n, double dt, int32_t flags, double* cm_parts, int32_t n_parts) {
    NEED_CTX(); return guard(ctx, [&] { ctx->e->halo_mid(step_n, dt, flags, cm_parts, n_parts); });
}
int32_t mhip_halo_region(mhip_ctx* ctx, int64_t rows_capacity, int32_t world, int32_t rank, void* ipc_handle_out) { NEED_CTX(); return guard(ctx, [&] { ctx->e->halo_region(rows_capacity, world, rank, ipc_handle_out); }); }
int32_t mhip_halo_open_peer(mhip_ctx* ctx, int32_t rank, const void* ipc_handle) { NEED_CTX(); return guard(ctx, [&] { ctx->e->halo_open_peer(rank, ipc_handle); }); }
int32_t mhip_set_launch_config(mhip_ctx* ctx, int32_t block_atoms, int32_t j_split) { NEED_CTX(); return guard(ctx, [&] { ctx->e->set_launch_config(block_atoms, j_split); }); }
int32_t mhip_optimize_launch_config(mhip_ctx* ctx, int32_t n_passes, mhip_launch_trial* trials, int32_t max_trials, int32_t* n_trials) {
    NEED_CTX();
    return guard(ctx, [&] {
        if (max_trials < 0 || (max_trials > 0 && !trials)) throw mhip::ApiError{MHIP_ERR_INVALID, "trials / max_trials"};
        const int n = ctx->e->tune_launch(n_passes, trials, max_trials);
        if (n_trials) *n_trials = n;
    });
}
int32_t mhip_halo_selftest(mhip_ctx* ctx, int32_t* ok) { NEED_CTX(); return guard(ctx, [&] { if (!ok) throw mhip::ApiError{MHIP_ERR_INVALID, "null output"}; *ok = ctx->e->halo_selftest(); }); }
int32_t mhip_set_halo_routes(mhip_ctx* ctx, const mhip_halo_routes* routes) { NEED_CTX(); return guard(ctx, [&] { ctx->e->set_halo_routes(routes); }); }
int32_t mhip_domain_run(mhip_ctx* ctx, int64_t first_step, int64_t n_steps, double dt, int32_t remove_cm_every, double* cm_parts_dev, int32_t n_parts,
                        int64_t* steps_done, int32_t* reason, int64_t* counters3) {
    NEED_CTX(); return guard(ctx, [&] {
        if (!steps_done || !reason || !(dt > 0) || n_steps < 0) throw mhip::ApiError{MHIP_ERR_INVALID, "mhip_domain_run: null output, dt <= 0 or n_steps < 0"};
        if (remove_cm_every != 0 && (!cm_parts_dev || n_parts < 1 || n_parts > 1024)) throw mhip::ApiError{MHIP_ERR_INVALID, "mhip_domain_run: n_parts must be 1..1024 when the centre-of-mass motion is removed"};
        ctx->e->domain_run(first_step, n_steps, dt, remove_cm_every, cm_parts_dev, n_parts, steps_done, reason, counters3); });
}
int32_t mhip_set_domain(mhip_ctx* ctx, const mhip_domain_geometry* g, const int64_t* gids_dev) { NEED_CTX(); return guard(ctx, [&] { ctx->e->set_domain(g, gids_dev); }); }
int32_t mhip_domain_info(mhip_ctx* ctx, int64_t* out8) { NEED_CTX(); return guard(ctx, [&] { if (!out8) throw mhip::ApiError{MHIP_ERR_INVALID, "null output"}; ctx->e->domain_info(out8); }); }
int32_t mhip_domain_export(mhip_ctx* ctx, int64_t* gid_dev, void* par4_dev) { NEED_CTX(); return guard(ctx, [&] { ctx->e->domain_export(gid_dev, par4_dev); }); }
int32_t mhip_vv_halo_begin(mhip_ctx* ctx, double dt, const int32_t* idx, const void* shift, int64_t n, void* out) { NEED_CTX(); return guard(ctx, [&] { ctx->e->halo_begin(dt, idx, shift, n, out); }); }
int32_t mhip_vv_halo_interior(mhip_ctx* ctx, int64_t step_n, int32_t* launched) { NEED_CTX(); return guard(ctx, [&] { int r = ctx->e->halo_interior(step_n); if (launched) *launched = r; }); }
int32_t mhip_vv_halo_end(mhip_ctx* ctx, int64_t step_n, double dt, int64_t first, int64_t n, const void* in, double* cm_out4) {
    NEED_CTX(); return guard(ctx, [&] { ctx->e->halo_end(step_n, dt, first, n, in, cm_out4); });
}

}  // extern "C"
