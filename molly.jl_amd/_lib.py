"""ctypes binding of libmollyhip.so (include/mollyhip.h).  There is no CPU fallback: if the HIP library is
missing or no gfx950 device is visible, calls fail loudly."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOLLYHIP_LIB_AB") or os.path.join(_HERE, "libmollyhip.so")   # (MOLLYHIP_LIB_AB: another build of the library, for A/B timing on one box)

MEM_HOST, MEM_DEVICE = 0, 1
CUTOFF_NONE, CUTOFF_DISTANCE, CUTOFF_SHIFTED_POTENTIAL, CUTOFF_SHIFTED_FORCE, CUTOFF_CUBIC_SPLINE, CUTOFF_POLYNOMIAL = range(6)
COUL_NONE, COUL_PLAIN, COUL_REACTION_FIELD, COUL_EWALD_DIRECT = range(4)

STATUS = {0: "MHIP_OK", -1: "MHIP_ERR_INVALID", -2: "MHIP_ERR_HIP", -3: "MHIP_ERR_STATE", -4: "MHIP_ERR_CAPACITY",
          -5: "MHIP_ERR_NO_DEVICE", -6: "MHIP_ERR_UNSUPPORTED", -7: "MHIP_ERR_NAN"}


class MollyHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class Interactions(C.Structure):
    _fields_ = [("lj_enabled", C.c_int32), ("lj_cutoff_kind", C.c_int32), ("lj_rc", C.c_double), ("lj_ra", C.c_double),
                ("lj_weight_special", C.c_double), ("coul_kind", C.c_int32), ("coul_cutoff_kind", C.c_int32),
                ("coul_rc", C.c_double), ("coul_ra", C.c_double), ("coul_ke", C.c_double),
                ("coul_weight_special", C.c_double), ("rf_dielectric", C.c_double), ("ewald_alpha", C.c_double),
                ("ewald_approx_erfc", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class Config(C.Structure):
    _fields_ = [("precision", C.c_int32), ("device_id", C.c_int32), ("n_atoms", C.c_int64), ("box", C.c_double * 3),
                ("origin", C.c_double * 3), ("periodic", C.c_int32 * 3), ("rebuild_every", C.c_int32),
                ("r_list", C.c_double), ("inter", Interactions)]


class Stats(C.Structure):
    _fields_ = [("n_atoms", C.c_int64), ("n_owned", C.c_int64), ("n_ghost", C.c_int64), ("n_rebuilds", C.c_int64),
                ("n_force_calls", C.c_int64), ("n_pairs_full", C.c_int64), ("n_list_slots", C.c_int64),
                ("n_blocks", C.c_int64), ("tile_atoms_total", C.c_int64), ("block_atoms", C.c_int32),
                ("j_split", C.c_int32), ("minimg_mode", C.c_int32), ("max_tile_atoms", C.c_int32),
                ("last_rebuild_ms", C.c_double), ("lds_bytes", C.c_int64), ("algorithmic_bytes_step", C.c_int64),
                ("force_pass_bytes", C.c_int64), ("prof_ms", C.c_double * 8), ("prof_calls", C.c_int64 * 8),
                ("n_outer_builds", C.c_int64), ("n_filter_passes", C.c_int64), ("tile_segments", C.c_int64),
                ("n_group_split_passes", C.c_int64), ("group_split", C.c_int32), ("n_adopted_outer_lists", C.c_int32),
                ("n_fused_steps", C.c_int64), ("n_outer_slots", C.c_int64), ("outer_tile_atoms_total", C.c_int64),
                ("build_pass_bytes", C.c_int64), ("prune_pass_bytes", C.c_int64), ("n_box_changes", C.c_int64)]

    def as_dict(self):
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["prof_ms"] = list(self.prof_ms); d["prof_calls"] = list(self.prof_calls)
        return d


class HaloPlan(C.Structure):
    """mhip_halo_plan: the device buffers of one ghost plan (include/mollyhip.h)"""
    _fields_ = [("first_ghost", C.c_int64), ("n_recv_rows", C.c_int64), ("recv", C.c_void_p), ("recv_dst", C.c_void_p),
                ("n_cm_peers", C.c_int32), ("cm_rows", C.c_int32), ("send_idx", C.c_void_p), ("send_shift", C.c_void_p),
                ("n_send_rows", C.c_int64), ("send", C.c_void_p), ("send_cm_pos", C.c_void_p), ("n_send_cm", C.c_int32)]


class HaloRoutes(C.Structure):
    """mhip_halo_routes: where the segments of a ghost plan's send buffer land (host arrays)"""
    _fields_ = [("n_peers", C.c_int32), ("peer_rank", C.POINTER(C.c_int32)), ("send_rows", C.POINTER(C.c_int64)),
                ("dst_row", C.POINTER(C.c_int64)), ("recv_rows", C.POINTER(C.c_int64))]


class DomainGeometry(C.Structure):
    """mhip_domain_geometry: the brick decomposition as one rank sees it (the re-plan inside the engine)"""
    _fields_ = [("grid", C.c_int32 * 3), ("rank", C.c_int32), ("box", C.c_double * 3), ("r_ghost", C.c_double)]


class LaunchTrial(C.Structure):
    """mhip_launch_trial: one timed workgroup shape of mhip_optimize_launch_config"""
    _fields_ = [("block_atoms", C.c_int32), ("j_split", C.c_int32), ("us_per_pass", C.c_float)]


IPC_HANDLE_BYTES = 64

# every entry point of include/mollyhip.h: name -> (restype, argtypes)
_P, _I32, _I64, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_double
SIGNATURES = {
    "mhip_create": (_I32, [C.POINTER(_P), C.POINTER(Config)]),
    "mhip_destroy": (_I32, [_P]),
    "mhip_last_error": (C.c_char_p, [_P]),
    "mhip_device_count": (_I32, [C.POINTER(_I32)]),
    "mhip_set_stream": (_I32, [_P, _P]),
    "mhip_synchronize": (_I32, [_P]),
    "mhip_set_profiling": (_I32, [_P, _I32]),
    "mhip_set_atom_counts": (_I32, [_P, _I64, _I64]),
    "mhip_set_atoms": (_I32, [_P, _P, _P, _P, _P, _P, _I32]),
    "mhip_set_exceptions": (_I32, [_P, _P, _P, _I64, _P, _P, _I64]),
    "mhip_set_bonds": (_I32, [_P, _I64, _P, _P, _P, _P]),
    "mhip_set_angles": (_I32, [_P, _I64, _P, _P, _P, _P, _P]),
    "mhip_set_torsions": (_I32, [_P, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "mhip_set_ewald_exclusions": (_I32, [_P, _I64, _P, _P]),
    "mhip_set_state": (_I32, [_P, _P, _P, _I32]),
    "mhip_get_state": (_I32, [_P, _P, _P, _I32]),
    "mhip_forces": (_I32, [_P, _I64, _I32, _P, _P, _I32]),
    "mhip_specific_forces": (_I32, [_P, _I32, _P, _I32]),
    "mhip_potential_energy": (_I32, [_P, _I64, C.POINTER(_D)]),
    "mhip_specific_potential_energy": (_I32, [_P, C.POINTER(_D)]),
    "mhip_kinetic_energy": (_I32, [_P, C.POINTER(_D)]),
    "mhip_remove_cm": (_I32, [_P]),
    "mhip_check_finite": (_I32, [_P]),
    "mhip_vv_run": (_I32, [_P, _I64, _I64, _D, _I32]),
    "mhip_vv_init": (_I32, [_P, _I64]),
    "mhip_vv_stage1": (_I32, [_P, _D]),
    "mhip_vv_stage2": (_I32, [_P, _I64, _D]),
    "mhip_rebuild": (_I32, [_P, _I64]),
    "mhip_export_neighbors": (_I32, [_P, _P, _P, _P, _I64, C.POINTER(_I64)]),
    "mhip_export_order": (_I32, [_P, _P, _I64]),
    "mhip_get_stats": (_I32, [_P, C.POINTER(Stats)]),
    "mhip_gather_coords": (_I32, [_P, _P, _P, _I64, _P]),
    "mhip_scatter_coords": (_I32, [_P, _I64, _I64, _P]),
    "mhip_cm_momentum": (_I32, [_P, C.POINTER(_D * 4)]),
    "mhip_shift_velocities": (_I32, [_P, C.POINTER(_D * 3)]),
    "mhip_cm_momentum_dev": (_I32, [_P, _P]),
    "mhip_remove_cm_dev": (_I32, [_P, _P]),
    "mhip_langevin_run": (_I32, [_P, _I64, _I64, _D, _D, _D, _I32, C.c_uint64, C.c_uint64]),
    "mhip_random_velocities": (_I32, [_P, _D, C.c_uint64, C.c_uint64]),
    "mhip_andersen": (_I32, [_P, _D, _D, C.c_uint64, C.c_uint64]),
    "mhip_set_andersen": (_I32, [_P, _D, _D, C.c_uint64]),
    "mhip_philox4x32_10": (_I32, [_P, _P, _P]),
    "mhip_specific_virial": (_I32, [_P, _P]),
    "mhip_general_virial": (_I32, [_P, _P]),
    "mhip_set_pme": (_I32, [_P, _I32, _P, _D, _D]),
    "mhip_set_triclinic": (_I32, [_P, _P, _I32]),
    "mhip_set_box": (_I32, [_P, _P, _P]),
    "mhip_general_forces": (_I32, [_P, _I32, _P, _I32]),
    "mhip_general_potential_energy": (_I32, [_P, C.POINTER(_D)]),
    "mhip_set_ghost_margin": (_I32, [_P, _D]),
    "mhip_plan_disp2_dev": (_I32, [_P, _P]),
    "mhip_request_prune": (_I32, [_P]),
    "mhip_vv_halo_begin": (_I32, [_P, _D, _P, _P, _I64, _P]),
    "mhip_vv_halo_end": (_I32, [_P, _I64, _D, _I64, _I64, _P, _P]),
    "mhip_vv_halo_interior": (_I32, [_P, _I64, C.POINTER(_I32)]),
    "mhip_vv_halo_end_parts": (_I32, [_P, _I64, _D, _I64, _I64, _P, _P, _I32]),
    "mhip_remove_cm_parts_dev": (_I32, [_P, _P, _I32]),
    "mhip_set_halo_plan": (_I32, [_P, C.POINTER(HaloPlan)]),
    "mhip_vv_halo_start": (_I32, [_P, _D]),
    "mhip_vv_halo_mid": (_I32, [_P, _I64, _D, _I32, _P, _I32]),
    "mhip_plan_state_dev": (_I32, [_P, _P]),
    "mhip_plan_decide": (_I32, [_P, _I64, C.POINTER(C.c_float), C.POINTER(_I32), C.POINTER(_I32)]),
    "mhip_halo_region": (_I32, [_P, _I64, _I32, _I32, _P]),
    "mhip_halo_open_peer": (_I32, [_P, _I32, _P]),
    "mhip_set_launch_config": (_I32, [_P, _I32, _I32]),
    "mhip_optimize_launch_config": (_I32, [_P, _I32, C.POINTER(LaunchTrial), _I32, C.POINTER(_I32)]),
    "mhip_halo_selftest": (_I32, [_P, C.POINTER(_I32)]),
    "mhip_set_halo_routes": (_I32, [_P, C.POINTER(HaloRoutes)]),
    "mhip_domain_run": (_I32, [_P, _I64, _I64, _D, _I32, _P, _I32, C.POINTER(_I64), C.POINTER(_I32), C.POINTER(_I64)]),
    "mhip_set_domain": (_I32, [_P, C.POINTER(DomainGeometry), _P]),
    "mhip_domain_info": (_I32, [_P, C.POINTER(_I64 * 8)]),
    "mhip_domain_export": (_I32, [_P, _P, _P]),
}


def build(verbose=False):
    """Compile libmollyhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MollyHipError(-5, f"{LIB_PATH} is missing: build it with molly_jl_amd.build() "
                                    "(the product path has no CPU fallback)")
        # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 with the same SONAMEs as /opt/rocm's.  The
        # first one loaded serves the whole process, and torch cannot initialise on top of the system copies — so when
        # torch is installed it is imported FIRST and libmollyhip.so binds to the runtime torch brought (one HIP runtime
        # per process: device pointers, the null stream and RCCL all live in it).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def device_count():
    n = _I32(0)
    lib().mhip_device_count(C.byref(n))
    return n.value
