"""Spatial domain decomposition of one periodic box over the GPUs of a node — one process per GPU, ghost-atom
coordinates exchanged with torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference has no multi-device path at all (README.md:54; docs/src/documentation.md:1826,1837): this module is new
design, following SURVEY.md §8(e).  The box is cut into gx×gy×gz bricks; a rank owns the atoms whose wrapped
coordinate lies in its brick and keeps, as ghosts, a full shell of width r_list of its neighbours' atoms, already
shifted by the periodic image that places them next to the brick.  Every pair that touches an owned atom is evaluated
by the owner (full shell, no ghost-force return), so one exchange of ghost COORDINATES per force evaluation is the only
data-path communication.

Per step and rank:   halo_begin (kick, drift, pack ghosts) → all_to_all_single → halo_end (unpack, forces, kick, Σmv)
                     → [all_reduce of 32 B for remove_CM_motion]
Axes that are not cut (g = 1) stay periodic inside the engine and need no ghosts.

Ghost plan lifetime.  With ghost_margin = 0 ownership and the ghost plan are redone at every neighbour-rebuild step.
With ghost_margin = Δ > 0 the shell is r_list + Δ wide.  A plan (ownership, ghost set, the engine's outer pair list) can
vouch for a prune of the inner lists as long as no atom has moved Δ/2 since it was made: until then every atom within
r_list of an owned atom is provably in the local set.  Between prunes only the skin criterion has to hold (2·displacement
since the prune ≤ r_list − cutoff), as in the single-domain engine.  The ranks agree on prune and re-plan steps through one
MAX all-reduce of two floats per rebuild interval (DomainRun.replan_if_due).
"""
import ctypes as C
import itertools
import math
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def choose_grid(world, box):
    """Factor `world` into gx·gy·gz with the most cubic bricks (1, 2×1×1, 2×2×1, 2×2×2 for a cubic box)."""
    best, best_score = (world, 1, 1), None
    for gx in range(1, world + 1):
        if world % gx:
            continue
        for gy in range(1, world // gx + 1):
            if (world // gx) % gy:
                continue
            gz = world // gx // gy
            b = (box[0] / gx, box[1] / gy, box[2] / gz)
            score = (max(b) / min(b), -gx, -gy)   # aspect ratio first, then prefer cutting x, y before z
            if best_score is None or score < best_score:
                best, best_score = (gx, gy, gz), score
    return best


import os as _os
CM_PARTS = 256   # per-rank partial sums of the centre-of-mass momentum carried by the all-reduce (8 KB)


class BrickGrid:
    """Geometry of the decomposition: brick of rank r, neighbour directions and their periodic shifts."""

    def __init__(self, box, grid, rank, r_ghost):
        self.box = np.asarray(box, dtype=np.float64)
        self.grid = tuple(int(g) for g in grid)
        self.world = self.grid[0] * self.grid[1] * self.grid[2]
        self.rank = rank
        self.r_ghost = float(r_ghost)
        self.brick = self.box / np.array(self.grid)
        self.coord = self.coords_of(rank)
        self.lo = self.brick * np.array(self.coord)
        self.hi = self.lo + self.brick
        for d in range(3):
            if self.grid[d] > 1 and self.brick[d] < self.r_ghost:
                raise ValueError(f"brick of {self.brick[d]:.3f} nm along axis {d} is narrower than the ghost reach {self.r_ghost:.3f} nm")
        # the 3^k − 1 neighbour directions over the cut axes, each mapped to (peer rank, coordinate shift of the ghosts)
        comps = [(-1, 0, 1) if g > 1 else (0,) for g in self.grid]
        dirs = []
        for dvec in itertools.product(*comps):
            if dvec == (0, 0, 0):
                continue
            peer_c, shift = [], []
            for d in range(3):
                c = self.coord[d] + dvec[d]
                s = 0.0
                if c < 0:
                    c += self.grid[d]; s = +self.box[d]      # my atom appears beyond the peer's upper face
                elif c >= self.grid[d]:
                    c -= self.grid[d]; s = -self.box[d]
                peer_c.append(c); shift.append(s)
            dirs.append((self.rank_of(tuple(peer_c)), dvec, tuple(shift)))
        dirs.sort(key=lambda t: (t[0], t[1]))                 # all_to_all wants the segments grouped by peer
        self.dirs = dirs

    def rank_of(self, c):
        return (c[2] * self.grid[1] + c[1]) * self.grid[0] + c[0]

    def coords_of(self, r):
        return (r % self.grid[0], (r // self.grid[0]) % self.grid[1], r // (self.grid[0] * self.grid[1]))

    def owner_of(self, x):
        """rank owning each wrapped coordinate (n,3) tensor"""
        r = torch.zeros(x.shape[0], dtype=torch.int64, device=x.device)
        mul = 1
        for d in range(3):
            c = torch.clamp(torch.floor(x[:, d] / self.brick[d]).to(torch.int64), 0, self.grid[d] - 1)
            r += c * mul
            mul *= self.grid[d]
        return r

    def engine_box(self, pad):
        """(box, origin, periodic) of the local engine: cut axes are open and cover brick + ghost shell + pad"""
        box, origin, periodic = [], [], []
        for d in range(3):
            if self.grid[d] == 1:
                box.append(self.box[d]); origin.append(0.0); periodic.append(1)
            else:
                box.append(self.brick[d] + 2 * (self.r_ghost + pad)); origin.append(self.lo[d] - self.r_ghost - pad); periodic.append(0)
        return box, origin, periodic


class HipDomainEngine:
    """The per-rank libmollyhip context driven through device pointers of torch tensors."""

    def __init__(self, inter, dtype, capacity, box, origin, periodic, r_list, rebuild_every, device_id, ghost_margin=0.0):
        L = _lib.lib()
        cfg = _lib.Config()
        cfg.precision = 32 if np.dtype(dtype) == np.float32 else 64
        cfg.device_id = device_id
        cfg.n_atoms = capacity
        for d in range(3):
            cfg.box[d], cfg.origin[d], cfg.periodic[d] = box[d], origin[d], periodic[d]
        cfg.rebuild_every = rebuild_every
        cfg.r_list = r_list
        cfg.inter = inter
        self.ctx = C.c_void_p()
        rc = L.mhip_create(C.byref(self.ctx), C.byref(cfg))
        if rc != 0:
            raise _lib.MollyHipError(rc, L.mhip_last_error(None).decode())
        self.L = L
        self.capacity = capacity
        self._chk(L.mhip_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        self._chk(L.mhip_set_ghost_margin(self.ctx, float(ghost_margin)))

    def _chk(self, rc):
        if rc != 0:
            raise _lib.MollyHipError(rc, self.L.mhip_last_error(self.ctx).decode())

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def set_local(self, n_owned, n_ghost, q, sigma, eps, mass, x_all, v_owned):
        if n_owned + n_ghost > self.capacity:
            raise _lib.MollyHipError(-4, f"domain holds {n_owned}+{n_ghost} atoms, engine capacity is {self.capacity}")
        self._chk(self.L.mhip_set_atom_counts(self.ctx, n_owned, n_ghost))
        self._keep = (q.contiguous(), sigma.contiguous(), eps.contiguous(), mass.contiguous(), x_all.contiguous(), v_owned.contiguous())
        self._chk(self.L.mhip_set_atoms(self.ctx, *[self._p(t) for t in self._keep[:4]], None, _lib.MEM_DEVICE))
        self._chk(self.L.mhip_set_state(self.ctx, self._p(self._keep[4]), self._p(self._keep[5]), _lib.MEM_DEVICE))

    def gather(self, idx_i32, shift, out):
        self._chk(self.L.mhip_gather_coords(self.ctx, self._p(idx_i32), self._p(shift), idx_i32.numel(), self._p(out)))

    def scatter(self, first, n, buf):
        self._chk(self.L.mhip_scatter_coords(self.ctx, first, n, self._p(buf)))

    def vv_init(self, step):
        self._chk(self.L.mhip_vv_init(self.ctx, step))

    def stage1(self, dt):
        self._chk(self.L.mhip_vv_stage1(self.ctx, dt))

    def stage2(self, step, dt):
        self._chk(self.L.mhip_vv_stage2(self.ctx, step, dt))

    def halo_begin(self, dt, idx_i32, shift, out):
        # the buffers of a ghost plan do not change between re-plans: their ctypes pointers are built once per plan
        key = (idx_i32.data_ptr(), shift.data_ptr(), out.data_ptr(), idx_i32.numel())
        if getattr(self, "_hb_key", None) != key:
            self._hb_key, self._hb = key, (self._p(idx_i32), self._p(shift), C.c_int64(idx_i32.numel()), self._p(out))
        rc = self.L.mhip_vv_halo_begin(self.ctx, dt, *self._hb)
        if rc != 0:
            self._chk(rc)

    def halo_interior(self, step):
        """pair forces of the blocks that need no ghost atom, issued while the ghost exchange is in flight; True if launched"""
        launched = C.c_int32(0)
        rc = self.L.mhip_vv_halo_interior(self.ctx, step, C.byref(launched))
        if rc != 0:
            self._chk(rc)
        return bool(launched.value)

    def halo_end(self, step, dt, first, n, buf, cm_parts):
        """cm_parts: None, or a device double tensor of 4·k entries (k per-block partials of Σ m v, Σ m — see CM_PARTS)"""
        key = (buf.data_ptr(), None if cm_parts is None else cm_parts.data_ptr())
        if getattr(self, "_he_key", None) != key:
            self._he_key, self._he = key, (self._p(buf), self._p(cm_parts), 0 if cm_parts is None else cm_parts.numel() // 4)
        rc = self.L.mhip_vv_halo_end_parts(self.ctx, step, dt, first, n, *self._he)
        if rc != 0:
            self._chk(rc)

    # -- the fused step: one engine call per step after the ghost exchange (include/mollyhip.h, mhip_halo_plan) --------
    def set_halo_plan(self, first_ghost, recv, recv_dst, n_cm_peers, cm_rows, send_idx, send_shift, send, send_cm_pos):
        hp = _lib.HaloPlan()
        hp.first_ghost, hp.n_recv_rows, hp.recv, hp.recv_dst = first_ghost, recv.shape[0], recv.data_ptr(), recv_dst.data_ptr()
        hp.n_cm_peers, hp.cm_rows = n_cm_peers, cm_rows
        hp.send_idx, hp.send_shift, hp.n_send_rows, hp.send = send_idx.data_ptr(), send_shift.data_ptr(), send.shape[0], send.data_ptr()
        hp.send_cm_pos, hp.n_send_cm = send_cm_pos.data_ptr(), send_cm_pos.numel()
        self._hp_keep = (recv, recv_dst, send_idx, send_shift, send, send_cm_pos)
        self._chk(self.L.mhip_set_halo_plan(self.ctx, C.byref(hp)))

    def halo_start(self, dt):
        rc = self.L.mhip_vv_halo_start(self.ctx, dt)
        if rc != 0:
            self._chk(rc)

    def halo_mid(self, step, dt, cm, stop, cm_parts):
        rc = self.L.mhip_vv_halo_mid(self.ctx, step, dt, (1 if cm else 0) | (2 if stop else 0), self._p(cm_parts) if (cm and stop) else None,
                                     cm_parts.numel() // 4 if (cm and stop) else 0)
        if rc != 0:
            self._chk(rc)

    # -- the exchange inside the engine (include/mollyhip.h: mhip_halo_region … mhip_domain_run) ---------------------------------
    def halo_region(self, rows_capacity, world, rank):
        """allocate this rank's receive region; returns its IPC handle (bytes) for the peers"""
        h = (C.c_ubyte * _lib.IPC_HANDLE_BYTES)()
        self._chk(self.L.mhip_halo_region(self.ctx, rows_capacity, world, rank, C.cast(h, C.c_void_p)))
        return bytes(h)

    def halo_open_peer(self, rank, handle):
        buf = (C.c_ubyte * _lib.IPC_HANDLE_BYTES).from_buffer_copy(handle)
        self._chk(self.L.mhip_halo_open_peer(self.ctx, rank, C.cast(buf, C.c_void_p)))

    def halo_selftest(self):
        ok = C.c_int32(0)
        self._chk(self.L.mhip_halo_selftest(self.ctx, C.byref(ok)))
        return bool(ok.value)

    def set_halo_routes(self, peer_rank, send_rows, dst_row, recv_rows):
        n = len(peer_rank)
        rt = _lib.HaloRoutes()
        a = ((C.c_int32 * max(n, 1))(*peer_rank), (C.c_int64 * max(n, 1))(*send_rows), (C.c_int64 * max(n, 1))(*dst_row), (C.c_int64 * max(n, 1))(*recv_rows))
        rt.n_peers, rt.peer_rank, rt.send_rows, rt.dst_row, rt.recv_rows = n, a[0], a[1], a[2], a[3]
        self._chk(self.L.mhip_set_halo_routes(self.ctx, C.byref(rt)))

    def domain_run(self, first_step, n_steps, dt, remove_cm_every, cm_parts, counters):
        """(steps_done, reason): reason 1 = stopped behind the step after which the host has to re-plan"""
        done, reason = C.c_int64(0), C.c_int32(0)
        rc = self.L.mhip_domain_run(self.ctx, first_step, n_steps, dt, int(remove_cm_every), self._p(cm_parts), cm_parts.numel() // 4, C.byref(done), C.byref(reason), counters)
        if rc != 0:
            self._chk(rc)
        return done.value, reason.value

    # -- the re-plan inside the engine (include/mollyhip.h: mhip_set_domain … mhip_domain_export) --------------------------------------
    def set_domain(self, grid, rank, box, r_ghost, gid_i64):
        g = _lib.DomainGeometry()
        for d in range(3):
            g.grid[d], g.box[d] = int(grid[d]), float(box[d])
        g.rank, g.r_ghost = int(rank), float(r_ghost)
        self._gid_keep = gid_i64.contiguous()
        self._chk(self.L.mhip_set_domain(self.ctx, C.byref(g), self._p(self._gid_keep)))

    def domain_info(self):
        """(owned atoms, ghost atoms, re-plans made by the engine, atoms that arrived in them, host µs planning, host µs searching, 0, 0)"""
        out = (C.c_int64 * 8)()
        self._chk(self.L.mhip_domain_info(self.ctx, C.byref(out)))
        return tuple(int(v) for v in out)

    def domain_export(self, gid_i64, par4):
        self._chk(self.L.mhip_domain_export(self.ctx, self._p(gid_i64), self._p(par4)))

    def plan_state(self, out3_f32):       # device float[3]: max displacement² since the plan / since the last prune, max speed²
        self._chk(self.L.mhip_plan_state_dev(self.ctx, self._p(out3_f32)))

    def plan_decide(self, step, reduced3):
        """(action, check_in): 0 nothing / 1 prune arranged / 2 re-plan; check_in > 0 = look again that many steps from now"""
        arr = (C.c_float * 3)(*reduced3)
        action, check_in = C.c_int32(0), C.c_int32(0)
        self._chk(self.L.mhip_plan_decide(self.ctx, step, arr, C.byref(action), C.byref(check_in)))
        return action.value, check_in.value

    def plan_disp2(self, out2_f32):       # device float[2]: max displacement² since the plan / since the last prune
        self._chk(self.L.mhip_plan_disp2_dev(self.ctx, self._p(out2_f32)))

    def request_prune(self):
        self._chk(self.L.mhip_request_prune(self.ctx))

    def export_neighbors(self):
        """the sub-domain's neighbour list of NOW in local indices (owned atoms first, then ghosts): pairs (i, j), i owned, i < j"""
        n = C.c_int64(0)
        self._chk(self.L.mhip_export_neighbors(self.ctx, None, None, None, 0, C.byref(n)))
        i = np.empty(n.value, np.int32); j = np.empty(n.value, np.int32); sp = np.empty(n.value, np.uint8)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(self.L.mhip_export_neighbors(self.ctx, ptr(i), ptr(j), ptr(sp), n.value, C.byref(n)))
        return i[: n.value], j[: n.value]

    def get_state(self, x_all, v_owned):
        self._chk(self.L.mhip_get_state(self.ctx, self._p(x_all), self._p(v_owned), _lib.MEM_DEVICE))

    def cm_momentum(self, out4):          # device double[4]
        self._chk(self.L.mhip_cm_momentum_dev(self.ctx, self._p(out4)))

    def remove_cm(self, total_parts):
        self._chk(self.L.mhip_remove_cm_parts_dev(self.ctx, self._p(total_parts), total_parts.numel() // 4))

    def synchronize(self):
        self._chk(self.L.mhip_synchronize(self.ctx))

    def set_profiling(self, on):
        self._chk(self.L.mhip_set_profiling(self.ctx, int(on)))

    def stats(self):
        st = _lib.Stats()
        self._chk(self.L.mhip_get_stats(self.ctx, C.byref(st)))
        return st.as_dict()

    def close(self):
        if self.ctx:
            self.L.mhip_destroy(self.ctx)
            self.ctx = None


class DomainRun:
    """One rank's share of a velocity-Verlet run.  `engine` implements set_local / vv_init / halo_begin / halo_end /
    plan_disp2 / get_state / remove_cm (HipDomainEngine on the GPU, an oracle-backed stand-in in the CPU tests).
    `grid.r_ghost` must be r_list + ghost_margin."""

    def __init__(self, grid: BrickGrid, engine, tdtype, device, rebuild_every, group=None, ghost_margin=0.0, skin=0.0):
        self.g, self.e = grid, engine
        self.gm = float(ghost_margin)
        self.skin = float(skin)              # r_list − largest cutoff: how far pairs may close in before the inner list must be re-pruned
        self.plan_step = self.prune_step = 0
        self.tdtype, self.device = tdtype, device
        self.every = rebuild_every
        self.group = group
        self.world, self.rank = grid.world, grid.rank
        self.boxt = torch.tensor(grid.box, dtype=tdtype, device=device)
        self.n_owned = self.n_ghost = 0
        # Σ m v travels as per-block partials (no finalize launch on the device): CM_PARTS × {Px, Py, Pz, M}, summed over the ranks
        self.cm_buf = torch.zeros(4 * CM_PARTS, dtype=torch.float64, device=device)
        self.d2_buf = torch.zeros(2, dtype=torch.float32, device=device)
        self.d3_buf = torch.zeros(3, dtype=torch.float32, device=device)
        self.next_check = -1                 # an extra check between two cadence steps, asked for by plan_decide
        # fused stepping (one engine call per step, Σ m v on the ghost message): needs every other rank as a peer — 1, 2, 4, 8 bricks
        peers = {p for (p, _, _) in grid.dirs}
        self.fused = (_os.environ.get("MOLLYHIP_HALO_FUSED", "1") != "0" and hasattr(engine, "halo_mid") and len(peers) == grid.world - 1)
        self.cm_rows = (3 if tdtype == torch.float32 else 2) if self.fused else 0
        # the whole step loop inside the engine, ghost rows stored straight into the peers' IPC-mapped regions (mhip_domain_run): the
        # fused layout + an engine that has the entry points + every rank on a GPU of this node; MOLLYHIP_ENGINE_LOOP=0 keeps the
        # host loop with torch.distributed collectives below (which is also what the CPU stand-in of the tests runs)
        self.engine_loop = (self.fused and hasattr(engine, "domain_run") and _os.environ.get("MOLLYHIP_ENGINE_LOOP", "1") != "0"
                            and _os.environ.get("MOLLYHIP_HOST_PRUNE", "0") == "0")
        self._ipc_ready = False
        # the re-plan inside the engine as well (mhip_set_domain): migration, ghost selection and the new routes are device compactions and peer
        # stores, mhip_domain_run never comes back for them; MOLLYHIP_DEVICE_REPLAN=0 (or an engine without the entry point) keeps migrate() below
        self.device_replan = self.engine_loop and hasattr(engine, "set_domain") and _os.environ.get("MOLLYHIP_DEVICE_REPLAN", "1") != "0"
        self._dev_replans_seen = 0
        self._dev_arrived_seen = 0
        self._counters = (C.c_int64 * 3)(0, 0, 0)
        self.stats = {"exchange_calls": 0, "ghost_atoms": 0, "migrated": 0, "plans": 0, "plan_checks": 0, "prunes": 0, "interior_passes": 0}
        self.overlap = hasattr(engine, "halo_interior")      # interior blocks while the ghosts travel (host loop)
        # gloo cannot move device memory: stage through the host (used by the multi-process tests that share ONE GPU;
        # the production path is backend "nccl" = RCCL, device buffers end to end)
        self.stage_host = torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"

    def _a2a_async(self, recv, send, rc, sc):
        """ghost exchange that leaves the compute stream free: RCCL runs it on its own stream (async_op) and the returned wait()
        makes the compute stream wait for it; the host-staged gloo form (tests: several ranks on one GPU) is synchronous, but the
        kernels issued before it keep the GPU busy while the host moves the bytes"""
        if self.stage_host or not (self.world > 1 and self.g.dirs):
            return None
        return dist.all_to_all_single(recv, send, rc, sc, group=self.group, async_op=True)

    def _a2a(self, recv, send, rc=None, sc=None):
        if self.stage_host:
            r = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(r, send.cpu(), rc, sc, group=self.group)
            recv.copy_(r)
        else:
            dist.all_to_all_single(recv, send, rc, sc, group=self.group)

    def _all_reduce(self, t, op=dist.ReduceOp.SUM):
        if self.world == 1:
            return                                           # a single rank's sum is its own
        if self.stage_host:
            c = t.cpu(); dist.all_reduce(c, op=op, group=self.group); t.copy_(c)
        else:
            dist.all_reduce(t, op=op, group=self.group)

    def _all_gather_cpu(self, t):
        """all_gather of a small CPU tensor over whatever backend the group has"""
        if dist.get_backend(self.group) == "nccl":
            d = t.to(self.device); out = [torch.empty_like(d) for _ in range(self.world)]
            dist.all_gather(out, d, group=self.group)
            return [o.cpu() for o in out]
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return out

    def _a2a_cpu(self, t):
        """all_to_all of one element per rank (CPU tensor in, CPU tensor out)"""
        if dist.get_backend(self.group) == "nccl":
            d = t.to(self.device); r = torch.empty_like(d)
            dist.all_to_all_single(r, d, group=self.group)
            return r.cpu()
        r = torch.empty_like(t)
        dist.all_to_all_single(r, t, group=self.group)
        return r

    def _all_gather(self, t):
        if self.stage_host:
            out = [torch.empty(t.shape, dtype=t.dtype) for _ in range(self.world)]
            dist.all_gather(out, t.cpu(), group=self.group)
            return out
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return out

    # -- setup from the full system (every rank generates the same synthetic system and keeps its atoms) -------------
    def setup_from_global(self, coords, velocities, charge, sigma, eps, mass, step=0):
        x = torch.as_tensor(np.asarray(coords), dtype=self.tdtype, device=self.device)
        x = x - torch.floor(x / self.boxt) * self.boxt
        mine = self.g.owner_of(x) == self.rank
        idx = torch.nonzero(mine).squeeze(1)
        tt = lambda a: torch.as_tensor(np.asarray(a), dtype=self.tdtype, device=self.device)[idx].contiguous()
        self.gid = idx.to(torch.int64)
        self.x, self.v = x[idx].contiguous(), tt(velocities)
        self.par = torch.stack([tt(charge), tt(sigma), tt(eps), tt(mass)], dim=1).contiguous()
        self._plan_and_load(step)
        self._hand_over_domain()

    def _hand_over_domain(self):
        """after a plan made by the host: the engine learns the decomposition and the global ids, and re-plans on its own from now on"""
        if not (self.device_replan and self.engine_loop):
            self.device_replan = False
            return
        try:
            self.e.set_domain(self.g.grid, self.rank, self.g.box, self.g.r_ghost, self.gid)
            info = self.e.domain_info()
            self._dev_replans_seen, self._dev_arrived_seen = info[2], info[3]
        except _lib.MollyHipError:                         # (a decomposition the device planner does not cover: the host keeps planning)
            self.device_replan = False

    # -- ghost plan: which of my atoms go to which neighbour, with which periodic shift -------------------------------
    def _plan_and_load(self, step):
        g = self.g
        n = self.x.shape[0]
        rg = g.r_ghost
        near = []   # per axis: [3, n] rows = near lower face, always, near upper face
        for d in range(3):
            xd = self.x[:, d]
            lo = xd < (g.lo[d] + rg)
            hi = xd >= (g.hi[d] - rg)
            near.append(torch.stack([lo, torch.ones_like(lo), hi]))
        if g.dirs:
            opt = torch.tensor([[dv[d] + 1 for (_, dv, _) in g.dirs] for d in range(3)], dtype=torch.int64, device=self.device)
            sel = near[0][opt[0]] & near[1][opt[1]] & near[2][opt[2]]   # [ndir, n]: one gather per axis instead of one launch per direction
            pairs = torch.nonzero(sel)                      # sorted by direction, then atom
            self.send_idx = pairs[:, 1].to(torch.int32).contiguous()
            per_dir = torch.bincount(pairs[:, 0], minlength=len(g.dirs))
            shifts = torch.tensor([s for (_, _, s) in g.dirs], dtype=self.tdtype, device=self.device)
            self.send_shift = shifts[pairs[:, 0]].contiguous()
            peer_of_dir = torch.tensor([p for (p, _, _) in g.dirs], dtype=torch.int64, device=self.device)
            send_counts = torch.zeros(self.world, dtype=torch.int64, device=self.device).index_add_(0, peer_of_dir, per_dir)
        else:
            self.send_idx = torch.zeros(0, dtype=torch.int32, device=self.device)
            self.send_shift = torch.zeros((0, 3), dtype=self.tdtype, device=self.device)
            send_counts = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        recv_counts = torch.empty_like(send_counts)
        self._a2a(recv_counts, send_counts)
        self.send_counts, self.recv_counts = send_counts.tolist(), recv_counts.tolist()     # one host sync per plan
        self._sc3, self._rc3 = [c * 3 for c in self.send_counts], [c * 3 for c in self.recv_counts]
        self.n_owned, self.n_ghost = n, int(sum(self.recv_counts))
        # ghost parameters (charge, σ, ϵ, mass) and first ghost coordinates
        send_par = self.par[self.send_idx.long()].contiguous()
        recv_par = torch.empty((self.n_ghost, 4), dtype=self.tdtype, device=self.device)
        self._a2a_rows(recv_par, send_par)
        send_x = (self.x[self.send_idx.long()] + self.send_shift).contiguous()
        self.recv_x = torch.empty((self.n_ghost, 3), dtype=self.tdtype, device=self.device)
        self._a2a_rows(self.recv_x, send_x)
        self.send_buf = torch.empty((self.send_idx.numel(), 3), dtype=self.tdtype, device=self.device)
        par_all = torch.cat([self.par, recv_par], dim=0)
        x_all = torch.cat([self.x, self.recv_x], dim=0).contiguous()
        self.e.set_local(self.n_owned, self.n_ghost, par_all[:, 0], par_all[:, 1], par_all[:, 2], par_all[:, 3], x_all, self.v)
        if self.fused:
            self._halo_layout()
        self.e.vv_init(step)                                 # neighbour structures + forces at this step
        self.stats["ghost_atoms"] = self.n_ghost
        self.stats["plans"] += 1
        self.plan_step = self.prune_step = step

    def _halo_layout(self):
        """per-step message of the fused path: each peer's coordinate rows followed by cm_rows rows for Σ m v (mhip_halo_plan)"""
        cr, dev = self.cm_rows, self.device
        peers = sorted({p for (p, _, _) in self.g.dirs})
        si, ss, cm_pos, rd = [], [], [], []
        s_off = r_off = row = 0
        cm_idx = torch.arange(-1, -1 - cr, -1, dtype=torch.int32, device=dev)
        zero_shift = torch.zeros((cr, 3), dtype=self.tdtype, device=dev)
        sc, rc = [0] * self.world, [0] * self.world
        g_off = 0
        for pi, p in enumerate(peers):
            ns, nr = self.send_counts[p], self.recv_counts[p]
            si += [self.send_idx[s_off:s_off + ns], cm_idx]; ss += [self.send_shift[s_off:s_off + ns], zero_shift]
            cm_pos += list(range(row + ns, row + ns + cr))
            rd += [torch.arange(g_off, g_off + nr, dtype=torch.int32, device=dev), torch.arange(-1 - pi * cr, -1 - (pi + 1) * cr, -1, dtype=torch.int32, device=dev)]
            s_off += ns; r_off += nr; row += ns + cr; g_off += nr
            sc[p], rc[p] = ns + cr, nr + cr
        cat = lambda ts, shape, dt: torch.cat(ts).contiguous() if ts else torch.zeros(shape, dtype=dt, device=dev)
        self.f_send_idx, self.f_send_shift = cat(si, (0,), torch.int32), cat(ss, (0, 3), self.tdtype)
        self.f_recv_dst = cat(rd, (0,), torch.int32)
        self.f_cm_pos = torch.tensor(cm_pos, dtype=torch.int32, device=dev)
        self.f_send = torch.zeros((self.f_send_idx.numel(), 3), dtype=self.tdtype, device=dev)
        self.f_recv = torch.zeros((self.f_recv_dst.numel(), 3), dtype=self.tdtype, device=dev)
        self._fsc3, self._frc3 = [c * 3 for c in sc], [c * 3 for c in rc]
        self.e.set_halo_plan(self.n_owned, self.f_recv, self.f_recv_dst, len(peers), cr, self.f_send_idx, self.f_send_shift, self.f_send, self.f_cm_pos)
        if self.engine_loop:
            self._engine_routes(peers, sc, rc)

    def _engine_routes(self, peers, sc, rc):
        """Once: every rank's receive region, its IPC handle to everybody.  Per plan: tell every peer where in MY region its segment
        starts (one small all_to_all), hand the engine the routes."""
        if self.world > 1 and not self._ipc_ready:
            ok = 1
            try:
                cap = int(getattr(self.e, "capacity", 0)) + 8 * 64
                mine = self.e.halo_region(cap, self.world, self.rank)
            except Exception:                             # (no IPC for fine-grained memory on this stack: everybody falls back together)
                ok, mine = 0, bytes(64)
            allh = self._all_gather_cpu(torch.tensor(list(mine) + [ok], dtype=torch.uint8))
            if not all(int(a[-1]) for a in allh):
                self.engine_loop = False
                return
            ok = 1
            try:
                for r, a in enumerate(allh):
                    if r != self.rank:
                        self.e.halo_open_peer(r, bytes(a[:-1].tolist()))
            except Exception:
                ok = 0
            if not all(int(a[0]) for a in self._all_gather_cpu(torch.tensor([ok], dtype=torch.uint8))):
                self.engine_loop = False                  # some rank could not map some region: the host loop for everybody
                return
            # one round over the mapped regions before anything depends on them (a store that never becomes visible on the other
            # side would otherwise surface as a time-out in the middle of a run)
            try:
                ok = 1 if self.e.halo_selftest() else 0
            except Exception:
                ok = 0
            if not all(int(a[0]) for a in self._all_gather_cpu(torch.tensor([ok], dtype=torch.uint8))):
                self.engine_loop = False
                return
            self._ipc_ready = True
        # my receive layout: peers in sorted order, rc[p] rows each → the offset of p's segment, told to p
        off, r_off = torch.zeros(self.world, dtype=torch.int64), 0
        for p in peers:
            off[p] = r_off; r_off += rc[p]
        theirs = self._a2a_cpu(off) if self.world > 1 else off
        self.e.set_halo_routes(list(peers), [sc[p] for p in peers], [int(theirs[p]) for p in peers], [rc[p] for p in peers])

    def _a2a_rows(self, recv, send):
        w = recv.shape[1]
        self._a2a(recv.view(-1), send.view(-1), [c * w for c in self.recv_counts], [c * w for c in self.send_counts])

    # -- one MD step ------------------------------------------------------------------------------------------------
    def step(self, step_n, dt, remove_cm_every=1):
        cm = bool(remove_cm_every) and step_n % remove_cm_every == 0
        self.e.halo_begin(dt, self.send_idx, self.send_shift, self.send_buf)      # kick, drift, pack my atoms the peers need
        exchange = self.world > 1 and bool(self.g.dirs)
        work = self._a2a_async(self.recv_x.view(-1), self.send_buf.view(-1), self._rc3, self._sc3) if (exchange and self.overlap) else None
        if self.overlap and self.e.halo_interior(step_n):                          # blocks without ghosts: computed while the ghosts travel
            self.stats["interior_passes"] += 1
        if exchange:
            if work is not None:
                work.wait()                                                       # the compute stream waits for the exchange here
            else:
                self._a2a(self.recv_x.view(-1), self.send_buf.view(-1), self._rc3, self._sc3)
            self.stats["exchange_calls"] += 1
        self.e.halo_end(step_n, dt, self.n_owned, self.n_ghost, self.recv_x, self.cm_buf if cm else None)   # unpack, forces, kick
        if cm:
            self._all_reduce(self.cm_buf)
            self.e.remove_cm(self.cm_buf)              # applied by the next halo_begin
        if step_n % self.every == 0 or step_n == self.next_check:      # (next_check: the engine vouched for the inner list for fewer than
            self.replan_if_due(step_n)                                  #  `every` steps at the last decision, mhip_plan_decide's check_in)

    def replan_if_due(self, step_n):
        if self._replan_due(step_n):
            self.migrate(step_n)

    def _replan_due(self, step_n):
        """Collective decision at the rebuild cadence (one MAX all-reduce of a few floats, one host sync): True = ownership, ghosts and
        outer lists have to be redone.  The inner pair lists are re-pruned — on every rank at the same step — when the displacement
        since the last prune is about to use up the inner skin.  A prune needs the ghost plan to be valid at that moment (nobody
        moved more than half the margin since it was made): if it is not, the plan is redone instead.  In between, nothing needs to
        hold but the skin criterion, exactly as in the single-domain engine."""
        if self.gm <= 0:
            return True
        if hasattr(self.e, "plan_decide") and _os.environ.get("MOLLYHIP_HOST_PRUNE", "0") == "0":
            # the engine's own criteria (tight inner skin, drift bound from the fastest atom), fed with the MAX over the ranks
            self.e.plan_state(self.d3_buf)
            self._all_reduce(self.d3_buf, dist.ReduceOp.MAX)
            red = [float(v) for v in self.d3_buf.tolist()]          # the one host sync per rebuild interval
            self.stats["plan_checks"] += 1
            action, check_in = (2, 0) if math.isinf(red[0]) else self.e.plan_decide(step_n, red)
            self.next_check = step_n + check_in if check_in > 0 else -1
            if action == 1:
                self.prune_step = step_n
                self.stats["prunes"] += 1
            return action == 2
        self.e.plan_disp2(self.d2_buf)
        self._all_reduce(self.d2_buf, dist.ReduceOp.MAX)
        d2_plan, d2_prune = (float(v) for v in self.d2_buf.tolist())      # the one host sync per rebuild interval
        self.stats["plan_checks"] += 1
        if math.isinf(d2_plan):
            return True
        k = max(1, (step_n - self.prune_step) // self.every)
        prune_due = math.isinf(d2_prune) or 2.0 * math.sqrt(d2_prune) * (k + 1) / k > 0.98 * self.skin
        if not prune_due:
            return False
        if 2.0 * math.sqrt(d2_plan) > 0.95 * self.gm:       # the plan cannot vouch for a prune any more
            return True
        self.e.request_prune()                               # the next force pass walks the outer list and prunes
        self.prune_step = step_n
        self.stats["prunes"] += 1
        return False

    def run(self, first_step, n_steps, dt, remove_cm_every=1):
        if self.engine_loop and n_steps > 0:
            # the steps, the ghost exchange and the prune decisions all happen inside the engine; the host comes back only when
            # ownership has to be re-planned (≈ every 100 steps at 1M atoms) and at the end of the run
            s, last = first_step, first_step + n_steps
            while s < last:
                done, reason = self.e.domain_run(s, last - s, dt, remove_cm_every, self.cm_buf, self._counters)
                s += done
                self.stats["exchange_calls"] += done
                if bool(remove_cm_every) and s % remove_cm_every == 0:
                    self._all_reduce(self.cm_buf)
                    self.e.remove_cm(self.cm_buf)          # applied by the next first kick (or any read of the state)
                if reason == 1:
                    self.migrate(s)
            self.stats["plan_checks"], self.stats["prunes"] = int(self._counters[0]), int(self._counters[1])
            self._sync_from_engine()
            return
        if not self.fused:
            for s in range(first_step + 1, first_step + n_steps + 1):
                self.step(s, dt, remove_cm_every)
            return
        # Fused stepping: per step the ghost exchange (with the step-before's Σ m v on board), the interior blocks meanwhile, then ONE
        # engine call — unpack, boundary blocks, second kick, first kick + drift of the next step, pack.  The collective prune /
        # re-plan decision is taken at the rebuild cadence BEFORE the step, from the displacements of the owned atoms (every ghost is
        # somebody's owned atom, and the MAX runs over all ranks).  Only a step that is followed by a re-plan, and the last one, stop
        # behind their second kick (state of step s in place; Σ m v all-reduced there); the next step then starts with halo_start.
        last = first_step + n_steps
        exchange = self.world > 1 and bool(self.g.dirs)
        if n_steps > 0:
            self.e.halo_start(dt)
        for s in range(first_step + 1, last + 1):
            cm = bool(remove_cm_every) and s % remove_cm_every == 0
            replan = (s % self.every == 0 or s == self.next_check) and self._replan_due(s)
            stop = replan or s == last
            work = self._a2a_async(self.f_recv.view(-1), self.f_send.view(-1), self._frc3, self._fsc3) if (exchange and self.overlap) else None
            if self.overlap and self.e.halo_interior(s):
                self.stats["interior_passes"] += 1
            if exchange:
                if work is not None:
                    work.wait()
                else:
                    self._a2a(self.f_recv.view(-1), self.f_send.view(-1), self._frc3, self._fsc3)
                self.stats["exchange_calls"] += 1
            self.e.halo_mid(s, dt, cm, stop, self.cm_buf)
            if stop:
                if cm:
                    self._all_reduce(self.cm_buf)
                    self.e.remove_cm(self.cm_buf)          # applied by the next first kick (or any read of the state)
                if replan:
                    self.migrate(s)
                if s != last:
                    self.e.halo_start(dt)

    # -- migration at the rebuild cadence -------------------------------------------------------------------------------
    def _sync_from_engine(self):
        """the engine re-planned on its own since the host last looked: atom counts, global ids and parameters as they are now"""
        if not self.device_replan:
            return
        n_owned, n_ghost, n_replans, n_arrived = self.e.domain_info()[:4]
        if n_replans == self._dev_replans_seen:
            return
        self.stats["plans"] += n_replans - self._dev_replans_seen
        self._dev_replans_seen = n_replans
        self.stats["migrated"] += n_arrived - self._dev_arrived_seen      # (on top of what earlier host migrations counted: the engine's figure is a running total of its own)
        self._dev_arrived_seen = n_arrived
        self.n_owned, self.n_ghost = n_owned, n_ghost
        self.stats["ghost_atoms"] = n_ghost
        self.gid = torch.empty(n_owned, dtype=torch.int64, device=self.device)
        self.par = torch.empty((n_owned, 4), dtype=self.tdtype, device=self.device)
        self.e.domain_export(self.gid, self.par)
        self.v = torch.empty((n_owned, 3), dtype=self.tdtype, device=self.device)

    def pull(self):
        self._sync_from_engine()
        x_all = torch.empty((self.n_owned + self.n_ghost, 3), dtype=self.tdtype, device=self.device)
        self.e.get_state(x_all, self.v)
        self.x = x_all[: self.n_owned].contiguous()

    def migrate(self, step):
        """Atoms that left the brick go to their new owner; everybody then makes a new ghost plan.  Only the leavers travel (a few
        per thousand per re-plan): the atoms that stay keep their local order, arrivals are appended in (source rank, sender order)."""
        self.pull()
        x = self.x - torch.floor(self.x / self.boxt) * self.boxt      # wrap the cut (open) axes too
        if self.world == 1:
            self.x = x
            self._plan_and_load(step)
            if self.device_replan:
                self._hand_over_domain()
            return
        dest = self.g.owner_of(x)
        li = torch.nonzero(dest != self.rank).squeeze(1)               # leavers (one host sync; the list is short)
        ld = dest[li]
        o = torch.argsort(ld, stable=True)
        li, ld = li[o], ld[o]
        send_counts = torch.bincount(ld, minlength=self.world)
        recv_counts = torch.empty_like(send_counts)
        self._a2a(recv_counts, send_counts)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        payload = torch.cat([x[li], self.v[li], self.par[li]], dim=1).contiguous()     # 10 reals per leaver
        gid = self.gid[li].contiguous()
        n_new = int(sum(rc))
        rp = torch.empty((n_new, 10), dtype=self.tdtype, device=self.device)
        rg = torch.empty(n_new, dtype=torch.int64, device=self.device)
        self._a2a(rp.view(-1), payload.view(-1), [c * 10 for c in rc], [c * 10 for c in sc])
        self._a2a(rg, gid, rc, sc)
        self.stats["migrated"] += int(li.numel())
        if li.numel() or n_new:
            keep = torch.ones(x.shape[0], dtype=torch.bool, device=self.device)
            keep[li] = False
            self.x = torch.cat([x[keep], rp[:, 0:3]], dim=0).contiguous()
            self.v = torch.cat([self.v[keep], rp[:, 3:6]], dim=0).contiguous()
            self.par = torch.cat([self.par[keep], rp[:, 6:10]], dim=0).contiguous()
            self.gid = torch.cat([self.gid[keep], rg], dim=0).contiguous()
        else:
            self.x = x
        self._plan_and_load(step)
        if self.device_replan:
            self._hand_over_domain()

    # -- gather the whole system on every rank (tests / final state) ------------------------------------------------------
    def gather_global(self, n_total):
        self.pull()
        counts = [int(c.item()) for c in self._all_gather(torch.tensor([self.n_owned], dtype=torch.int64, device=self.device))]
        mx = max(counts)
        pack = torch.zeros((mx, 7), dtype=torch.float64, device=self.device)
        pack[: self.n_owned, 0] = self.gid.to(torch.float64)
        pack[: self.n_owned, 1:4] = self.x.to(torch.float64)
        pack[: self.n_owned, 4:7] = self.v.to(torch.float64)
        out = self._all_gather(pack)
        xs = np.zeros((n_total, 3)); vs = np.zeros((n_total, 3))
        for r, t in enumerate(out):
            t = t[: counts[r]].cpu().numpy()
            ids = t[:, 0].astype(np.int64)
            xs[ids], vs[ids] = t[:, 1:4], t[:, 4:7]
        return xs, vs


def make_interactions(case, dtype):
    """mhip_interactions of a tests.systems.Case-like description (lj / coul dicts) without building a System."""
    it = _lib.Interactions()
    d = case.inter_dict(dtype)
    it.lj_weight_special = 1.0; it.coul_weight_special = 1.0; it.coul_ke = 138.93545764; it.rf_dielectric = 1.0; it.ewald_approx_erfc = 1
    for k, v in d.items():
        setattr(it, k, v)
    return it


def bench_distributed(m, case, dtype, dt, args, rank, local_rank, world):
    """bench.py body for N > 1: strong scaling of one box over `world` GPUs.  Returns (ms_per_step, stats, extra) on rank 0."""
    if case.excluded is not None or case.bonds is not None:
        raise SystemExit("the multi-GPU path covers exception-free systems (LJ fluids); 6mrr runs as replicas only (SURVEY §8(e))")
    import os
    # test hooks: several ranks on ONE GPU (the single-GPU box) with gloo + host staging; production = one GPU per rank, RCCL
    backend = os.environ.get("MOLLYHIP_DIST_BACKEND", "nccl")
    if "MOLLYHIP_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["MOLLYHIP_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world == 1:   # the N = 1 box driven through this loop (MOLLYHIP_FORCE_DOMAIN): a single-rank group needs no launcher
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511")):
            os.environ.setdefault(k, v)
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    device = torch.device("cuda", local_rank)
    tdtype = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
    grid = choose_grid(world, case.box)
    gm = float(os.environ.get("MOLLYHIP_GHOST_MARGIN_PM", "200")) * 1e-3
    brick_min = min([b / g for b, g in zip(case.box, grid) if g > 1], default=math.inf)
    if case.r_list + gm > brick_min:
        gm = 0.0                                            # bricks too thin for a margin: re-plan at every rebuild step
    bg = BrickGrid(case.box, grid, rank, case.r_list + gm)
    box, origin, periodic = bg.engine_box(pad=0.3)
    vol_frac = np.prod([b / L for b, L in zip(box, case.box)])
    capacity = int(case.n * min(1.0, vol_frac) * 1.25) + 4096
    rc_max = max([c[1] for c in ([case.lj.get("cutoff", ("none", 0.0))] if case.lj else []) if len(c) > 1] + ([case.coul.get("rc", 0.0)] if case.coul else []) + [0.0])
    equil = getattr(args, "equil", None)
    equil = equil if equil is not None else (2000 if getattr(args, "workload", "lj1m").startswith("lj") else 0)
    # Three forms of the step loop, fastest first.  No rank pair of this repository has met over xGMI before the driver's scaling run, so a form that fails
    # during the UNTIMED part — a wait for a peer that times out, a HIP error — does not end the job: every rank learns of it (MIN all-reduce of a flag), all contexts are
    # dropped and the next form starts from the initial state.  The record names the form that ran.  (A fault that kills a process cannot be caught, of course.)
    forms = [("fused", {}), ("separate launches", {"MOLLYHIP_FUSE_STEP": "0"}), ("host loop", {"MOLLYHIP_ENGINE_LOOP": "0"})]
    if os.environ.get("MOLLYHIP_ENGINE_LOOP", "1") == "0":
        forms = forms[2:]
    elif os.environ.get("MOLLYHIP_FUSE_STEP", "1") == "0":
        forms = forms[1:]
    eng = run = None
    for k, (form, env) in enumerate(forms):
        os.environ.update(env)
        ok, why = 1, ""
        try:
            eng = HipDomainEngine(make_interactions(case, dtype), dtype, capacity, box, origin, periodic, case.r_list, case.rebuild_every, local_rank, ghost_margin=gm)
            run = DomainRun(bg, eng, tdtype, device, case.rebuild_every, ghost_margin=gm, skin=case.r_list - rc_max)
            run.setup_from_global(case.coords, case.velocities, np.zeros(case.n) if case.charge is None else case.charge, case.sigma, case.eps, case.mass)
            # (tests: the last rank gives up here, behind the collective set-up — its peers then run into the bounded waits of the engine loop, as they would if it had died)
            if form in (getattr(args, "fail_forms", "") or "").split(",") and rank == world - 1:
                raise RuntimeError("injected failure (bench.py --fail-forms, tests)")
            # untimed: equilibrate the jittered lattice first (SURVEY §8(d) cfg 4), then the warm-up
            run.run(0, equil + args.warmup, dt)
            torch.cuda.synchronize()
        except (_lib.MollyHipError, RuntimeError) as e:
            ok, why = 0, str(e)
            print(f"[bench rank {rank}] step loop form '{form}' failed in the untimed part: {why[:300]}", file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            break
        try:
            if eng is not None:
                eng.close()
        except Exception:
            pass
        eng = run = None
        if k == len(forms) - 1:
            raise SystemExit(f"every form of the multi-GPU step loop failed (last on this rank: {why[:300] or 'a peer failed'})")
    loop_form = form
    first = equil + args.warmup
    # the K timed steps are taken as scheduled; windows shorter than a pair-list cycle are repeated back to back until they cover
    # 100 steps and the headline is their mean (bench.py's single-domain leg does the same)
    n_win = 1 if args.steps >= 100 else -(-100 // args.steps)
    win = []
    for _ in range(n_win):
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        run.run(first, args.steps, dt)
        torch.cuda.synchronize(); dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        win.append(float(el.item()) * 1e3 / args.steps)
        first += args.steps
    ms_per_step = float(np.mean(win))
    window = "as scheduled" if n_win == 1 else f"as scheduled: mean of {n_win} consecutive windows of {args.steps} steps (one whole pair-list cycle)"
    eng.set_profiling(True)
    run.run(first, args.profile_steps, dt)
    torch.cuda.synchronize()
    st = eng.stats()
    eng.set_profiling(False)
    # whole-job figures for the roofline block: pairs and bytes summed over the ranks
    tot = torch.tensor([st["n_pairs_full"], st["force_pass_bytes"], st["algorithmic_bytes_step"], run.n_ghost, run.n_owned], dtype=torch.float64,
                       device=device if backend == "nccl" else "cpu")
    per_rank = [torch.zeros_like(tot) for _ in range(world)]
    dist.all_gather(per_rank, tot)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()
    if rank != 0:
        return None
    agg = torch.stack(per_rank).sum(0).tolist()
    st["n_pairs_full"] = int(agg[0])
    transport = (("peer stores into IPC-mapped receive regions, step loop inside the engine (mhip_domain_run" + (", one launch per plain step)" if loop_form == "fused" else ", separate launches per step)"))
                 if run.engine_loop else "all_to_all_single (RCCL) from the host loop")
    extra = {"parallelism": f"spatial bricks {grid[0]}x{grid[1]}x{grid[2]}, full-shell ghost coordinates via {transport}, "
                            f"{int(agg[3] / world)} ghosts / {int(agg[4] / world)} owned atoms per GPU, ghost margin {gm:.2f} nm "
                            f"({run.stats['plans']} ghost plans, {run.stats['prunes']} prunes in {run.stats['plan_checks']} checks)",
             "per_gpu_force_pass_bytes": st["force_pass_bytes"], "ghost_fraction": agg[3] / max(agg[4], 1), "timed_window": window,
             "window_ms_per_step": {"n": n_win, "mean": ms_per_step, "min": float(min(win)), "max": float(max(win))}}
    return ms_per_step, st, extra
