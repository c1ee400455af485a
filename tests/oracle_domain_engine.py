"""CPU stand-in for molly_jl_amd.domain.HipDomainEngine, backed by the oracle — TEST INFRASTRUCTURE ONLY.  It lets the
world_size > 1 host logic (ownership, ghost plan, all_to_all exchange, migration) run under gloo without a GPU."""
import math
import os

import numpy as np
import torch

from oracle import pyoracle as orc


class OracleDomainEngine:
    def __init__(self, inter_dict, box, periodic, r_list, dtype=np.float64, ghost_margin=0.0, skin=0.2, every=10):
        self.inter, self.r_list, self.dtype = inter_dict, r_list, dtype
        self.ghost_margin, self.n_prunes = ghost_margin, 0
        self.skin, self.every, self.prune_step = skin, every, 0     # what plan_decide schedules with (the HIP engine: its inner skin)
        # MOLLYHIP_TEST_EXTRA_CHECK=k: a cadence decision that keeps the lists asks for one more look k steps later, as the HIP engine
        # does when the inner list is good for k < rebuild_every steps only (mhip_plan_decide's check_in); every decision is logged
        self.extra_check = int(os.environ.get("MOLLYHIP_TEST_EXTRA_CHECK", "0")); self.decide_steps = []
        self.box = np.array([b if p else math.inf for b, p in zip(box, periodic)])   # open axes: no minimum image
        self.periodic = periodic

    def set_local(self, n_owned, n_ghost, q, sigma, eps, mass, x_all, v_owned):
        self.n_owned, self.n_ghost = n_owned, n_ghost
        f = lambda t: t.detach().cpu().numpy().astype(np.float64).copy()
        self.q, self.sigma, self.eps, self.mass = f(q), f(sigma), f(eps), f(mass)
        self.x, self.v = f(x_all), f(v_owned)
        self.x_plan = self.x.copy(); self.x_prune = self.x.copy()
        self.f = None

    def _forces(self):
        o = orc.OracleSystem(self.x, self.box, self.inter, dtype=np.float64, charge=self.q, sigma=self.sigma, eps=self.eps,
                             mass=self.mass, r_list=self.r_list)
        nl = o.neighbors("brute")
        self.f = o.forces(nl)[: self.n_owned]

    def gather(self, idx, shift, out):
        out.copy_(torch.from_numpy(self.x[idx.numpy().astype(np.int64)] + shift.numpy().astype(np.float64)).to(out.dtype))

    def scatter(self, first, n, buf):
        self.x[first:first + n] = buf.numpy().astype(np.float64)

    def vv_init(self, step):
        self._forces()

    def stage1(self, dt):
        n, m = self.n_owned, self.mass[: self.n_owned, None]
        self.v += self.f / m * (dt / 2)
        self.x[:n] += self.v * dt
        for d in range(3):
            if self.periodic[d]:
                L = self.box[d]
                self.x[:n, d] -= np.floor(self.x[:n, d] / L) * L

    def stage2(self, step, dt):
        self._forces()
        self.v += self.f / self.mass[: self.n_owned, None] * (dt / 2)

    # the fused per-step entry points of HipDomainEngine
    def halo_begin(self, dt, idx, shift, out):
        self.stage1(dt)
        if idx.numel():
            self.gather(idx, shift, out)

    def halo_interior(self, step):
        return False                       # the stand-in computes its forces in one piece

    def halo_end(self, step, dt, first, n, buf, cm_out4):
        if n:
            self.scatter(first, n, buf)
        self.stage2(step, dt)
        if cm_out4 is not None:
            self.cm_momentum(cm_out4)

    # -- the fused step (mhip_set_halo_plan / mhip_vv_halo_start / mhip_vv_halo_mid): same message layout, same delayed removal of
    #    the centre-of-mass motion (v −= v_cm and x −= v_cm·dt one integrator pass late) as the HIP engine
    def set_halo_plan(self, first_ghost, recv, recv_dst, n_cm_peers, cm_rows, send_idx, send_shift, send, send_cm_pos):
        self.hp = dict(first=first_ghost, recv=recv, dst=recv_dst.numpy().astype(np.int64), n_peers=n_cm_peers, cm_rows=cm_rows,
                       idx=send_idx.numpy().astype(np.int64), shift=send_shift.numpy().astype(np.float64), send=send,
                       cm_pos=send_cm_pos.numpy().astype(np.int64))
        self.cm_all = np.zeros((1 + n_cm_peers, 4)); self.halo_cm_in = False; self.cm_own = np.zeros(4)

    def _wrap_owned(self):
        n = self.n_owned
        for d in range(3):
            if self.periodic[d]:
                L = self.box[d]
                self.x[:n, d] -= np.floor(self.x[:n, d] / L) * L

    def _pack(self, with_cm):
        h = self.hp
        np_dt = h["send"].numpy().dtype
        out = np.zeros((h["idx"].shape[0], 3), dtype=np_dt)
        m = h["idx"] >= 0
        out[m] = (self.x[h["idx"][m]] + h["shift"][m]).astype(np_dt)
        tot = self.cm_own if with_cm else np.zeros(4)
        words = np.zeros(3 * max(h["cm_rows"], 1), dtype=np_dt)
        w = tot.astype(np.float64).view(np_dt)
        words[: w.shape[0]] = w
        for q, k in enumerate(h["cm_pos"]):
            r = q % h["cm_rows"]
            out[k] = words[3 * r: 3 * r + 3]
        if with_cm:
            self.cm_all[0] = tot
        h["send"].copy_(torch.from_numpy(out))

    def halo_start(self, dt):
        self.stage1(dt)
        self._pack(False)
        self.halo_cm_in = False

    def halo_mid(self, step, dt, cm, stop, cm_parts):
        h = self.hp
        buf, dst = h["recv"].numpy(), h["dst"]
        g = dst >= 0
        self.x[h["first"] + dst[g]] = buf[g].astype(np.float64)
        if h["cm_rows"] > 0:
            for k in np.nonzero(~g)[0]:
                code = -1 - dst[k]; peer, r = divmod(code, h["cm_rows"])
                w = self.cm_all[1 + peer].view(buf.dtype)                 # the peer's four doubles as words of the message type
                hi = min(3 * r + 3, w.shape[0])
                w[3 * r: hi] = buf[k, : hi - 3 * r]
        self._forces()
        n, m = self.n_owned, self.mass[: self.n_owned, None]
        corrected = False
        if self.halo_cm_in:
            t = self.cm_all.sum(axis=0); c = t[:3] / t[3]
            self.v -= c; self.x[:n] -= c * dt; corrected = True
        kick = self.f / m * (dt / 2)
        self.v += kick
        if cm:
            self.cm_own = np.concatenate([(self.v * m).sum(axis=0), [float(m.sum())]])
        if not stop:
            self.v += kick
            self.x[:n] += self.v * dt
            self._wrap_owned()
            self._pack(cm)
            self.halo_cm_in = bool(cm)
        else:
            if corrected:
                self._wrap_owned()
            if cm:
                cm_parts.zero_(); cm_parts[:4] = torch.from_numpy(self.cm_own)
            self.halo_cm_in = False

    def plan_state(self, out3):
        out3[0] = self._disp2(self.x_plan); out3[1] = self._disp2(self.x_prune); out3[2] = float((self.v * self.v).sum(axis=1).max())

    def plan_decide(self, step, red):
        d2_plan, d2_prune = red[0], red[1]
        self.decide_steps.append(int(step))
        k = max(1, (step - self.prune_step) // self.every)
        if not (math.isinf(d2_prune) or 2.0 * math.sqrt(d2_prune) * (k + 1) / k > 0.98 * self.skin):
            return 0, (self.extra_check if step % self.every == 0 else 0)
        if 2.0 * math.sqrt(d2_plan) > 0.95 * self.ghost_margin:
            return 2, 0
        self.request_prune(); self.prune_step = step
        return 1, 0

    def _disp2(self, ref):
        d = self.x - ref
        for k in range(3):
            if self.periodic[k]:
                d[:, k] -= np.round(d[:, k] / self.box[k]) * self.box[k]
        return float((d * d).sum(axis=1).max())

    def plan_disp2(self, out2):
        out2[0] = self._disp2(self.x_plan); out2[1] = self._disp2(self.x_prune)

    def request_prune(self):
        # this stand-in searches its neighbours at every force call; what it checks is that a prune is only asked for while the
        # ghost shell still covers r_list around every owned atom — i.e. while the plan is valid
        assert 2.0 * np.sqrt(self._disp2(self.x_plan)) <= self.ghost_margin + 1e-12
        self.x_prune = self.x.copy()
        self.n_prunes += 1

    def get_state(self, x_all, v_owned):
        x_all.copy_(torch.from_numpy(self.x).to(x_all.dtype)); v_owned.copy_(torch.from_numpy(self.v).to(v_owned.dtype))

    def cm_momentum(self, out):            # out: 4·k doubles, k partials; this stand-in fills the first one
        m = self.mass[: self.n_owned]
        out.zero_()
        out[:3] = torch.from_numpy((self.v * m[:, None]).sum(axis=0)); out[3] = float(m.sum())

    def remove_cm(self, total):
        t = total.view(-1, 4).sum(dim=0)
        self.v -= (t[:3] / t[3]).numpy()
