"""CPU stand-in for molly_jl_amd.domain.HipDomainEngine, backed by the oracle — TEST INFRASTRUCTURE ONLY.  It lets the
world_size > 1 host logic (ownership, ghost plan, all_to_all exchange, migration) run under gloo without a GPU."""
import math

import numpy as np
import torch

from oracle import pyoracle as orc


class OracleDomainEngine:
    def __init__(self, inter_dict, box, periodic, r_list, dtype=np.float64, ghost_margin=0.0):
        self.inter, self.r_list, self.dtype = inter_dict, r_list, dtype
        self.ghost_margin, self.n_prunes = ghost_margin, 0
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
