"""Host time of the multi-GPU host loop on ONE rank (1M-atom fluid): seconds spent inside each engine / collective call per step,
against the wall time per step.  python tools/micro/domain_host_time.py [steps]"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import molly_loader; molly_loader.load()
from molly_jl_amd import domain
from tests import systems as S

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
case = S.lj_fluid(100, dtype=np.float32, rebuild_every=10)
gm = 0.2
bg = domain.BrickGrid(case.box, (1, 1, 1), 0, case.r_list + gm)
box, origin, periodic = bg.engine_box(pad=0.3)
eng = domain.HipDomainEngine(domain.make_interactions(case, np.float32), np.float32, case.n + 4096, box, origin, periodic, case.r_list, 10, 0, ghost_margin=gm)
run = domain.DomainRun(bg, eng, torch.float32, torch.device("cuda", 0), 10, ghost_margin=gm, skin=0.2)
run.setup_from_global(case.coords, case.velocities, np.zeros(case.n), case.sigma, case.eps, case.mass)
run.run(0, 2300, 0.002)
torch.cuda.synchronize()
acc = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
for n in ("halo_mid", "halo_interior", "halo_start", "plan_state", "plan_decide", "remove_cm"):
    wrap(eng, n)
wrap(run, "replan_if_due")
t0 = time.perf_counter()
run.run(2300, steps, 0.002)
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("per step: wall %.1f us, host loop returned after %.1f us" % (wall / steps * 1e6, host / steps * 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("   %-16s %8.1f us / step" % (k, v / steps * 1e6))
print(run.stats, eng.stats()["n_outer_builds"], eng.stats()["n_filter_passes"])
eng.close(); dist.destroy_process_group()
