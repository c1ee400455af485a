"""Where a re-plan of the multi-GPU host loop spends its time (one rank, 1M-atom fluid): python tools/micro/replan_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import molly_loader; molly_loader.load()
from molly_jl_amd import domain
from tests import systems as S

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
case = S.lj_fluid(100, dtype=np.float32, rebuild_every=10)
gm = 0.2
bg = domain.BrickGrid(case.box, (1, 1, 1), 0, case.r_list + gm)
box, origin, periodic = bg.engine_box(pad=0.3)
eng = domain.HipDomainEngine(domain.make_interactions(case, np.float32), np.float32, case.n + 4096, box, origin, periodic, case.r_list, 10, 0, ghost_margin=gm)
run = domain.DomainRun(bg, eng, torch.float32, torch.device("cuda", 0), 10, ghost_margin=gm, skin=0.2)
run.setup_from_global(case.coords, case.velocities, np.zeros(case.n), case.sigma, case.eps, case.mass)
run.run(0, 500, 0.002)
torch.cuda.synchronize()
acc = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
for n in ("set_local", "vv_init", "set_halo_plan"):
    wrap(eng, n)
for n in ("migrate", "pull", "_plan_and_load", "_halo_layout"):
    wrap(run, n)
reps = 5
for k in range(reps):
    run.migrate(500)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("   %-16s %8.3f ms / re-plan" % (k, v / reps * 1e3))
eng.close(); dist.destroy_process_group()
