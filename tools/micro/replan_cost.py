"""What a re-plan of the multi-GPU loop costs the host.

  python tools/micro/replan_cost.py                         one rank, 1M-atom fluid, the HOST planner (DomainRun.migrate) by stage — rounds 2-4
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/micro/replan_cost.py --device [lj256k]
                                                            the planner INSIDE the engine (mhip_set_domain, replan.h), ranks sharing the one GPU over gloo:
                                                            host time per re-plan = launches + the one read-back, and the search behind it, from mhip_domain_info
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import molly_loader; molly_loader.load()
from molly_jl_amd import domain
from tests import systems as S

device_mode = "--device" in sys.argv
n_side = 64 if "lj256k" in sys.argv else 100
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
if world > 1 or device_mode:
    dist.init_process_group("gloo")
else:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
case = S.lj_fluid(n_side, dtype=np.float32, rebuild_every=10)
gm = 0.05 if device_mode else 0.2          # (a thin margin: the plan goes stale every ≈ 20 steps, so a short run holds many re-plans)
grid = domain.choose_grid(world, case.box)
bg = domain.BrickGrid(case.box, grid, rank, case.r_list + gm)
box, origin, periodic = bg.engine_box(pad=0.3)
vol_frac = np.prod([b / L for b, L in zip(box, case.box)])
eng = domain.HipDomainEngine(domain.make_interactions(case, np.float32), np.float32, int(case.n * min(1.0, vol_frac) * 1.25) + 4096, box, origin, periodic, case.r_list, 10, 0, ghost_margin=gm)
run = domain.DomainRun(bg, eng, torch.float32, torch.device("cuda", 0), 10, ghost_margin=gm, skin=0.2)
run.setup_from_global(case.coords, case.velocities, np.zeros(case.n), case.sigma, case.eps, case.mass)
run.run(0, 500, 0.002)
torch.cuda.synchronize()
if device_mode:
    i0 = eng.domain_info()
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    run.run(500, 400, 0.002)
    torch.cuda.synchronize(); dist.barrier(); el = time.perf_counter() - t0
    i1 = eng.domain_info()
    n = max(i1[2] - i0[2], 1)
    print("rank %d/%d (%d owned, %d ghosts): %d re-plans in 400 steps (%.4f ms/step) | host per re-plan: planning %.3f ms, search %.3f ms | %d atoms arrived"
          % (rank, world, i1[0], i1[1], i1[2] - i0[2], el * 1e3 / 400, (i1[4] - i0[4]) * 1e-3 / n, (i1[5] - i0[5]) * 1e-3 / n, i1[3] - i0[3]), flush=True)
    dist.barrier()
    eng.close(); dist.destroy_process_group()
    sys.exit(0)
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
