"""Wall time of one re-plan (migration + ghost plan + outer search) of the multi-GPU host loop on a single-rank communicator."""
import os, sys, time, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import molly_loader; m = molly_loader.load()
from molly_jl_amd import domain
from tests import systems as S
torch.cuda.set_device(0); dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
case = S.lj_fluid(int(sys.argv[1]) if len(sys.argv) > 1 else 100, seed=4, dtype=np.float32)
dev = torch.device("cuda", 0)
bg = domain.BrickGrid(case.box, (1, 1, 1), 0, case.r_list + 0.2)
box, origin, periodic = bg.engine_box(pad=0.3)
eng = domain.HipDomainEngine(domain.make_interactions(case, np.float32), np.float32, case.n + 4096, box, origin, periodic, case.r_list, 10, 0, ghost_margin=0.2)
run = domain.DomainRun(bg, eng, torch.float32, dev, 10, ghost_margin=0.2, skin=0.2)
run.setup_from_global(case.coords, case.velocities, np.zeros(case.n), case.sigma, case.eps, case.mass)
run.run(0, 50, 0.002)
for name, fn in (("pull", run.pull), ("migrate (all)", lambda: run.migrate(50))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) / 5 * 1e3, "ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); run.migrate(50); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
dist.destroy_process_group()
