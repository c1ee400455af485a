"""Forces of a brick decomposition at step 0 against the single-domain engine:  python tools/micro/brick_forces.py WORLD N_SIDE GM [ENV=VALUE ...]
Each rank sets up its sub-domain (ghosts from the host planner), evaluates the pair forces of its owned atoms through mhip_forces and saves them with the
atoms' global ids; the parent evaluates the same system on one context and reports which atoms differ (diagnostic)."""
import os, sys, tempfile, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

world, n_side, gm = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
for kv in sys.argv[4:]:
    k, v = kv.split("="); os.environ[k] = v


def worker(rank, world, port, n_side, gm, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C
    import molly_loader; molly_loader.load()
    from molly_jl_amd import domain, _lib
    from tests import test_gpu_domain as T
    torch.cuda.set_device(0); dev = torch.device("cuda", 0)
    case = T._case(n_side, np.float32, 0.17)
    grid = domain.choose_grid(world, case.box)
    bg = domain.BrickGrid(case.box, grid, rank, case.r_list + gm)
    box, origin, periodic = bg.engine_box(pad=0.3)
    vol_frac = np.prod([b / L for b, L in zip(box, case.box)])
    eng = domain.HipDomainEngine(domain.make_interactions(case, np.float32), np.float32, int(case.n * min(1.0, vol_frac) * 1.25) + 4096, box, origin, periodic, case.r_list, 10, 0, ghost_margin=gm)
    run = domain.DomainRun(bg, eng, torch.float32, dev, 10, ghost_margin=gm, skin=0.2)
    run.setup_from_global(case.coords, case.velocities, np.zeros(case.n), case.sigma, case.eps, case.mass)
    f = torch.zeros((run.n_owned, 3), dtype=torch.float32, device=dev)
    eng._chk(eng.L.mhip_forces(eng.ctx, 0, 0, C.c_void_p(f.data_ptr()), None, _lib.MEM_DEVICE))
    torch.cuda.synchronize()
    st = eng.stats()
    np.savez(os.path.join(out, f"f{rank}.npz"), gid=run.gid.cpu().numpy(), f=f.cpu().numpy(), x=run.x.cpu().numpy(), block=st["block_atoms"], js=st["j_split"], ghosts=run.n_ghost, tile=st["max_tile_atoms"], seg=st["tile_segments"], minimg=st["minimg_mode"])
    dist.barrier(); eng.close(); dist.destroy_process_group()


if __name__ == "__main__":
    from tests import test_gpu_domain as T
    import molly_loader
    pkg = molly_loader.load()
    out = tempfile.mkdtemp()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(world, port, n_side, gm, out), nprocs=world, join=True)
    case = T._case(n_side, np.float32, 0.17)
    sysm = case.system(pkg, np.float32)
    f_ref = pkg.forces(sysm).astype(np.float64)
    scale = np.linalg.norm(f_ref, axis=1).mean()
    for r in range(world):
        d = np.load(os.path.join(out, f"f{r}.npz"))
        err = np.linalg.norm(d["f"].astype(np.float64) - f_ref[d["gid"]], axis=1)
        bad = err > 1e-3 * scale
        x = d["x"]
        print(f"rank {r}: {len(err)} owned, {int(d['ghosts'])} ghosts, shape {int(d['block'])}x{int(d['js'])}, max tile {int(d['tile'])}, segments {int(d['seg'])}, minimg {int(d['minimg'])} | max err {err.max():.3e} (mean |f| {scale:.3e}), "
              f"{int(bad.sum())} atoms off by more than 1e-3 of the mean force", end="")
        if bad.any():
            lo, hi = x[:, 0].min(), x[:, 0].max()
            near = np.minimum(x[bad, 0] - lo, hi - x[bad, 0])
            idx = np.nonzero(bad)[0]
            print(f" | their distance to the nearest x face: min {near.min():.3f} median {np.median(near):.3f} max {near.max():.3f} nm | local index range {idx.min()} .. {idx.max()}, first ten {idx[:10].tolist()} | nan {int(np.isnan(d['f']).any())}")
        else:
            print()
