"""Bricks on the one GPU against the single-domain engine, any size:  python tools/micro/brick_check.py WORLD N_SIDE GM N_STEPS [SHIFT] [DEV_REPLAN]
prints the mean / worst coordinate deviation after N_STEPS and each rank's plan counters (diagnostic for tests/test_gpu_domain.py's benchmark-size cases)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch.multiprocessing as mp

world, n_side, gm, n_steps = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
shift = float(sys.argv[5]) if len(sys.argv) > 5 else 0.17
if len(sys.argv) > 6:
    os.environ["MOLLYHIP_DEVICE_REPLAN"] = sys.argv[6]

if __name__ == "__main__":
    from tests import test_gpu_domain as T
    import molly_loader
    pkg = molly_loader.load()
    out = tempfile.mkdtemp()
    try:
        mp.spawn(T._big_worker, args=(world, T._free_port(), n_side, n_steps, out, gm, False, shift), nprocs=world, join=True)
    except Exception as e:
        print(f"world {world} n_side {n_side} gm {gm} steps {n_steps}: FAILED {str(e).strip().splitlines()[-1][:300]}")
        sys.exit(0)
    res = [np.load(os.path.join(out, f"big{r}.npz")) for r in range(world)]
    case = T._case(n_side, np.float32, shift)
    s = case.system(pkg, np.float32)
    pkg.simulate(s, pkg.VelocityVerlet(dt=0.002), n_steps)
    d = res[0]["x"] - s.coords.astype(np.float64)
    d -= np.round(d / case.box) * case.box
    print(f"world {world} n_side {n_side} ({case.n} atoms) gm {gm} steps {n_steps} dev_replan {os.environ.get('MOLLYHIP_DEVICE_REPLAN', '1')}: mean |dx| {np.abs(d).mean():.3e} max {np.abs(d).max():.3e} nan {int(np.isnan(res[0]['x']).sum())}"
          f" | per rank (ghosts, dev re-plans, plans, migrated, outer, prunes, block): {[(int(r['ghosts']), int(r['dev_replans']), int(r['plans']), int(r['migrated']), int(r['outer']), int(r['prunes']), int(r['block_atoms'])) for r in res]}")
