#!/usr/bin/env python3
"""Geometry behind the choice of per-atom pair lists over cluster-pair lists ("cluster formulation").

A cluster-pair kernel (c×c atom clusters, Newton's third law, j-data broadcast instead of gathered — the layout of the reference's
32×32 tile kernel, ext/MollyCUDAExt.jl:1595-2045, and of GROMACS' nbnxm) evaluates EVERY atom pair of a listed cluster pair.  How
many pair evaluations that is per atom depends only on the density, the list radius and the cluster shape; this script measures it
on an equilibrated configuration of the benchmark fluid (argon, 21.1 atoms/nm³) for the two cluster constructions that are used in
practice: chunks of a space-filling-curve order and z-sorted chunks of x,y grid columns (GROMACS), with the cluster pair kept only if
at least one of its atom pairs lies within the list radius (the best a cluster list can do).

Runs on the CPU (oracle for the equilibration, SciPy for the searches):  python tools/cluster_fill.py
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import systems as S

RHO = 21.105


def morton(x, res):
    q = np.floor(x / res).astype(np.int64)
    key = np.zeros(len(x), np.int64)
    for b in range(10):
        for d in range(3):
            key |= ((q[:, d] >> b) & 1) << (3 * b + d)
    return key


def clusters_curve(x, L, c):
    order = np.argsort(morton(x, 0.3), kind="stable")
    return [order[k:k + c] for k in range(0, len(x), c)]


def clusters_columns(x, L, c):
    a = (c / RHO) ** (1 / 3)
    ncol = int(np.floor(L / a)); a = L / ncol
    cx = np.minimum((x[:, 0] / a).astype(int), ncol - 1); cy = np.minimum((x[:, 1] / a).astype(int), ncol - 1)
    col = cx * ncol + cy
    order = np.lexsort((x[:, 2], col))
    out = []
    for seg in np.split(order, np.flatnonzero(np.diff(col[order])) + 1):
        out += [seg[k:k + c] for k in range(0, len(seg), c)]
    return out


def evaluate(x, L, clusters, c, R):
    n, nc = len(x), len(clusters)
    pad = np.full((nc, c, 3), 1e9)
    for k, idx in enumerate(clusters):
        p = x[idx]; p = p[0] + (p - p[0] - np.round((p - p[0]) / L) * L)     # unwrap around the first member
        pad[k, :len(idx)] = p
    real = pad[..., 0] < 1e8
    lo = np.where(real[..., None], pad, np.inf).min(1); hi = np.where(real[..., None], pad, -np.inf).max(1)
    ctr, half = (lo + hi) / 2, (hi - lo) / 2
    pairs = cKDTree(np.mod(ctr, L), boxsize=L).query_pairs(R + 2 * np.linalg.norm(half, axis=1).max(), output_type="ndarray")
    i, j = pairs[:, 0], pairs[:, 1]
    d = ctr[j] - ctr[i]; sh = np.round(d / L) * L; d -= sh
    keep = (np.maximum(np.abs(d) - half[i] - half[j], 0) ** 2).sum(1) <= R * R
    i, j, sh = i[keep], j[keep], sh[keep]
    r2 = ((pad[i][:, :, None, :] - (pad[j] - sh[:, None, :])[:, None, :, :]) ** 2).sum(-1)
    listed = (r2 <= R * R).any(axis=(1, 2)).sum()
    evals = listed * c * c + nc * c * (c - 1) // 2                       # cluster pairs + the pairs inside every cluster
    return evals / n, listed / n, float(np.linalg.norm(hi - lo, axis=1).mean())


def j_groups(x, L, tree, R):
    """The space between one partner per list entry and 4 x 4 clusters (round-2 review, item 1): an entry names an ALIGNED GROUP of g
    consecutive tile slots and every member of a listed group is evaluated (one 64-bit LDS read per component would return two
    partners).  The tile order is what the engine's tile has: cell-major on the search grid (cell 0.709 nm here), optionally with a
    Morton order of sub-cells inside a cell.  Best case for g = 2: greedy nearest-neighbour matching, pairs as compact as the fluid
    allows."""
    n = len(x)
    nb = tree.query_ball_point(x, R)
    single = np.mean([len(v) - 1 for v in nb])
    print(f"1 x g j-groups at list radius {R} nm (single entries per atom: {single:.1f}):")
    nc = int(np.floor(L / 0.709)); cs = L / nc
    c = np.minimum((x / cs).astype(int), nc - 1)
    key = (c[:, 2] * nc + c[:, 1]) * nc + c[:, 0]

    def count(grp):
        return sum(len({grp[j] for j in nb[i] if j != i}) for i in range(n)) / n

    orders = [("cell-major", np.lexsort((np.arange(n), key)))]
    for sub in (2, 3):
        f = np.minimum(((x / cs - c) * sub).astype(int), sub - 1)
        orders.append((f"cell-major + {sub}^3 sub-cells", np.lexsort((np.arange(n), (f[:, 2] * sub + f[:, 1]) * sub + f[:, 0], key))))
    orders.append(("cell-major, z-sorted in the cell", np.lexsort((x[:, 2], key))))
    for label, order in orders:
        rank = np.empty(n, int); rank[order] = np.arange(n)
        for g in (2, 4):
            e = count(rank // g)
            print(f"  {label:34s} 1x{g}: {e:6.1f} entries per atom, {e * g:6.1f} evaluations = {e * g / single:.2f} x the single-entry list")
    d, j = tree.query(x, k=6)
    mate = -np.ones(n, int)
    for dist, a, b in sorted((d[i, k], i, j[i, k]) for i in range(n) for k in range(1, 6)):
        if mate[a] < 0 and mate[b] < 0:
            mate[a] = b; mate[b] = a
    left = np.flatnonzero(mate < 0)
    for a, b in zip(left[0::2], left[1::2]):
        mate[a] = b; mate[b] = a
    e = count(np.minimum(np.arange(n), mate))
    print(f"  {'greedy nearest-neighbour pairs':34s} 1x2: {e:6.1f} entries per atom, {2 * e:6.1f} evaluations = {2 * e / single:.2f} x the single-entry list")


def main():
    case = S.lj_fluid(20, dtype=np.float64)              # 8000 atoms, box 7.24 nm
    o = case.oracle(np.float64)
    o.vv_run(400, 0.002, remove_cm_every=1, nthreads=8)   # melt the jittered lattice
    L = case.box[0]
    x = o.coords - np.floor(o.coords / L) * L
    n = len(x)
    tree = cKDTree(x, boxsize=L)
    half_rc = len(tree.query_pairs(1.0)) / n
    print(f"{n} atoms, box {L:.3f} nm; unique pairs per atom within rc = 1.0 nm: {half_rc:.1f}")
    print("per-atom full list (this engine): DIRECTED evaluations per atom = 2 x unique pairs within the list radius")
    for R in (1.1, 1.2):
        print(f"  R {R}: {2 * len(tree.query_pairs(R)) / n:.0f} per atom (+ ~14 % padding to the wave's longest lane)")
    print("cluster pairs (Newton's third law: every listed atom pair evaluated ONCE):")
    for name, make in (("curve chunks", clusters_curve), ("x,y columns", clusters_columns)):
        for c in (4, 8):
            cl = make(x, L, c)
            for R in (1.1, 1.2):
                ev, cp, diag = evaluate(x, L, cl, c, R)
                print(f"  {name:12s} {c}x{c} R {R}: {ev:5.0f} evaluations per atom, {cp:4.1f} cluster pairs per atom, bounding-box diagonal {diag:.2f} nm, "
                      f"{half_rc / ev:.2f} of the evaluations within the cutoff")
    j_groups(x, L, tree, 1.1)


if __name__ == "__main__":
    main()
