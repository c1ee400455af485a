#!/usr/bin/env python3
"""Per-wave time stamps of k_build (library built with -DMHIP_STAMPS=1, MOLLYHIP_DBG_TIMES=1, MOLLYHIP_DBG_DUMP=path: the search kernel's stamps land in path.build): which blocks a search waits for.

    python tools/build_times.py dump.bin [waves per block = lanes / 64 of the launch shape: 16 for 64 x 16, 8 for 256 x 2]
"""
import sys

import numpy as np


def main():
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    raw = np.fromfile(sys.argv[1], dtype=np.uint64)
    d = raw[: len(raw) // (nw * 8) * (nw * 8)].reshape(-1, nw, 8)      # (the buffer is sized for 16 waves per block; a launch of fewer fills its front)
    d = d[d[:, 0, 3] != 0]
    t = d[:, :, 0:4].astype(np.int64)
    t0 = t[:, :, 0].min()
    us = (t - t0) * 0.01
    found, tile, nxl = d[:, :, 4].astype(np.int64), d[:, 0, 5].astype(np.int64), d[:, :, 6].astype(np.int64)
    start, end = us[:, :, 0].min(axis=1), us[:, :, 3].max(axis=1)
    start -= start.min(); end -= us[:, :, 0].min()
    stage, search = (us[:, :, 1] - us[:, :, 0]).max(axis=1), (us[:, :, 2] - us[:, :, 1]).max(axis=1)
    fine = d[:, :, 7].any() and (d[:, :, 6] > d[:, :, 0]).all()      # slot 6 is a time stamp only for systems without exception lists (else: the lists' lengths)
    if fine:      # finer stamps of the staging part (systems without exception lists): boxes | cell pruning + scan | atom pruning + compaction (+ the cell-offset scan)
        t7 = (d[:, :, 7].astype(np.int64) - t0) * 0.01; t6 = (d[:, :, 6].astype(np.int64) - t0) * 0.01
        for name, a in (("  boxes", (t7 - us[:, :, 0]).max(axis=1)), ("  cell pruning + scan", (t6 - t7).max(axis=1)), ("  atoms: fetch, prune, compact", (us[:, :, 1] - t6).max(axis=1))):
            print(f"  {name:24s} mean {a.mean():.1f} p10 {np.percentile(a, 10):.1f} p50 {np.median(a):.1f} p90 {np.percentile(a, 90):.1f} max {a.max():.1f} us")
    print(f"{len(d)} blocks, first entry -> last exit {end.max():.1f} us; starts within {start.max():.1f} us")
    for name, a in (("boxes + cells + staging", stage), ("search", search), ("block total", end - start)):
        print(f"  {name:24s} mean {a.mean():.1f} p10 {np.percentile(a, 10):.1f} p50 {np.median(a):.1f} p90 {np.percentile(a, 90):.1f} max {a.max():.1f} us")
    if fine:
        nxl = nxl * 0
    x = nxl.sum(axis=1) / float(nw) / 64.0
    print(f"  exception-list entries per atom of a block: mean {x.mean():.1f} p10 {np.percentile(x, 10):.1f} p90 {np.percentile(x, 90):.1f} max {x.max():.1f}")
    print(f"  correlation of a block's search time with: exception entries {np.corrcoef(search, x)[0, 1]:.2f} | entries found {np.corrcoef(search, found.sum(axis=1))[0, 1]:.2f} | tile size {np.corrcoef(search, tile)[0, 1]:.2f}")
    o = np.argsort(-search)[:8]
    print("  slowest searches: " + " | ".join(f"b{i} {search[i]:.0f} us, tile {tile[i]}, found {found[i].sum()}, exc/atom {x[i]:.1f}" for i in o))
    o = np.argsort(search)[:4]
    print("  fastest searches: " + " | ".join(f"b{i} {search[i]:.0f} us, tile {tile[i]}, found {found[i].sum()}, exc/atom {x[i]:.1f}" for i in o))


if __name__ == "__main__":
    main()
