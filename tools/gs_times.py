#!/usr/bin/env python3
"""Per-wave time stamps of the group-split pair pass (library built with -DMHIP_STAMPS=1, MOLLYHIP_DBG_TIMES=n, MOLLYHIP_DBG_DUMP=file):
where a pass's time goes — per workgroup start, staging, row walk, reduction — and how evenly the compute units are loaded.

    python tools/gs_times.py dump.bin
"""
import sys

import numpy as np


def main():
    d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 8)       # [workgroup][wave][8]
    d = d[d[:, 0, 7] != 0]
    t = d[:, :, 4:8].astype(np.int64)
    t0 = t[:, :, 0].min()
    us = (t - t0) * 0.01
    rows = d[:, :, 1].astype(np.int64)
    blk = (d[:, 0, 2] & np.uint64(0xfffff)).astype(np.int64); grp = ((d[:, 0, 2] >> np.uint64(20)) & np.uint64(0xf)).astype(np.int64)
    hw = (d[:, 0, 2] >> np.uint64(32)).astype(np.int64); xcc = (d[:, 0, 3] & np.uint64(0xf)).astype(np.int64)
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    cu_key = xcc * 1000 + se * 100 + sh * 16 + cu
    start, end = us[:, :, 0].min(axis=1), us[:, :, 3].max(axis=1)
    print(f"{len(d)} workgroups, first entry -> last exit {end.max():.2f} us; workgroup starts: p50 {np.median(start):.2f} p90 {np.percentile(start, 90):.2f} max {start.max():.2f}")
    for name, a in (("staging", us[:, :, 1] - us[:, :, 0]), ("row walk", us[:, :, 2] - us[:, :, 1]), ("reduce+store", us[:, :, 3] - us[:, :, 2])):
        print(f"  {name:13s} per wave mean {a.mean():.2f} p10 {np.percentile(a, 10):.2f} p50 {np.median(a):.2f} p90 {np.percentile(a, 90):.2f} max {a.max():.2f} us")
    print(f"  rows per wave mean {rows.mean():.2f} p10 {np.percentile(rows, 10):.0f} p90 {np.percentile(rows, 90):.0f} max {rows.max()}")
    walk = (us[:, :, 2] - us[:, :, 1])
    print(f"  us per row: mean {(walk / np.maximum(rows, 1)).mean():.3f}; waves with the most rows: {np.sort(rows.ravel())[-8:]}")
    dur = end - start
    order = np.argsort(-end)[:10]
    print("  last workgroups to finish: " + " | ".join(f"b{blk[i]} g{grp[i]} start {start[i]:.1f} end {end[i]:.1f} rows {rows[i].max()}" for i in order))
    keys, inv = np.unique(cu_key, return_inverse=True)
    per_cu_rows = np.bincount(inv, weights=rows.sum(axis=1)); per_cu_n = np.bincount(inv); per_cu_end = np.zeros(len(keys)); np.maximum.at(per_cu_end, inv, end)
    print(f"  {len(keys)} compute units seen; workgroups per CU min {per_cu_n.min()} mean {per_cu_n.mean():.2f} max {per_cu_n.max()}; wave-rows per CU min {per_cu_rows.min():.0f} mean {per_cu_rows.mean():.0f} max {per_cu_rows.max():.0f}")
    print(f"  CU finish time: p10 {np.percentile(per_cu_end, 10):.2f} p50 {np.median(per_cu_end):.2f} p90 {np.percentile(per_cu_end, 90):.2f} max {per_cu_end.max():.2f}; corr(rows on CU, finish) {np.corrcoef(per_cu_rows, per_cu_end)[0, 1]:.2f}")
    # which workgroups share a compute unit (is the placement the round robin it looks like?)
    wg = np.nonzero(np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 8)[:, 0, 7] != 0)[0]
    same = sum(1 for k in range(len(keys)) if len(set(int(w) % 256 for w in wg[inv == k])) == 1)
    print(f"  compute units whose workgroups are all congruent mod 256: {same} of {len(keys)}; examples: " + " | ".join(str(list(wg[inv == k])) for k in range(3)))
    print(f"  workgroup duration mean {dur.mean():.2f} p90 {np.percentile(dur, 90):.2f} max {dur.max():.2f}; corr(duration, rows of the workgroup) {np.corrcoef(dur, rows.max(axis=1))[0, 1]:.2f}")


if __name__ == "__main__":
    main()
