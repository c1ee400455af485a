#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of a multi-rank run (one *_kernel_trace.csv per process): per process, the dispatches of a window of plain steps and the
histogram of "dispatches per MD step", a step being what lies between two consecutive launches that integrate (k_forces<…, STEP> or k_vv_mid)."""
import csv, glob, os, re, sys, collections

def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void ", "").replace("mhip::", "")
    m = re.match(r"k_forces<float, 3, 0, false, false, (\w+), (\w+), (\d+), (\w+), (\w+)>", n)
    if m:
        seg, prune, stride, step, halo = m.groups()
        return "k_forces<" + ("PRUNE" if prune == "true" else "plain") + (",STEP" if step == "true" else "") + (",HALO" if halo == "true" else "") + ">"
    return re.sub(r"<.*", "<…>", n)

for path in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)):
    rows = list(csv.DictReader(open(path)))
    if len(rows) < 200: continue
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    t0 = [int(r["Start_Timestamp"]) for r in rows]; t1 = [int(r["End_Timestamp"]) for r in rows]
    integ = [i for i, n in enumerate(names) if n.startswith("k_forces<plain,STEP") or n.startswith("k_vv_mid")]
    if len(integ) < 50: continue
    per_step = collections.Counter(); kinds = collections.Counter()
    for a, b in zip(integ[:-1], integ[1:]):
        seq = tuple(names[a + 1:b + 1])
        per_step[len(seq)] += 1; kinds[seq] += 1
    print(f"== {os.path.basename(os.path.dirname(path))}/{os.path.basename(path)}: {len(rows)} dispatches, {len(integ)} integrating launches")
    print("   dispatches per step (count of steps):", dict(sorted(per_step.items())))
    for seq, c in kinds.most_common(4):
        print(f"   {c:5d} x  " + " | ".join(seq))
    # a window of ten plain steps from the middle, with durations and gaps (µs)
    mid = integ[len(integ) // 2]
    print("   window (name, duration µs, gap to the previous dispatch's end µs):")
    for i in range(mid, min(mid + 24, len(rows))):
        print(f"      {names[i]:34s} {(t1[i] - t0[i]) / 1e3:8.1f} {(t0[i] - t1[i - 1]) / 1e3:8.1f}")
