"""Durations and the idle time in front of every kernel of a rocprofv3 kernel trace (--kernel-trace --output-format csv), grouped by (kernel before, kernel):
where a step's wall time goes that no kernel accounts for.   python tools/kernel_gaps.py <kernel_trace.csv> [skip_first_n]"""
import csv
import re
import statistics as st
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 3


def short(n):
    m = re.search(r"k_forces<float, 3, 0, false, false, false, (true|false), \d+, (true|false)>", n)
    if m:
        return "k_forces" + (" prune" if m.group(1) == "true" else "") + (" STEP" if m.group(2) == "true" else "")
    return re.sub(r"^void |mhip::|\(.*$|<.*$", "", n)[:30]


ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows][skip:]
dur, gap = defaultdict(list), defaultdict(list)
for a, b in zip(ks, ks[1:]):
    dur[b[0]].append((b[2] - b[1]) / 1e3)
    gap[(a[0], b[0])].append((b[1] - a[2]) / 1e3)
total = (ks[-1][2] - ks[0][1]) / 1e3
busy = sum(sum(v) for v in dur.values())
print(f"{len(ks)} kernels over {total:.0f} us: {busy:.0f} us inside kernels, {total - busy:.0f} us between them ({100 * (total - busy) / total:.1f} %)")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print(f"  {n:32s} n {len(v):6d}  median {st.median(v):8.2f} us  total {sum(v):10.0f} us")
print("idle time in front of a kernel, by (kernel before -> kernel):")
for (a, b), v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print(f"  {a:30s} -> {b:30s} n {len(v):6d}  median {st.median(v):6.2f} us  p90 {sorted(v)[len(v) * 9 // 10]:6.2f}  total {sum(v):9.0f} us")
