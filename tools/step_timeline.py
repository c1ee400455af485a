"""Timeline of one MD step from a rocprofv3 kernel trace (--kernel-trace --output-format csv): for every kernel between two consecutive
integrating launches the median start offset, duration and the gap to the kernel before it.   python tools/step_timeline.py <kernel_trace.csv>"""
import csv
import re
import statistics as st
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"^void |mhip::|\(.*$|<.*$", "", n)[:28]
steps, cur = [], []
for r in rows:
    cur.append((short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    # a step ends with the launch that integrates: k_vv_mid, or — round 5 — the last force launch of a small system (k_gather_collect_vv) / the pair pass of a large fluid (STEP)
    if "k_vv_mid" in r["Kernel_Name"] or "k_gather_collect_vv" in r["Kernel_Name"] or re.search(r"k_forces<float, 3, 0, false, false, false, false, \d+, true>", r["Kernel_Name"]):
        steps.append(cur); cur = []
shape = Counter(tuple(k[0] for k in s) for s in steps).most_common(1)[0][0]      # the most common step shape
sel = [s for s in steps if tuple(k[0] for k in s) == shape][5:]
print(f"{len(sel)} steps of the shape {shape}")
for i, name in enumerate(shape):
    dur = st.median((s[i][2] - s[i][1]) / 1e3 for s in sel)
    gap = st.median((s[i][1] - (s[i - 1][2] if i else s[i][1])) / 1e3 for s in sel)
    off = st.median((s[i][1] - s[0][1]) / 1e3 for s in sel)
    print(f"{name:30s} start +{off:7.2f} us  gap before {gap:5.2f}  duration {dur:6.2f}")
pairs = [(a, b) for a, b in zip(steps[5:], steps[6:]) if tuple(k[0] for k in a) == shape and tuple(k[0] for k in b) == shape]
print("step (first start -> last end) median %.2f us; start-to-start of consecutive steps median %.2f us" % (
    st.median((s[-1][2] - s[0][1]) / 1e3 for s in sel), st.median((b[0][1] - a[0][1]) / 1e3 for a, b in pairs) if pairs else 0.0))
