#!/bin/bash
# which kernels surround the runtime's fill / copy kernels?  (kernel trace of a short run, dispatch order)
out=$PWD/gpurun_out/trace_seq; mkdir -p $out; export TMPDIR=/tmp; wl=${1:-lj1m}; root=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/t -- python $root/bench.py --workload $wl --steps 300 --warmup 50 --profile-steps 0 --no-cpu-baseline --no-secondary --traffic file --equil 0 > /dev/null 2> $out/err.txt
python - "$out" <<'PY'
import csv, glob, sys, re, collections
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(k_[a-z0-9_]+|fillBuffer\w*|copyBuffer\w*|rocprim[\w:]*?(sort|scan|merge)\w*)", n)
    return m.group(1) if m else n[:30]
names = [short(r["Kernel_Name"]) for r in rows]
ctx = collections.Counter()
for i, n in enumerate(names):
    if "fillBuffer" in n or "copyBuffer" in n:
        ctx[(names[i - 1] if i else "-", n, names[i + 1] if i + 1 < len(names) else "-")] += 1
for k, v in ctx.most_common(25):
    print(v, k)
print("total", len(names), collections.Counter(names).most_common(12))
PY
rm -rf $out/t
