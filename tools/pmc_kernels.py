"""Average PMC counters per kernel from a rocprofv3 --pmc csv (diagnostic)."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:int(__import__("os").environ.get("PMC_NAME_CHARS", "46"))]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
want = sys.argv[2:]
for k in acc:
    if not want or any(w in k for w in want):
        print(k, cnt[k], {c: round(v / cnt[k], 1) for c, v in acc[k].items()})
