#!/usr/bin/env python3
"""Condenses the rocprofv3 CSV output of profiles/collect.sh into one JSON summary (per kernel: calls,
average duration, PMC counters per launch; HBM bytes per launch with the gfx950 FETCH_SIZE×2 correction)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"k_forces<([^>]*)>", name)
    if m:   # the PRUNE instantiation (7th template argument; an 8th, the tile stride of the packed loop, may follow) is a different
        args = [a.strip() for a in m.group(1).split(",")]      # kernel: it also writes the inner pair list
        if len(args) > 8 and args[8] == "true":      # the STEP instantiation (9th argument): the pair pass with the integrator in its epilogue
            return "k_forces_step"
        return "k_forces_prune" if len(args) > 6 and args[6] == "true" else "k_forces"
    m = re.search(r"k_pme_dft<([^>]*)>", name)
    if m:
        mode = m.group(1).split(",")[1].strip()
        return {"0": "k_pme_dft", "1": "k_pme_dft_zfwd", "2": "k_pme_dft_xconv"}.get(mode, "k_pme_dft")
    m = re.search(r"(k_[a-z0-9_]+)", name)
    if m:
        return m.group(1)
    return name[:60]


def kernel_stats(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row.get("Name", ""))
            e = out.setdefault(k, {"calls": 0, "total_ns": 0.0})
            e["calls"] += int(float(row.get("Calls", 0)))
            e["total_ns"] += float(row.get("TotalDurationNs", 0))
    for e in out.values():
        e["avg_us"] = e["total_ns"] / max(e["calls"], 1) / 1e3
    return out


def counters(d):
    """per kernel: counter → (sum, dispatches)"""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row.get("Kernel_Name", ""))
            c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0))
            a = acc[k][c]; a[0] += v; a[1] += 1
    return {k: {c: {"per_launch": a[0] / max(a[1], 1), "launches": a[1]} for c, a in cs.items()} for k, cs in acc.items()}


def main():
    out_dir, workload = sys.argv[1], sys.argv[2]
    summ = {"workload": workload, "kernels": kernel_stats(os.path.join(out_dir, "trace")), "pmc": {}}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        for k, cs in counters(os.path.join(out_dir, sub)).items():
            summ["pmc"].setdefault(k, {}).update(cs)
    # the dominant kernel: k_forces, or — small systems with per-atom parameters — its group-split form k_forces_gs (the profiling pass of
    # bench.py runs it on its own; the timed region runs it inside k_pair_spread_bonded, beside the spreading and the bonded terms)
    dom = next((k for k in ("k_forces_step", "k_forces", "k_forces_gs") if summ["pmc"].get(k)), "k_forces")
    kf = summ["pmc"].get(dom, {})
    summ["dominant_kernel"] = dom
    if "FETCH_SIZE" in kf and "WRITE_SIZE" in kf:
        # rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B
        # (MI355X_MICROARCH.md §HBM) → ×2 on the read side
        rd = kf["FETCH_SIZE"]["per_launch"] * 1024 * 2
        wr = kf["WRITE_SIZE"]["per_launch"] * 1024
        summ["hbm_bytes_per_force_launch"] = rd + wr
        summ["hbm_read_bytes_per_force_launch"] = rd
        summ["hbm_write_bytes_per_force_launch"] = wr
    # one small file per kernel for bench.py's roofline entries (load_traffic): the plain pass, the pruning pass, the outer search — each with the id of the
    # library the counters were taken with (bench.py compares it with the library it runs)
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        build_id = hashlib.sha256(open(os.path.join(root, "molly.jl_amd", "libmollyhip.so"), "rb").read()).hexdigest()[:16]
    except OSError:
        build_id = None
    summ["lib_build_id"] = build_id
    sys.path.insert(0, root)
    try:
        import bench
        src_id = bench.kernel_src_id()
    except Exception:
        src_id = None
    summ["kernel_src_id"] = src_id
    tag = sys.argv[3] if len(sys.argv) > 3 else "rXX"
    for kind, kern in (("", dom), ("_prune", "k_forces_prune"), ("_build", "k_build")):
        c = summ["pmc"].get(kern, {})
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        rd = c["FETCH_SIZE"]["per_launch"] * 1024 * 2; wr = c["WRITE_SIZE"]["per_launch"] * 1024
        rec = {"workload": workload, "kernel": kern, "hbm_bytes_per_launch": rd + wr, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
               "launches_counted": c["FETCH_SIZE"]["launches"], "avg_us_rocprofv3": summ["kernels"].get(kern, {}).get("avg_us"), "lib_build_id": build_id, "kernel_src_id": src_id,
               "source": f"{tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, profiles/collect.sh), FETCH_SIZE x2 gfx950 correction, KiB units"}
        if kind == "":
            rec["hbm_bytes_per_force_launch"] = rd + wr
        for extra in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE"):
            if extra in c:
                rec[extra + "_per_launch"] = c[extra]["per_launch"]
        json.dump(rec, open(os.path.join(out_dir, f"{tag}_traffic_{workload}{kind}.json"), "w"), indent=1)
    try:
        summ["bench_line"] = json.loads(open(os.path.join(out_dir, "bench_trace.json")).read().strip().splitlines()[-1])
    except Exception as e:
        summ["bench_line_error"] = str(e)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
