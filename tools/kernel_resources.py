#!/usr/bin/env python3
"""Resource table of the gfx950 kernels in a device-only assembly file (hipcc --cuda-device-only -S):
VGPRs, SGPRs, scratch bytes per lane (spills), LDS, plus counts of scratch_ instructions per kernel.

    python tools/kernel_resources.py /tmp/dis/engine.s [--filter k_forces] [--top 30]
"""
import argparse
import re
import subprocess


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except FileNotFoundError:
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--filter", default="")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    info, body, cur = {}, {}, None
    keys = {".amdhsa_next_free_vgpr": "vgpr", ".amdhsa_next_free_sgpr": "sgpr", ".amdhsa_private_segment_fixed_size": "scratch",
            ".amdhsa_group_segment_fixed_size": "lds", ".amdhsa_accum_offset": "accum"}
    label = None
    for line in open(a.asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            label = m.group(1); body[label] = 0
            continue
        if label and "scratch_" in line and not line.lstrip().startswith(";"):
            body[label] += 1
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", line)
        if m:
            cur = m.group(1); info[cur] = {}
            continue
        for k, short in keys.items():
            m = re.match(r"\s*" + re.escape(k) + r"\s+(\d+)", line)
            if m and cur:
                info[cur][short] = int(m.group(1))
    dm = demangle(list(info))
    rows = sorted(((v.get("scratch", 0), v.get("vgpr", 0), v.get("sgpr", 0), v.get("lds", 0), body.get(k, 0), dm[k]) for k, v in info.items()), reverse=True)
    rows = [r for r in rows if a.filter in r[5]]
    print(f"{len(rows)} kernels, {sum(1 for r in rows if r[0] > 0)} with scratch")
    print("scratchB vgpr sgpr ldsB scratch_instr kernel")
    for r in rows[:a.top]:
        print(f"{r[0]:7d} {r[1]:4d} {r[2]:4d} {r[3]:5d} {r[4]:6d}  {r[5][:170]}")


if __name__ == "__main__":
    main()
