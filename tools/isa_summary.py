#!/usr/bin/env python3
"""Per-kernel resource table of a gfx950 assembly file (hipcc -S --cuda-device-only): VGPRs, SGPRs, scratch bytes, instruction count, VALU count, spill instructions.
usage: tools/isa_summary.py a.s [b.s]   — with two files: the kernels whose figures differ."""
import re, subprocess, sys

def table(path):
    info, body, label, kern = {}, {}, None, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m: label = m.group(1); body[label] = []; continue
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", line)
        if m: kern = m.group(1); info[kern] = {}; label = None; continue
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|private_segment_fixed_size) (\d+)", line)
        if m and kern: info[kern][m.group(1)] = int(m.group(2))
        if label and re.match(r"\s+[a-z]", line): body[label].append(line)
    names = subprocess.run(["c++filt"], input="\n".join(info), capture_output=True, text=True).stdout.split("\n")
    out = {}
    for k, n in zip(info, names):
        b = body.get(k, [])
        out[n] = (info[k].get("next_free_vgpr", 0), info[k].get("next_free_sgpr", 0), info[k].get("private_segment_fixed_size", 0), len(b),
                  sum(1 for l in b if re.match(r"\s+v_", l)), sum(1 for l in b if re.match(r"\s+scratch_", l)))
    return out

if __name__ == "__main__":
    a = table(sys.argv[1])
    if len(sys.argv) > 2:
        b = table(sys.argv[2])
        for n in sorted(set(a) | set(b)):
            if a.get(n) != b.get(n): print(f"{n[:150]}\n    {a.get(n)} -> {b.get(n)}")
    else:
        print("vgpr sgpr scratch_bytes instr valu spill_instr  kernel")
        for n, v in sorted(a.items()): print(*v, n[:150])
