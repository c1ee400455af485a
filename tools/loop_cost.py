"""Static cost model of a kernel's loops from hipcc's assembly (-S --cuda-device-only): for every innermost loop (a backward branch target
.. the branch) the instruction mix and an issue-time estimate per trip with the rates measured by tools/micro/valu_rate.hip on gfx950
(cycles per wave-instruction per SIMD at eight waves: packed fp32 5.4, plain fp32 4.2, transcendental 9.0, integer / move 3.0 — LDS and
scalar instructions are counted, not priced).   python tools/loop_cost.py file.s 'mangled-name-substring' [label-from label-to]"""
import re
import sys
from collections import Counter

TRANS = ("v_rsq_f32", "v_rcp_f32", "v_exp_f32", "v_sqrt_f32", "v_log_f32", "v_rcp_f64", "v_rsq_f64")


def classify(op):
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith(("v_mov", "v_and", "v_or", "v_lshl", "v_lshr", "v_bfe", "v_add_u", "v_sub_u", "v_add_co", "v_cndmask", "v_cmp", "v_bitop", "v_perm", "v_alignbit", "v_readfirstlane", "v_readlane", "v_writelane", "v_add3", "v_mad_u", "v_mul_lo", "v_mul_u", "v_min_u", "v_max_u", "v_min_i", "v_max_i", "v_ashr", "v_xor", "v_not", "v_sub_co", "v_subrev", "v_mbcnt", "v_accvgpr")):
        return "int"
    if op.startswith("v_"):
        return "fp"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_nop":
        return "nop"
    if op.startswith("s_waitcnt"):
        return "wait"
    return "salu"


RATE = {"pk": 5.4, "fp": 4.2, "trans": 9.0, "int": 3.0}


def kernel_lines(path, name):
    out, on = [], False
    for ln in open(path):
        if not on and ln.startswith("_Z") and name in ln and ln.rstrip().split(":")[0].endswith(name.split("$")[-1]) or (not on and re.match(r"^[_A-Za-z0-9]+:", ln) and name in ln):
            on = True
        if on:
            out.append(ln.rstrip("\n"))
            if "s_endpgm" in ln:
                break
    return out


def mix(lines):
    c = Counter()
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":") or re.match(r"^\.?[A-Za-z_0-9]+:", t):
            continue
        c[classify(t.split()[0])] += 1
    return c


def report(tag, lines):
    c = mix(lines)
    cyc = sum(RATE.get(k, 0) * v for k, v in c.items())
    print(f"{tag}: " + " ".join(f"{k}={c[k]}" for k in ("pk", "fp", "trans", "int", "lds", "vmem", "salu", "nop", "wait")) + f"  VALU={c['pk'] + c['fp'] + c['trans'] + c['int']}  issue≈{cyc:.0f} cycles")
    return cyc


if __name__ == "__main__":
    ls = kernel_lines(sys.argv[1], sys.argv[2])
    if len(sys.argv) >= 5:
        a = next(i for i, l in enumerate(ls) if l.startswith(sys.argv[3] + ":"))
        b = next(i for i, l in enumerate(ls) if l.startswith(sys.argv[4] + ":") and i > a)
        report(f"{sys.argv[3]}..{sys.argv[4]}", ls[a:b])
        sys.exit(0)
    labels = {l.split(":")[0]: i for i, l in enumerate(ls) if re.match(r"^\.LBB[0-9_]+:", l)}
    for i, l in enumerate(ls):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB[0-9_]+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = ls[labels[m.group(1)]:i + 1]
            if any("ds_read" in x or "ds_load" in x for x in body) and not any(re.match(r"^\s+s_cbranch_\w+\s+(\.LBB[0-9_]+)", x) and labels.get(re.match(r"^\s+s_cbranch_\w+\s+(\.LBB[0-9_]+)", x).group(1), 1 << 30) < labels[m.group(1)] for x in body[:-1]):
                report(f"loop {m.group(1)} ({len(body)} lines)", body)
