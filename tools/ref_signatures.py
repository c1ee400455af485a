#!/usr/bin/env python3
"""Method heads of the reference's plug-in points (the generics the INTEGRATION.md shim overrides) → tests/golden/reference_signatures.json.

Julia is not installed in this image, so nothing can load the shim against Molly.jl.  What CAN be checked statically is the class of bug that breaks
dispatch silently — a method head whose positional arguments do not line up with the generic's and with the call sites — so the reference's heads and
call sites are extracted here (this container has /root/reference; the GPU box does not, hence the committed fixture) and
tests/test_integration_md.py holds INTEGRATION.md's heads against them.

    python tools/ref_signatures.py [/root/reference]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (key, file, first line of the head or call)
HEADS = [
    ("pairwise_forces_loop_gpu!/cuda", "ext/MollyCUDAExt.jl", 845),
    ("pairwise_pe_loop_gpu!/cuda", "ext/MollyCUDAExt.jl", 936),
    ("remove_CM_motion!/cuda", "ext/MollyCUDAExt.jl", 2373),
    ("pairwise_forces_loop_gpu!/generic", "src/kernels.jl", 91),
    ("pairwise_pe_loop_gpu!/generic", "src/kernels.jl", 393),
]
CALLS = [
    ("pairwise_forces_loop_gpu!/call_nonl", "src/force.jl", 1223),
    ("pairwise_forces_loop_gpu!/call_nl", "src/force.jl", 1228),
    ("pairwise_pe_loop_gpu!/call_nonl", "src/energy.jl", 422),
    ("pairwise_pe_loop_gpu!/call_nl", "src/energy.jl", 427),
]


def balanced(text, start):
    """text[start] == '(' → (inner, index behind the matching ')')"""
    depth, i = 0, start
    while True:
        ch = text[i]
        depth += ch in "({["
        depth -= ch in ")}]"
        i += 1
        if depth == 0:
            return text[start + 1:i - 1], i


def split_top(s):
    out, cur, d = [], "", 0
    for ch in s:
        d += ch in "({["
        d -= ch in ")}]"
        if ch == "," and d == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def parse_params(inner):
    """positional parameters of a Julia method head: [(name or "", type or "")]; keyword parameters (behind ';') are returned separately"""
    pos, kw = inner, ""
    d = 0
    for i, ch in enumerate(inner):
        d += ch in "({["
        d -= ch in ")}]"
        if ch == ";" and d == 0:
            pos, kw = inner[:i], inner[i + 1:]
            break
    params = []
    for p in split_top(pos):
        p = " ".join(p.split())
        p = re.sub(r"\s*=.*$", "", p)                      # default value
        if "::" in p:
            name, ty = p.split("::", 1)
        else:
            name, ty = p, ""
        params.append([name.strip(), ty.strip()])
    kws = [re.sub(r"\s*=.*$", "", " ".join(k.split())).split("::")[0].strip() for k in split_top(kw)]
    return params, kws


def head_at(path, line):
    lines = open(path).read().split("\n")
    text = "\n".join(lines[line - 1:line + 12])
    m = re.match(r"\s*(?:@inline\s+)?function\s+([\w\.!]+)\s*\(", text)
    assert m, (path, line, text[:80])
    inner, _ = balanced(text, m.end() - 1)
    params, kws = parse_params(inner)
    return {"function": m.group(1).split(".")[-1], "positional": params, "keywords": kws, "where": f"{os.path.relpath(path, ref)}:{line}"}


def call_at(path, line, fname):
    lines = open(path).read().split("\n")
    text = "\n".join(lines[line - 1:line + 6])
    i = text.index(fname + "(")
    inner, _ = balanced(text, i + len(fname))
    args = split_top(inner.split(";")[0])
    return {"function": fname, "n_positional": len(args), "arguments": [" ".join(a.split()) for a in args], "where": f"{os.path.relpath(path, ref)}:{line}"}


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = {"reference": "JuliaMolSim/Molly.jl (as checked out under /root/reference)", "heads": {}, "calls": {}}
    for key, f, ln in HEADS:
        out["heads"][key] = head_at(os.path.join(ref, f), ln)
    for key, f, ln in CALLS:
        out["calls"][key] = call_at(os.path.join(ref, f), ln, key.split("/")[0])
    dst = os.path.join(ROOT, "tests", "golden", "reference_signatures.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for k, v in out["heads"].items():
        print(k, [p[0] or "::" + p[1] for p in v["positional"]], v["where"])
    for k, v in out["calls"].items():
        print(k, v["n_positional"], v["where"])
