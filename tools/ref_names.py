#!/usr/bin/env python3
"""Names the Julia front ends (julia/) rely on in the reference → tests/golden/reference_names.json.

Nothing here can load julia/ext/MollyHIPExt.jl against Molly.jl (no Julia in the image), so what can drift silently — a function, type or struct FIELD of
Molly that the shim uses and the reference renamed — is checked statically: this tool (run where /root/reference exists; the GPU box has only the committed
fixture) records every name the reference exports, every top-level function / struct / const it defines under src/ and ext/MollyCUDAExt.jl, and the field
names of every struct; tests/test_integration_md.py holds the names and fields used under julia/ to them.

    python tools/ref_names.py [/root/reference]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    files = []
    for base, _, names in os.walk(os.path.join(ref, "src")):
        files += [os.path.join(base, n) for n in names if n.endswith(".jl")]
    files.append(os.path.join(ref, "ext", "MollyCUDAExt.jl"))
    exported, defined, fields = set(), set(), {}
    for path in sorted(files):
        text = open(path, encoding="utf-8").read()
        text = re.sub(r'"""(.*?)"""', "", text, flags=re.S)      # docstrings quote code
        lines = text.split("\n")
        k = 0
        while k < len(lines):
            line = lines[k]
            m = re.match(r"\s*export\b(.*)", line)
            if m:
                buf = m.group(1)
                while buf.rstrip().endswith(",") and k + 1 < len(lines):
                    k += 1; buf += lines[k]
                exported.update(t for t in re.findall(r"[^\s,]+", buf) if not t.startswith("#"))
            m = re.match(r"\s*(?:@\w+\s+)*(?:function\s+)(?:[A-Za-z_][\w.]*\.)?([^\s(.{]+!?)\s*[({]", line)
            if m:
                defined.add(m.group(1))
            m = re.match(r"^([A-Za-z_]\w*!?)\(.*\)\s*(?:where\s.*)?=\s", line)      # one-line method definitions
            if m:
                defined.add(m.group(1))
            m = re.match(r"\s*(?:Base\.@kwdef\s+|@kwdef\s+)?(?:mutable\s+)?struct\s+([A-Za-z_]\w*)", line)
            if m:
                name = m.group(1); defined.add(name)
                fs, k2 = [], k + 1
                depth = 1
                while k2 < len(lines) and depth > 0:
                    l2 = lines[k2].split("#")[0].strip()
                    if re.match(r"(function|if|for|while|let|begin|struct|do)\b", l2) or re.search(r"\bdo\b\s*(\w+\s*)?$", l2):
                        depth += 1
                    if l2 == "end" or l2.startswith("end "):
                        depth -= 1
                    elif depth == 1:
                        mm = re.match(r"([A-Za-zα-ωΑ-Ω_][\wα-ωΑ-Ω′₀-₉]*)\s*(::.*)?(=.*)?$", l2)
                        if mm and mm.group(1) not in ("end", "function", "new"):
                            fs.append(mm.group(1))
                    k2 += 1
                fields.setdefault(name, fs)
            m = re.match(r"\s*const\s+([A-Za-z_]\w*)\s*=", line)
            if m:
                defined.add(m.group(1))
            k += 1
    out = {"exported": sorted(exported), "defined": sorted(defined), "fields": {k: v for k, v in sorted(fields.items())}}
    path = os.path.join(ROOT, "tests", "golden", "reference_names.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0, sort_keys=True)
    print(f"{len(exported)} exported names, {len(defined)} definitions, {len(fields)} structs -> {path}")


if __name__ == "__main__":
    main()
