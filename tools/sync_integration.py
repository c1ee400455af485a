#!/usr/bin/env python3
"""INTEGRATION.md prints the Julia files of julia/ verbatim: everything between `<!-- BEGIN FILE path -->` and `<!-- END FILE path -->` is regenerated from
`path` (a fenced block).  tests/test_integration_md.py fails when the two differ.     python tools/sync_integration.py [--check]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LANG = {".jl": "julia", ".toml": "toml", ".fragment": "toml"}


def render(text):
    def sub(m):
        path = m.group(1)
        body = open(os.path.join(ROOT, path), encoding="utf-8").read().rstrip("\n")
        lang = LANG.get(os.path.splitext(path)[1], "")
        return f"<!-- BEGIN FILE {path} -->\n```{lang}\n{body}\n```\n<!-- END FILE {path} -->"
    return re.sub(r"<!-- BEGIN FILE (\S+) -->.*?<!-- END FILE \1 -->", sub, text, flags=re.S)


if __name__ == "__main__":
    p = os.path.join(ROOT, "INTEGRATION.md")
    old = open(p, encoding="utf-8").read()
    new = render(old)
    if "--check" in sys.argv:
        sys.exit(0 if new == old else 1)
    if new != old:
        open(p, "w", encoding="utf-8").write(new)
        print("INTEGRATION.md updated")
