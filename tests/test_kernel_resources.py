"""Structural properties of the hot kernel that its speed depends on, checked on the compiler's output (no GPU needed: hipcc
cross-compiles gfx950).  The packed one-type LJ loop runs four 512-lane blocks per CU only while it stays within 64 VGPRs and uses
no scratch; DESIGN §4 records what a fifth of that occupancy costs (−12 %)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "molly.jl_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_uniform_lj_kernels_fit_64_vgprs_without_scratch(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "forces_uniform.s"
    # the flags of csrc/Makefile for this translation unit
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-slp-vectorize", "--cuda-device-only", "-S",
                    os.path.join(CSRC, "forces_uniform.hip"), "-o", str(out)], check=True, capture_output=True, timeout=900)
    kern, info = None, {}
    for line in open(out):
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", line)
        if m:
            kern = m.group(1); info[kern] = {}
            continue
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|private_segment_fixed_size) (\d+)", line)
        if m and kern:
            info[kern][m.group(1)] = int(m.group(2))
    names = subprocess.run(["c++filt"], input="\n".join(info), capture_output=True, text=True).stdout.split("\n")
    plain = {n: info[k] for k, n in zip(info, names) if "k_forces<float, 3, 0, false, false, false, false" in n}   # not SEG, not PRUNE: the passes of every step
    assert len(plain) == 3                                            # the three tile strides
    for n, r in plain.items():
        assert r["next_free_vgpr"] <= 64, (n, r)
        assert r["private_segment_fixed_size"] == 0, (n, r)
    for k, n in zip(info, names):
        if "k_forces<" in n:
            assert info[k]["private_segment_fixed_size"] == 0, (n, info[k])   # no variant of this file spills
