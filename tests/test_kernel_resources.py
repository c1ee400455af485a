"""Structural properties of the hot kernel that its speed depends on, checked on the compiler's output (no GPU needed: hipcc
cross-compiles gfx950).  The packed one-type LJ loop runs four 512-lane blocks per CU only while it stays within 64 VGPRs; it must
not EXECUTE a spill anywhere (a reload is a memory round trip on the critical path of a wave that lives 20 µs: 10-25 % per pass with
13 spill instructions outside every loop, profiles/r05_force_ab.txt §10), and its row loop must keep its wait pattern (§9: the same
instructions re-scheduled around lgkmcnt(0) walked the rows 40-65 % slower)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "molly.jl_amd", "csrc")


def _compile(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "forces_uniform.s"
    # the flags of csrc/Makefile for this translation unit
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-slp-vectorize", "--cuda-device-only", "-S",
                    os.path.join(CSRC, "forces_uniform.hip"), "-o", str(out)], check=True, capture_output=True, timeout=900)
    return out


def _kernels(asm):
    """name -> (resource dict, body lines)"""
    info, body, label = {}, {}, None
    kern = None
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            label = m.group(1); body[label] = []
            continue
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", line)
        if m:
            kern = m.group(1); info[kern] = {}; label = None
            continue
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|private_segment_fixed_size|group_segment_fixed_size) (\d+)", line)
        if m and kern:
            info[kern][m.group(1)] = int(m.group(2))
        if label:
            body[label].append(line)
    names = subprocess.run(["c++filt"], input="\n".join(info), capture_output=True, text=True).stdout.split("\n")
    return {n: (info[k], body.get(k, [])) for k, n in zip(info, names)}


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_uniform_lj_kernels_fit_64_vgprs_without_scratch(tmp_path):
    ks = _kernels(_compile(tmp_path))
    plain = {n: v for n, v in ks.items() if "k_forces<float, 3, 0, false, false, false, false" in n}   # not SEG, not PRUNE: the passes of every step
    # the three tile strides, each as the plain pass, as the STEP variant (round 5: the integrator in the epilogue) and as the STEP + HALO variant (round 6: the
    # fused step of a ghosted sub-domain — waits for the peers' rows in its prologue, stores to the peers in its epilogue) — none of which may cost the loop a register
    # … and (round 6, third session) as the STEP + LANG variant: the Langevin-middle update in the epilogue (Philox block, Box-Muller pair) instead of the velocity-Verlet one
    assert len(plain) == 12 and all(sum(1 for n in plain if t in n) == 3 for t in (", false, false, false>(", ", true, false, false>(", ", true, true, false>(", ", true, false, true>("))
    for n, (r, body) in plain.items():
        assert r["next_free_vgpr"] <= 64, (n, r)
        assert not any(re.match(r"\s+scratch_", l) for l in body), n      # no spill instruction anywhere in the kernel
        # the packed loop itself: two rows per trip = 68 packed instructions, loop control on the scalar unit (no exec-mask loop)
        text = "".join(body)
        assert "v_pk_fma_f32" in text and "clamp" in text
    for n, (r, _) in ks.items():
        if "k_forces<" in n:
            # no variant of this file spills: not one scratch instruction.  (A private segment in the DESCRIPTOR that no instruction touches — the compiler leaves 36
            # bytes behind in the STEP variants when it moves scalars into VGPR lanes — costs nothing: tools/micro/scratch_cost.hip, ±0.5 %.  A spill that is
            # EXECUTED costs a memory round trip of a wave that lives 20 µs, wherever it sits: profiles/r05_force_ab.txt §10.)
            assert not any(re.match(r"\s+scratch_", l) for l in ks[n][1]), n
            # … and none has a __shared__ array: the packed loop addresses the tile from LDS address 0 (kernels.h, walk_rows), so the
            # launch's dynamic LDS must start there (a helper with a static array, tried in an epilogue, moved the tile by 128 bytes: NaN forces)
            assert r["group_segment_fixed_size"] == 0, (n, r)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_packed_loop_issue_count(tmp_path):
    """The hot loop of the plain pass: at most 48 VALU instructions per row of four partners (round 2: 61, plus 15 s_nop), one
    address instruction per partner, row count in a scalar register."""
    ks = _kernels(_compile(tmp_path))
    counts = {}
    base = "k_forces<float, 3, 0, false, false, false, false, 3073, "
    tags = ("false, false, false>", "true, false, false>", "true, true, false>", "true, false, true>")      # plain, STEP, STEP + HALO, STEP + LANG
    for t in tags:
        counts[base + t] = _loop_counts(ks, base + t)
    a, b2, c, d = (counts[base + t] for t in tags)
    assert a[:2] == b2[:2] == c[:2] == d[:2]      # the same loop in all four (each held to the wait pattern by _loop_counts; the exact stages may differ by one)


def _loop_counts(ks, variant):
    n, (r, body) = next((n, v) for n, v in ks.items() if variant in n)
    # the innermost loop that holds the clamped packed fma of the cutoff
    blocks, cur = [], []
    for line in body:
        if re.match(r"^\.LBB\d+_\d+:", line):
            blocks.append(cur); cur = []
        cur.append(line)
    blocks.append(cur)
    loops = [b for b in blocks if any("clamp" in l for l in b) and any(re.search(r"s_cbranch_scc[01]", l) for l in b)]
    assert loops, "packed loop with scalar loop control not found"
    loop = max(loops, key=len)
    valu = [l for l in loop if re.match(r"\s+v_", l)]
    rows = sum(1 for l in loop if "clamp" in l) / 2.0               # two clamped fmas per row
    assert rows >= 1
    assert len(valu) / rows <= 48, (len(valu), rows)
    assert not any("scratch_" in l for l in loop)
    assert sum(1 for l in loop if "ds_read_b32" in l) == 12 * rows
    # the SCHEDULE, not only the instruction count: a row's twelve LDS reads are issued together and waited for in stages, and the row fetched ahead
    # stays in flight (a variant whose loop had the same 87 VALU instructions but drained lgkmcnt after every pair of reads and vmcnt once per trip
    # walked its rows 40-65 % slower: profiles/r05_force_ab.txt §9)
    waits = sorted(re.sub(r"\s+", " ", l.strip()) for l in loop if "s_waitcnt" in l)
    assert sum(1 for w in waits if w == "s_waitcnt lgkmcnt(0)") <= rows + 1 and not any("vmcnt(0)" in w for w in waits), waits
    assert sum(1 for w in waits if re.match(r"s_waitcnt lgkmcnt\((4|6|8|10)\)", w)) >= rows, waits      # … the reads ARE waited for in stages
    return len(valu), rows, tuple(waits)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_charge_spreading_uses_integer_lds_atomics(tmp_path):
    """ds_add_f32 runs at a tenth of the rate of ds_add_u64 on gfx950 (tools/micro/lds_atomic_rate.hip): the PME charge
    spreading accumulates its LDS sub-mesh in 64-bit fixed point.  Checked on the compiler's output so that a float atomic cannot slip
    back into that loop unnoticed; the flush to the global mesh stays a float atomic."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "pme_spread.s"
    src = tmp_path / "pme_spread.hip"
    src.write_text('#include "pme.h"\nnamespace mhip { template __global__ void k_pme_spread<float, 5, 64>(int64_t, const Vec<float>::T4*, float*, PmeP<float>);\n'
                   'template __global__ void k_pme_spread<double, 5, 64>(int64_t, const Vec<double>::T4*, double*, PmeP<double>); }\n')
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S", "-I", CSRC, str(src), "-o", str(out)],
                   check=True, capture_output=True, timeout=900)
    text = open(out).read()
    assert text.count(".amdhsa_kernel") == 2
    assert "ds_add_u64" in text and "ds_add_f32" not in text and "ds_add_rtn_f32" not in text and "ds_add_f64" not in text
    assert "global_atomic_add_f32" in text

