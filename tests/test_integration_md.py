"""The reference-side binding ships as Julia FILES (julia/ext/MollyHIPExt.jl + mhip_abi.jl, julia/MollyHIP/, julia/test/runtests.jl) that INTEGRATION.md prints
verbatim; Julia is not installed here, so nothing executes them.  This guard keeps them from drifting: every `ccall((:mhip_…, libmollyhip), Ret, (ArgTypes…), …)`
of the files is checked against the prototype of include/mollyhip.h — the symbol exists, the argument count matches, every Julia argument type is compatible
with the C parameter type, the return type matches —, the `struct Mhip…` definitions mirror the C structs field by field, the overriding method heads line up
with the reference's, and every Molly name and struct field the files use exists in the reference (tests/golden/reference_names.json)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JULIA_FILES = ["julia/ext/mhip_abi.jl", "julia/ext/MollyHIPExt.jl", "julia/MollyHIP/src/MollyHIP.jl", "julia/test/runtests.jl"]


def julia_text(files=JULIA_FILES[:3]):
    return "\n".join(open(os.path.join(ROOT, f), encoding="utf-8").read() for f in files)


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def c_prototypes():
    """name -> (return type, [parameter types]) from include/mollyhip.h, types normalised ('const' dropped, names dropped)"""
    text = _strip_comments(open(os.path.join(ROOT, "include", "mollyhip.h")).read())
    out = {}
    for ret, name, params in re.findall(r"([A-Za-z_][\w\s\*]*?)\b(mhip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ps = []
        params = " ".join(params.split())
        if params not in ("", "void"):
            for p in params.split(","):
                p = p.replace("const", " ").strip()
                stars = p.count("*")
                base = re.sub(r"\*", " ", p).split()
                # drop the parameter name (last identifier) unless the type is a single word like "void"
                if len(base) > 1:
                    base = base[:-1]
                ps.append(" ".join(base) + "*" * stars)
        out[name] = (" ".join(ret.replace("const", " ").split()), ps)
    return out


def c_struct_fields(name):
    text = _strip_comments(open(os.path.join(ROOT, "include", "mollyhip.h")).read())
    m = re.search(r"typedef\s+struct\s+" + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", text, flags=re.S)
    assert m, f"struct {name} not found in the header"
    fields = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        mm = re.match(r"([A-Za-z_]\w*(?:\s+[A-Za-z_]\w*)*?)\s+(.+)$", decl)
        ctype, names = mm.group(1), mm.group(2)
        for nm in names.split(","):
            nm = nm.strip()
            arr = re.match(r"(\w+)\[(\d+)\]$", nm)
            fields.append((arr.group(1), ctype, int(arr.group(2))) if arr else (nm, ctype, 0))
    return fields


# Julia ccall argument type -> the C parameter types it may stand for
JULIA_TO_C = {
    "Int32": {"int32_t"}, "Int64": {"int64_t"}, "UInt64": {"uint64_t"}, "Float64": {"double"},
    "Ptr{Cvoid}": {"void*", "mhip_ctx*", "mhip_halo_plan*", "int32_t*", "uint8_t*", "float*", "double*"},   # opaque handles and raw device pointers
    "Ptr{Int32}": {"int32_t*"}, "Ptr{UInt8}": {"uint8_t*"}, "Ptr{Float64}": {"double*", "void*"}, "Ptr{UInt32}": {"uint32_t*"},
    "Ptr{T}": {"void*"},                                     # arrays of the working precision travel as const void*
    "Ref{Float64}": {"double*"}, "Ref{Int64}": {"int64_t*"}, "Ref{Int32}": {"int32_t*"},
    "Ref{Ptr{Cvoid}}": {"mhip_ctx**"}, "Ref{MhipConfig}": {"mhip_config*"}, "Ptr{MhipLaunchTrial}": {"mhip_launch_trial*"},
    "Cstring": {"char*"},
}
JULIA_FIELD = {"Int32": "int32_t", "Int64": "int64_t", "Float64": "double", "Float32": "float"}


def julia_ccalls(text):
    """[(symbol, return type, [arg types])] of every ccall((:mhip_…, libmollyhip), …) in the text"""
    calls = []
    for m in re.finditer(r"ccall\(\(:(mhip_[a-z0-9_]+),\s*libmollyhip\),\s*([A-Za-z0-9_{}]+),\s*\(", text):
        i, depth, start = m.end(), 1, m.end()
        while depth:                                          # the argument-type tuple, braces and parentheses balanced
            ch = text[i]
            depth += ch == "("; depth -= ch == ")"
            i += 1
        inner = text[start:i - 1]
        args, cur, d = [], "", 0
        for ch in inner:
            if ch in "{(":
                d += 1
            if ch in "})":
                d -= 1
            if ch == "," and d == 0:
                args.append(cur.strip()); cur = ""
            else:
                cur += ch
        if cur.strip():
            args.append(cur.strip())
        calls.append((m.group(1), m.group(2), args))
    return calls


def test_integration_md_prints_the_julia_files_verbatim():
    import importlib.util
    spec = importlib.util.spec_from_file_location("sync_integration", os.path.join(ROOT, "tools", "sync_integration.py"))
    si = importlib.util.module_from_spec(spec); spec.loader.exec_module(si)
    text = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read()
    assert si.render(text) == text, "INTEGRATION.md differs from the files under julia/: run tools/sync_integration.py"
    for f in JULIA_FILES + ["julia/Project.toml.fragment"]:
        assert f"<!-- BEGIN FILE {f} -->" in text, f
    frag = open(os.path.join(ROOT, "julia", "Project.toml.fragment")).read()
    assert re.search(r"\[weakdeps\]\s*\nAMDGPU = \"[0-9a-f-]{36}\"", frag) and re.search(r"\[extensions\]\s*\nMollyHIPExt = \"AMDGPU\"", frag)      # ≙ /root/reference/Project.toml:41-54


def test_every_ccall_matches_the_header():
    text = julia_text()
    protos = c_prototypes()
    calls = julia_ccalls(text)
    assert len(calls) >= 40, len(calls)
    seen = set()
    for sym, ret, args in calls:
        assert sym in protos, f"INTEGRATION.md calls {sym}, which include/mollyhip.h does not declare"
        c_ret, c_params = protos[sym]
        assert len(args) == len(c_params), f"{sym}: {len(args)} Julia argument types {args} vs C parameters {c_params}"
        for k, (jt, ct) in enumerate(zip(args, c_params)):
            assert jt in JULIA_TO_C, f"{sym}: unknown Julia argument type {jt}"
            assert ct in JULIA_TO_C[jt], f"{sym} argument {k + 1}: Julia {jt} cannot stand for C {ct}"
        assert (ret == "Int32" and c_ret == "int32_t") or (ret == "Cstring" and c_ret == "char*"), f"{sym}: return {ret} vs {c_ret}"
        seen.add(sym)
    # the boundary's core entry points are all bound somewhere in the text
    for need in ("mhip_create", "mhip_destroy", "mhip_set_atoms", "mhip_set_exceptions", "mhip_set_state", "mhip_get_state", "mhip_forces",
                 "mhip_potential_energy", "mhip_remove_cm", "mhip_vv_run", "mhip_langevin_run", "mhip_export_neighbors", "mhip_set_pme"):
        assert need in seen, need


def test_julia_structs_mirror_the_c_structs():
    text = julia_text()
    for jname, cname in (("MhipInteractions", "mhip_interactions"), ("MhipConfig", "mhip_config"), ("MhipLaunchTrial", "mhip_launch_trial")):
        m = re.search(r"struct " + jname + r"\n(.*?)\nend", text, flags=re.S)
        assert m, jname
        jfields = []
        for part in re.split(r"[;\n]", m.group(1)):
            part = part.strip()
            if part:
                nm, ty = part.split("::")
                jfields.append((nm.strip(), ty.strip()))
        cfields = c_struct_fields(cname)
        assert [f[0] for f in jfields] == [f[0] for f in cfields], (jname, [f[0] for f in jfields], [f[0] for f in cfields])
        for (nm, jt), (_, ct, arr) in zip(jfields, cfields):
            if arr:
                mm = re.match(r"NTuple\{(\d+),\s*(\w+)\}$", jt)
                assert mm and int(mm.group(1)) == arr and JULIA_FIELD[mm.group(2)] == ct, (jname, nm, jt, ct, arr)
            elif ct.startswith("mhip_"):
                assert jt == "MhipInteractions" and ct == "mhip_interactions", (jname, nm, jt, ct)
            else:
                assert JULIA_FIELD[jt] == ct, (jname, nm, jt, ct)


def test_entry_point_count_in_design_is_current():
    protos = c_prototypes()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"\((\d+) entry points", design)
    assert m and int(m.group(1)) == len(protos), (m and m.group(1), len(protos))


# ---- the shim's method heads against the reference's own (tests/golden/reference_signatures.json, made by tools/ref_signatures.py) ------------
def _golden_signatures():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_signatures.json")))


def _integration_heads(fname):
    """every `function fname(…)` head of INTEGRATION.md: [(positional [(name, type)], keywords)]"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_signatures", os.path.join(ROOT, "tools", "ref_signatures.py"))
    rs = importlib.util.module_from_spec(spec); spec.loader.exec_module(rs)
    text = julia_text()
    out = []
    for m in re.finditer(r"^function\s+(?:[A-Za-z]+\.)*" + re.escape(fname) + r"\s*\(", text, flags=re.M):
        inner, _ = rs.balanced(text, m.end() - 1)
        out.append(rs.parse_params(inner))
    return out


def test_overriding_method_heads_line_up_with_the_reference():
    """Nothing here can run Julia, so the class of bug that breaks dispatch without an error — a method head whose positional arguments do not
    line up with the generic's, so that the stock method keeps being called — is caught statically: for each generic the shim overrides
    (ext/MollyCUDAExt.jl:845, 936, 2373; generics src/kernels.jl:91, 393) the positional argument COUNT equals the reference's at the method
    heads and at the call sites (src/force.jl:1223, 1228; src/energy.jl:422, 427 — SURVEY §0 records a 5-against-6 mismatch of exactly this kind
    inside the reference), the argument NAMES are the CUDA extension's in the same order, `nbs::Nothing` and `::Val{needs_vir}` sit where the
    extension has them (they select the no-neighbour-list path), and the array type is the only thing specialised."""
    g = _golden_signatures()
    for fname in ("pairwise_forces_loop_gpu!", "pairwise_pe_loop_gpu!", "remove_CM_motion!"):
        ref = g["heads"][fname + "/cuda"]
        heads = _integration_heads(fname)
        # one method per way the reference calls the generic: `nbs::Nothing` (the GPUNeighborFinder route, force.jl:1228 / energy.jl:427) and — for the two pairwise
        # loops — `nbs::Molly.NoNeighborList` (the use_neighbors = false interactions, force.jl:1223 / energy.jl:422; the CUDA extension's own method for it, ext:757, lacks
        # the Val argument its call site passes — SURVEY §0 — so the head is held to the GENERIC's six arguments, which is what dispatch sees)
        assert len(heads) == (1 if fname == "remove_CM_motion!" else 2), (fname, len(heads))
        for k, (pos, kws) in enumerate(heads):
            assert len(pos) == len(ref["positional"]), (fname, pos, ref["positional"])
            assert not kws and not ref["keywords"]
            for (n_i, t_i), (n_r, t_r) in zip(pos, ref["positional"]):
                assert n_i == n_r, (fname, n_i, n_r)                               # same names in the same order (anonymous ::Val{…} included)
                if n_r == "sys":
                    assert t_r.replace(" ", "").startswith("System{") and "<:CuArray" in t_r and "<:ROCArray" in t_i, (fname, t_i, t_r)
                elif t_r == "Nothing":
                    assert t_i == ("Nothing" if k == 0 else "Molly.NoNeighborList"), (fname, k, t_i)
                elif t_r == "Val{needs_vir}":
                    assert t_i == t_r, (fname, n_i, t_i, t_r)
            for key, call in g["calls"].items():
                if key.startswith(fname + "/"):
                    assert call["n_positional"] == len(pos), (key, call, pos)
            if fname + "/generic" in g["heads"]:
                assert len(g["heads"][fname + "/generic"]["positional"]) == len(pos)


def test_reference_signature_fixture_is_current():
    """where the reference checkout is present (this container, not the GPU box) the committed fixture is what tools/ref_signatures.py extracts now"""
    import json
    import subprocess
    import sys
    import pytest
    if not os.path.isdir("/root/reference/ext"):
        pytest.skip("no reference checkout here")
    before = open(os.path.join(ROOT, "tests", "golden", "reference_signatures.json")).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_signatures.py"), "/root/reference"], check=True, capture_output=True)
    after = open(os.path.join(ROOT, "tests", "golden", "reference_signatures.json")).read()
    assert json.loads(before) == json.loads(after)


# ---- every Molly name and struct field the Julia files use, against the reference's sources (tests/golden/reference_names.json, tools/ref_names.py) ------
# what the shim reads of Molly's structs: (struct, [fields]) — a renamed field would be a runtime error on the first call, with no Julia here to raise it
FIELDS_USED = {
    "GPUNeighborFinder": ["dist_cutoff", "n_steps_reorder", "initialized", "cache_generation", "excluded_i", "excluded_j", "special_i", "special_j"],
    "BuffersGPU": ["fs_mat", "virial_nounits", "step_n_preprocessed"],
    "PME": ["mesh_dims", "order", "α"],
    "TriclinicBoundary": ["basis_vectors"], "CubicBoundary": ["side_lengths"],
    "Atom": ["charge", "σ", "λ"],
    "HarmonicBond": ["k", "r0"], "HarmonicAngle": ["k", "θ0"], "PeriodicTorsion": ["periodicities", "phases", "ks"],
    "AndersenThermostat": ["temperature", "coupling_const"],
    "System": ["atoms", "coords", "boundary", "velocities", "pairwise_inters", "specific_inter_lists", "general_inters", "virtual_sites", "neighbor_finder", "loggers", "energy_units", "k"],
    "InteractionList2Atoms": ["is", "js", "inters"], "InteractionList3Atoms": ["is", "js", "ks", "inters"], "InteractionList4Atoms": ["is", "js", "ks", "ls", "inters"],
    "NeighborList": ["n", "list"],
    "LennardJones": ["cutoff", "weight_special"], "Coulomb": ["cutoff", "weight_special", "coulomb_const"],
    "CoulombReactionField": ["dist_cutoff", "solvent_dielectric", "weight_special", "coulomb_const"],
    "CoulombEwald": ["dist_cutoff", "weight_special", "coulomb_const", "α", "approximate_erfc"],
    "DistanceCutoff": ["dist_cutoff"], "CubicSplineCutoff": ["dist_activation", "dist_cutoff"],
}
# Molly functions / types the files call or extend (qualified `Molly.x`, `import Molly: x`, or exported names used bare)
NAMES_USED = ["pairwise_forces_loop_gpu!", "pairwise_pe_loop_gpu!", "remove_CM_motion!", "uses_gpu_neighbor_finder", "simulate!", "random_velocities!", "from_device", "masses",
              "ustrip_vec", "optimize_cuda_launch_config!", "apply_loggers!", "float_type", "find_neighbors", "forces", "potential_energy", "wrap_coords", "MolecularForceField",
              "System", "Atom", "LennardJones", "Coulomb", "CoulombReactionField", "CoulombEwald", "NoCutoff", "DistanceCutoff", "ShiftedPotentialCutoff", "ShiftedForceCutoff",
              "CubicSplineCutoff", "PolynomialCutoff", "CubicBoundary", "TriclinicBoundary", "GPUNeighborFinder", "DistanceNeighborFinder", "NoNeighborFinder", "NeighborList",
              "InteractionList2Atoms", "InteractionList3Atoms", "InteractionList4Atoms", "HarmonicBond", "HarmonicAngle", "PeriodicTorsion", "EwaldExclusion", "PME",
              "AndersenThermostat", "VelocityVerlet", "NoNeighborList", "init_buffers!", "apply_coupling!", "needs_virial"]


def _reference_names():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_names.json"), encoding="utf-8"))


def test_molly_names_and_fields_used_by_the_julia_files_exist_in_the_reference():
    g = _reference_names()
    known = set(g["exported"]) | set(g["defined"])
    text = julia_text(JULIA_FILES)
    # (1) everything imported from / qualified with Molly in the files is in NAMES_USED (so that the list cannot silently fall behind the files) …
    used = set()
    code = re.sub(r"#[^\n]*", "", text)                                   # comments quote file names (Molly.jl) and prose
    for m in re.finditer(r"import Molly:\s*((?:[^\n,]+,\s*\n?\s*)*[^\n,]+)", code):
        used.update(t.strip() for t in m.group(1).replace("\n", " ").split(",") if t.strip())
    used.update(re.findall(r"\bMolly\.([A-Za-z_]\w*!?)", code))
    assert used <= set(NAMES_USED), sorted(used - set(NAMES_USED))
    # … and every name of the list is defined by the reference and really appears in the files
    for n in NAMES_USED:
        assert n in known, f"the Julia files use Molly's `{n}`, which the reference does not define"
        assert re.search(r"(?<![\w!])" + re.escape(n) + r"(?![\w!])", text), f"`{n}` is listed but not used"
    # (2) struct fields
    for st, fields in FIELDS_USED.items():
        assert st in g["fields"], st
        for f in fields:
            assert f in g["fields"][st], f"{st}.{f} is read by the Julia files but the reference's struct has {g['fields'][st]}"
            assert re.search(r"\." + re.escape(f) + r"(?![\w])", text), f"{st}.{f} is listed but no `.{f}` appears in the files"


def test_reference_names_fixture_is_current():
    """where the reference checkout is present the committed fixture is what tools/ref_names.py extracts now"""
    import json
    import subprocess
    import sys
    import pytest
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("no reference checkout here")
    path = os.path.join(ROOT, "tests", "golden", "reference_names.json")
    before = open(path, encoding="utf-8").read()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_names.py"), "/root/reference"], check=True, capture_output=True)
    assert json.loads(before) == json.loads(open(path, encoding="utf-8").read())


def test_runtests_restates_the_reference_systems():
    """julia/test/runtests.jl carries the closed-form systems of test/gpu_consistency.jl (33 / 100 / 10 / 20 atoms, their boxes, cutoffs, exception pairs) and the
    reference's tolerances — the numbers SURVEY §8(c) lists — and takes its 6mrr bars from test/protein.jl:267-299"""
    t = open(os.path.join(ROOT, "julia", "test", "runtests.jl"), encoding="utf-8").read()
    for needle in ("diagonal(33)", "CubicBoundary(T64(20))", "lj_atoms(100, 1.0)", "T64(1.5)", "excluded=[(1, 2), (2, 3)], special=[(1, 3)]", "diagonal(20)", "use_list=false",
                   "rtol=1e-8, atol=1e-10", "1e-7u\"kJ * mol^-1 * nm^-1\"", "1e-5u\"kJ * mol^-1\"", "1e-10u\"nm\"", "1e-7u\"nm * ps^-1\"", "ROCArray{Int32, 1}", "AMDGPU.functional()", "scale_coords!(gpu, μ)", "gpu.boundary = old_boundary", "MonteCarloBarostat("):
        assert needle in t, needle
