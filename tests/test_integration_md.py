"""INTEGRATION.md is the reference-side binding (Julia `ccall`s); Julia is not installed here, so nothing executes it.  This guard
keeps the text from drifting: every `ccall((:mhip_…, libmollyhip), Ret, (ArgTypes…), …)` of the file is checked against the
prototype of include/mollyhip.h — the symbol exists, the argument count matches, every Julia argument type is compatible with
the C parameter type, the return type matches — and the `struct Mhip…` definitions mirror the C structs field by field."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def test_every_ccall_matches_the_header():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
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
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
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
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
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
        assert len(heads) == 1, (fname, len(heads))
        pos, kws = heads[0]
        assert len(pos) == len(ref["positional"]), (fname, pos, ref["positional"])
        assert not kws and not ref["keywords"]
        for (n_i, t_i), (n_r, t_r) in zip(pos, ref["positional"]):
            assert n_i == n_r, (fname, n_i, n_r)                                   # same names in the same order (anonymous ::Val{…} included)
            if n_r == "sys":
                assert t_r.replace(" ", "").startswith("System{") and "<:CuArray" in t_r and "<:ROCArray" in t_i, (fname, t_i, t_r)
            elif t_r in ("Nothing", "Val{needs_vir}"):
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
