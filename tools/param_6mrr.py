#!/usr/bin/env python3
"""tools/param_6mrr.py — builds molly.jl_amd/data/6mrr_system.npz (the workload's INPUTS: coordinates, velocities, parameters, topology) and
tests/golden/6mrr.npz (the OpenMM Reference-platform OUTPUTS the parity tests compare with) from the reference's data files.  Not product code.

Restates just enough of Molly's setup (src/setup.jl:512-1010, src/residues.jl:190-723, src/force_field.jl:179-290)
to turn data/6mrr_equil.pdb + ff99SBildn.xml + tip3p_standard.xml into flat parameter arrays, and copies the
OpenMM Reference-platform force / energy / trajectory files of data/openmm_6mrr/ into the same archive, so that
the parity tests can run on the GPU box where /root/reference does not exist.

    python tools/param_6mrr.py [/root/reference] [tests/golden/6mrr.npz] [molly.jl_amd/data/6mrr_system.npz]

SURVEY.md Appendix C documents the rules restated here; self-checks at the bottom reproduce the reference's
own assertions (test/protein.jl:141-190, test/basic.jl:592).
"""
import itertools
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "6mrr.npz")
OUT_SYSTEM = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "molly.jl_amd", "data", "6mrr_system.npz")
DATA = os.path.join(REF, "data")


# ---- PDB (setup.jl:585-614: Å → nm, CRYST1 box, center_coords=false as test/protein.jl:90, wrap) --------
def read_pdb(path):
    atoms, box = [], None
    for line in open(path):
        if line.startswith("CRYST1"):
            box = np.array([float(line[6:15]), float(line[15:24]), float(line[24:33])]) / 10.0
        elif line.startswith(("ATOM", "HETATM")):
            atoms.append(dict(name=line[12:16].strip(), res=line[17:20].strip(), chain=line[21], resnum=int(line[22:26]),
                              xyz=np.array([float(line[30:38]), float(line[38:46]), float(line[46:54])]) / 10.0,
                              element=line[76:78].strip()))
    return atoms, box


# ---- force field XML -------------------------------------------------------------------------------------
class FF:
    def __init__(self, files):
        self.types, self.residues = {}, {}
        self.bonds, self.angles, self.propers, self.impropers, self.nb = {}, {}, [], [], {}
        self.c14, self.lj14 = None, None
        for f in files:
            root = ET.parse(f).getroot()
            for t in root.find("AtomTypes"):
                self.types[t.get("name")] = dict(cls=t.get("class"), mass=float(t.get("mass")), element=t.get("element", "?"))
            for r in root.find("Residues"):
                atoms = [(a.get("name"), a.get("type"), float(a.get("charge"))) for a in r.findall("Atom")]
                ext = [e.get("atomName") for e in r.findall("ExternalBond")]
                self.residues[r.get("name")] = dict(atoms=atoms, ext=ext)
            sec = root.find("HarmonicBondForce")
            if sec is not None:
                for b in sec:
                    self.bonds[(b.get("type1"), b.get("type2"))] = (float(b.get("k")), float(b.get("length")))
            sec = root.find("HarmonicAngleForce")
            if sec is not None:
                for a in sec:
                    self.angles[(a.get("type1"), a.get("type2"), a.get("type3"))] = (float(a.get("k")), float(a.get("angle")))
            sec = root.find("PeriodicTorsionForce")
            if sec is not None:
                ordering = sec.get("ordering", "default")
                for t in sec:
                    pats = tuple(t.get(f"type{i}") for i in range(1, 5))
                    terms, n = [], 1
                    while t.get(f"periodicity{n}") is not None:
                        terms.append((int(t.get(f"periodicity{n}")), float(t.get(f"phase{n}")), float(t.get(f"k{n}"))))
                        n += 1
                    rule = dict(pats=pats, terms=terms, wild=any(p == "" for p in pats),
                                spec=sum(2 for p in pats if p != ""), ordering=ordering)   # spec_score force_field.jl:24
                    (self.propers if t.tag == "Proper" else self.impropers).append(rule)
            sec = root.find("NonbondedForce")
            if sec is not None:
                self.c14, self.lj14 = float(sec.get("coulomb14scale")), float(sec.get("lj14scale"))
                for a in sec.findall("Atom"):
                    self.nb[a.get("type")] = (float(a.get("sigma")), float(a.get("epsilon")))

    def bond(self, t1, t2):
        return self.bonds.get((t1, t2)) or self.bonds.get((t2, t1))

    def angle(self, t1, t2, t3):
        return self.angles.get((t1, t2, t3)) or self.angles.get((t3, t2, t1))

    @staticmethod
    def _m(p, t):
        return p == "" or p == t

    def proper(self, t1, t2, t3, t4):
        """find_proper_match (force_field.jl:179-230): forward then reverse; first non-wildcard match wins, else the
        most specific wildcard rule; candidates are ordered central-type first, wildcard-central rules last."""
        best, bestspec = None, -1
        for q in ((t1, t2, t3, t4), (t4, t3, t2, t1)):
            cand = [r for r in self.propers if r["pats"][1] == q[1]] + [r for r in self.propers if r["pats"][1] == ""]
            for r in cand:
                if all(self._m(p, t) for p, t in zip(r["pats"], q)):
                    if not r["wild"]:
                        return r
                    if r["spec"] > bestspec:
                        best, bestspec = r, r["spec"]
        return best

    def improper(self, t1, t2, t3, t4):
        """find_improper_match (force_field.jl:232-290): central atom first, 6 permutations of the peripherals."""
        best, bestperm, bestspec = None, (0, 1, 2, 3), -1
        cand = [r for r in self.impropers if r["pats"][0] == t1] + [r for r in self.impropers if r["pats"][0] == ""]
        ts = (t1, t2, t3, t4)
        for perm in ((0, 1, 2, 3), (0, 1, 3, 2), (0, 2, 1, 3), (0, 2, 3, 1), (0, 3, 1, 2), (0, 3, 2, 1)):
            p2, p3, p4 = ts[perm[1]], ts[perm[2]], ts[perm[3]]
            for r in cand:
                if not self._m(r["pats"][0], t1):
                    continue
                if self._m(r["pats"][1], p2) and self._m(r["pats"][2], p3) and self._m(r["pats"][3], p4):
                    if not r["wild"]:
                        return r, perm
                    if r["spec"] > bestspec:
                        best, bestperm, bestspec = r, perm, r["spec"]
        return best, bestperm


def standard_bonds(path):
    out = {}
    for r in ET.parse(path).getroot():
        out[r.get("name")] = [(b.get("from"), b.get("to")) for b in r.findall("Bond")]
    return out


def main():
    atoms, box = read_pdb(os.path.join(DATA, "6mrr_equil.pdb"))
    n = len(atoms)
    ff = FF([os.path.join(DATA, "force_fields", "ff99SBildn.xml"), os.path.join(DATA, "force_fields", "tip3p_standard.xml")])
    std = standard_bonds(os.path.join(DATA, "force_fields", "residues.xml"))
    coords = np.array([a["xyz"] for a in atoms])
    coords = coords - np.floor(coords / box) * box                       # wrap_coords, setup.jl:614

    # residues in file order
    residues = []
    for i, a in enumerate(atoms):
        key = (a["chain"], a["resnum"], a["res"])
        if not residues or residues[-1]["key"] != key:
            residues.append(dict(key=key, name=a["res"], idx=[], names=[]))
        residues[-1]["idx"].append(i); residues[-1]["names"].append(a["name"])
    protein = [r for r in residues if r["name"] != "HOH"]
    res_of = np.zeros(n, int); pos_in_res = np.zeros(n, int)
    for ri, r in enumerate(residues):
        for k, i in enumerate(r["idx"]):
            res_of[i], pos_in_res[i] = ri, k

    # bonds from the standard-bond table (residues.jl:190-260), inter-residue "-C"–"N" inside the protein chain
    bonds = set()
    for ri, r in enumerate(residues):
        amap = dict(zip(r["names"], r["idx"]))
        for a1, a2 in std.get(r["name"], []):
            def resolve(nm):
                if nm.startswith("-"):
                    if ri == 0 or residues[ri - 1]["name"] == "HOH" or r["name"] == "HOH":
                        return None
                    return dict(zip(residues[ri - 1]["names"], residues[ri - 1]["idx"])).get(nm[1:])
                if nm.startswith("+"):
                    return None   # every "+X" bond is also listed as "-X" on the next residue
                if nm == "H" and "H" not in amap:      # pdbNames.xml alias: the first N-terminal hydrogen H1 ≡ H (residues.jl:105)
                    return amap.get("H1")
                return amap.get(nm)
            i, j = resolve(a1), resolve(a2)
            if i is not None and j is not None:
                bonds.add((min(i, j), max(i, j)))
    bonds = sorted(bonds)
    adj = [[] for _ in range(n)]
    for i, j in bonds:
        adj[i].append(j); adj[j].append(i)
    adj = [sorted(set(a)) for a in adj]
    n_ext = np.zeros(n, int)
    for i, j in bonds:
        if res_of[i] != res_of[j]:
            n_ext[i] += 1; n_ext[j] += 1

    # template matching by residue name, atom names and external-bond pattern (setup.jl:633-690 matches graphs;
    # for this OpenMM-written PDB the names are canonical, so name matching selects the same template)
    atype, charge = [None] * n, np.zeros(n)
    for r in residues:
        names = set(r["names"])
        cands = [r["name"], "N" + r["name"], "C" + r["name"]]
        if r["name"] == "HIS":
            cands = ["HID", "HIE", "HIP", "NHID", "NHIE", "NHIP", "CHID", "CHIE", "CHIP"]
        hit = None
        for c in cands:
            t = ff.residues.get(c)
            if t is not None and {a[0] for a in t["atoms"]} == names:
                ext_t = sorted(t["ext"])
                ext_r = sorted(nm for nm, i in zip(r["names"], r["idx"]) for _ in range(n_ext[i]))
                if ext_t == ext_r:
                    hit = t; break
        if hit is None:
            raise SystemExit(f"no template for residue {r['key']}")
        tmap = {a[0]: a for a in hit["atoms"]}
        for nm, i in zip(r["names"], r["idx"]):
            atype[i], charge[i] = tmap[nm][1], tmap[nm][2]
    sigma = np.array([ff.nb[t][0] for t in atype]); eps = np.array([ff.nb[t][1] for t in atype])
    mass = np.array([ff.types[t]["mass"] for t in atype])
    element = [ff.types[t]["element"] for t in atype]

    # angles / torsions / impropers from the adjacency (residues.jl:618-723); 0-based here
    angles = set()
    for b1, b2 in bonds:
        for a in adj[b1]:
            if a != b2:
                angles.add((a, b1, b2) if a < b2 else (b2, b1, a))
        for a in adj[b2]:
            if a != b1:
                angles.add((b1, b2, a) if a > b1 else (a, b2, b1))
    angles = sorted(angles)
    tors = set()
    for a1, a2, a3 in angles:
        for a in adj[a1]:
            if a not in (a1, a2, a3):
                tors.add((a, a1, a2, a3) if a < a3 else (a3, a2, a1, a))
        for a in adj[a3]:
            if a not in (a1, a2, a3):
                tors.add((a1, a2, a3, a) if a > a1 else (a, a3, a2, a1))
    tors = sorted(tors)
    imps = [(i, *sub) for i, nb in enumerate(adj) if len(nb) > 2 for sub in itertools.combinations(nb, 3)]

    excluded, special = set(), set()
    b_i, b_j, b_k, b_r0 = [], [], [], []
    for i, j in bonds:
        k, r0 = ff.bond(atype[i], atype[j])
        b_i.append(i); b_j.append(j); b_k.append(k); b_r0.append(r0)
        excluded.add((i, j))                                                              # setup.jl:787-788
    a_i, a_j, a_k, a_kth, a_th0 = [], [], [], [], []
    for i, j, k in angles:
        kth, th0 = ff.angle(atype[i], atype[j], atype[k])
        a_i.append(i); a_j.append(j); a_k.append(k); a_kth.append(kth); a_th0.append(th0)
        excluded.add((min(i, k), max(i, k)))                                              # setup.jl:804-805
    pt = {k: [] for k in ("i", "j", "k", "l", "periodicity", "phase", "k0")}
    for i, j, k, l in tors:
        r = ff.proper(atype[i], atype[j], atype[k], atype[l])
        if r is None:
            continue
        for per, ph, k0 in r["terms"]:
            for key, v in zip(pt, (i, j, k, l, per, ph, k0)):
                pt[key].append(v)
        special.add((min(i, l), max(i, l)))                                               # setup.jl:853-854
    it = {k: [] for k in ("i", "j", "k", "l", "periodicity", "phase", "k0")}
    for c, j, k, l in imps:
        r, perm = ff.improper(atype[c], atype[j], atype[k], atype[l])
        if r is None:
            continue
        src = (c, j, k, l)
        j, k, l = src[perm[1]], src[perm[2]], src[perm[3]]                                 # setup.jl:876-881
        t2, t3, t4 = atype[j], atype[k], atype[l]
        key = lambda a: (res_of[a], pos_in_res[a])
        if r["ordering"] == "amber":                                                       # setup.jl:904-934
            if not r["wild"]:
                if t2 == t4 and key(j) > key(l):
                    j, l = l, j
                if t3 == t4 and key(k) > key(l):
                    k, l = l, k
                if t2 == t3 and key(j) > key(k):
                    j, k = k, j
            else:
                if element[j] == element[l] and key(j) > key(l):
                    j, l = l, j
                if element[k] == element[l] and key(k) > key(l):
                    k, l = l, k
                if key(j) > key(k):
                    j, k = k, j
        for per, ph, k0 in r["terms"]:
            for kk, v in zip(it, (j, k, c, l, per, ph, k0)):                               # setup.jl:1000-1003: (j, k, c, l)
                it[kk].append(v)

    excl = np.array(sorted(excluded), dtype=np.int32)
    spec = np.array(sorted(special), dtype=np.int32)
    ewx = np.array(sorted(excluded | special), dtype=np.int32)                             # find_excluded_pairs ewald.jl:946-961

    # ---- self-checks against the reference's own assertions ---------------------------------------------------
    assert n == 15954 and len(protein) == 68                                               # test/protein.jl:151
    assert abs(charge.sum()) < 1e-6 and abs(charge[1] - 0.1642) < 1e-12                      # test/protein.jl:141-144
    assert len(excl) == 18096, len(excl)
    print(f"atoms {n}, bonds {len(bonds)}, angles {len(angles)}, proper terms {len(pt['i'])}, improper terms {len(it['i'])}, "
          f"excluded {len(excl)}, special {len(spec)} (special∧¬excluded {len(special - excluded)})")

    out = dict(coords=coords, box=box, charge=charge, sigma=sigma, eps=eps, mass=mass,
               bonds_i=np.array(b_i, np.int32), bonds_j=np.array(b_j, np.int32), bonds_k=np.array(b_k), bonds_r0=np.array(b_r0),
               angles_i=np.array(a_i, np.int32), angles_j=np.array(a_j, np.int32), angles_k=np.array(a_k, np.int32),
               angles_kth=np.array(a_kth), angles_th0=np.array(a_th0), excluded=excl, special=spec, ewald_excl=ewx,
               weight_14_coulomb=np.float64(ff.c14), weight_14_lj=np.float64(ff.lj14))
    for name, d in (("proper", pt), ("improper", it)):
        for k, v in d.items():
            out[f"{name}_{k}"] = np.array(v, np.int32 if k in ("i", "j", "k", "l", "periodicity") else np.float64)
    omm = os.path.join(DATA, "openmm_6mrr")
    for inter in ("lj_only", "coul_only", "bond_only", "angle_only", "proptor_only", "improptor_only", "all_cut", "all_pme", "all_pme_exact"):
        out[f"openmm_forces_{inter}"] = np.loadtxt(os.path.join(omm, "amber", f"forces_{inter}.txt")).astype(np.float64)
        out[f"openmm_energy_{inter}"] = np.float64(np.loadtxt(os.path.join(omm, "amber", f"energy_{inter}.txt")))
    out["velocities_300K"] = np.loadtxt(os.path.join(omm, "velocities_300K.txt"))
    out["openmm_coordinates_100steps"] = np.loadtxt(os.path.join(omm, "amber", "coordinates_100steps.txt"))
    out["openmm_velocities_100steps"] = np.loadtxt(os.path.join(omm, "amber", "velocities_100steps.txt"))
    golden = {k: v for k, v in out.items() if k.startswith("openmm_")}
    system = {k: v for k, v in out.items() if not k.startswith("openmm_")}
    for path, arrays in ((OUT, golden), (OUT_SYSTEM, system)):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez_compressed(path, **arrays)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
