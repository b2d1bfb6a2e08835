#!/usr/bin/env python3
"""Generates the committed data fixtures.  Run in the BUILD container only
(it reads the reference checkout at /root/reference; the GPU box never does):

    python tests/golden/make_fixtures.py

Outputs
  blues_amd/data/tol_box.npz
      975-atom toluene/TIP3P box as a flat SystemData, derived from the data files the
      reference's own tests hold (blues/tests/data/TOL-parm.prmtop / .inpcrd): parameters
      converted to OpenMM units, TIP3P assigned to the water atoms (their ATOM_TYPE_INDEX
      is 0 in that file, TOL-parm.prmtop:556-560), HBonds + rigid water constraints,
      hydrogen mass 3.024 Da; coordinates relaxed by steepest descent and thermalised
      with the CPU oracle's Langevin integrator (seeded), velocities included.
  tests/golden/reference_known_answers.json
      values the reference's own files state for this path.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/blues/tests/data"

from blues_amd import amber, systems  # noqa: E402
from blues_amd._abi import IntegratorData  # noqa: E402
from oracle import oracle  # noqa: E402


def make_tol_box():
    prm = amber.read_prmtop(os.path.join(REF, "TOL-parm.prmtop"))
    pos, _, box = amber.read_inpcrd(os.path.join(REF, "TOL-parm.inpcrd"))
    s = amber.system_from_amber(prm, pos, box, cutoff=1.0, ewald_error_tolerance=0.005, constraints="HBonds",
                                rigid_water=True, hydrogen_mass=3.024, remove_cm_motion=True, reciprocal_space=False,
                                alchemical_atoms=range(15))
    print("atoms", s.n_atoms, "box", s.box, "constraints", len(s.constraint_dist), "bonds", len(s.bond_atoms),
          "angles", len(s.angle_atoms), "torsions", len(s.torsion_atoms), "excl", len(s.exclusions), "exc14", len(s.exception_atoms))
    print("net charge", s.charge.sum())

    def langevin(nsteps, dt, gamma, seed):
        return IntegratorData(timestep=dt, temperature=300.0, nsteps_neq=10 ** 9, lambda_sterics=[1.0],
                              lambda_electrostatics=[1.0], splitting="V R O R V", collision_rate=gamma, seed=seed)

    o = oracle.Oracle(s, langevin(0, 0.001, 10.0, 11))
    e0 = o.energy_forces()[0]
    e1 = o.minimize(max_iter=600, step0=0.005)
    print("energy before/after minimisation", e0, e1)
    x = o.get_positions()
    # thermalise: 2 ps at 1 fs with strong friction, then 4 ps at 2 fs, then 4 ps at 4 fs (gamma 1/ps)
    v = np.zeros_like(x)
    for (nsteps, dt, gamma, seed) in ((2000, 0.001, 10.0, 11), (2000, 0.002, 5.0, 12), (1000, 0.004, 1.0, 13)):
        o = oracle.Oracle(s, langevin(nsteps, dt, gamma, seed))
        o.set_positions(x)
        if not v.any():
            o.set_velocities_to_temperature(300.0, 2017)
        else:
            o.set_velocities(v)
        for blk in range(nsteps // 500):
            o.step(500)
            ke = o.kinetic_energy()
            ndof = 3 * s.n_atoms - len(s.constraint_dist)
            print("  dt", dt, "step", (blk + 1) * 500, "PE", o.potential_energy(), "T", 2 * ke / (ndof * 0.0083144626))
        x, v = o.get_positions(), o.get_velocities()
    s.positions = x
    out = os.path.join(ROOT, "blues_amd", "data", "tol_box.npz")
    systems.save_system(out, s, velocities=v)
    print("wrote", out, os.path.getsize(out), "bytes")


def make_known_answers():
    """Values stated by the reference's own sources (file:line) for the host logic of this path."""
    ka = {
        "_source": "MobleyLab/blues; see SURVEY.md Appendix C and section 4 for the file:line of each entry",
        # blues/utils.py:89-145, hand-executed on the reference formulas
        "calculateNCMCSteps": [
            {"nstepsNC": 10, "nprop": 1, "propLambda": 0.3, "expect": [10, 10, 5]},
            {"nstepsNC": 20, "nprop": 1, "propLambda": 0.3, "expect": [20, 20, 10]},
            {"nstepsNC": 100, "nprop": 1, "propLambda": 0.3, "expect": [100, 100, 50]},
            {"nstepsNC": 1000, "nprop": 1, "propLambda": 0.3, "expect": [1000, 1000, 500]},
            {"nstepsNC": 2000, "nprop": 1, "propLambda": 0.3, "expect": [2000, 2000, 1000]},
            {"nstepsNC": 5000, "nprop": 1, "propLambda": 0.3, "expect": [5000, 5000, 2500]},
            {"nstepsNC": 10000, "nprop": 1, "propLambda": 0.3, "expect": [10000, 10000, 5000]},
            {"nstepsNC": 1000, "nprop": 2, "propLambda": 0.3, "expect": [1000, 1000, 500]},
            {"nstepsNC": 3000, "nprop": 3, "propLambda": 0.3, "expect": [3000, 3000, 1500]},
            {"nstepsNC": 100, "nprop": 2, "propLambda": 0.1, "expect": [100, 100, 50]},
            {"nstepsNC": 101, "nprop": 1, "propLambda": 0.3, "expect": [100, 100, 50]},
        ],
        # blues/integrators.py:147-157; blues/tests/test_simulation.py:286-288 expects (0.4, 0.6) for 0.1
        "get_prop_lambda": [
            {"prop_lambda": 0.3, "expect": [0.2, 0.8]}, {"prop_lambda": 0.1, "expect": [0.4, 0.6]},
            {"prop_lambda": 0.0, "expect": [2.0, -1.0]}, {"prop_lambda": -0.2, "expect": [2.0, -1.0]},
        ],
        # default schedules, blues/simulation.py:654-659
        "lambda_table": {
            "lambda": [0.0, 0.1, 0.2, 0.35, 0.5, 0.65, 0.8, 0.9, 1.0],
            "lambda_sterics": [1.0, 1.0, 1.0, 0.5, 0.0, 0.5, 1.0, 1.0, 1.0],
            "lambda_electrostatics": [1.0, 0.5, 0.0, 0.0, 0.0, 0.0, 0.0, 0.5, 1.0],
        },
        # blues/tests/test_simulation.py:262-289 (integrator attributes for the given cfg)
        "integrator_attributes": {
            "cfg": {"nstepsNC": 100, "temperature": 100.0, "dt": 0.001, "nprop": 2, "propLambda": 0.1,
                    "splitting": "V H R O R H V",
                    "alchemical_functions": {"lambda_sterics": "1", "lambda_electrostatics": "1"}},
            "expect": {"_n_steps_neq": 100, "_n_lambda_steps": 200, "_prop_lambda": [0.4, 0.6],
                       "_splitting": "V H R O R H V"},
        },
        # unit constants
        "kT_300K": 2.494339, "kT_200K": 1.662893,
        "ewald_alpha": [{"tol": 0.005, "cutoff": 1.0, "alpha": 2.145966}, {"tol": 0.0005, "cutoff": 1.0, "alpha": 2.628261}],
        # blues/tests/test_watertranslation.py:106,112 ; blues/moves.py:1082
        "protocol_work_before_first_step": 0.0, "afterMove_rejection_work": 999999,
        # blues/tests/test_ethylene.py:144-163
        "ethylene_populations": [0.25, 0.75],
        # TOL-parm.prmtop:7-10 POINTERS
        "tol_parm_pointers": {"NATOM": 975, "NTYPES": 4, "NBONH": 648, "MBONA": 7, "NTHETH": 336, "MTHETA": 8,
                              "NPHIH": 27, "NPHIA": 9, "NNB": 1348, "NRES": 321},
        # ethylene_system.xml:27 cross-check of the Amber->OpenMM bond constant conversion
        "bond_k_conversion": {"amber_k": 569.4, "openmm_k": 476473.92},
    }
    out = os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")
    with open(out, "w") as fh:
        json.dump(ka, fh, indent=1)
    print("wrote", out)


def make_ethylene_fixture():
    """Flat JSON form of the data files behind the reference's only result-pinning test of this path
    (blues/tests/data/ethylene_system.xml, ethylene_structure.pdb; blues/tests/test_ethylene.py)."""
    import xml.etree.ElementTree as ET
    root = ET.parse(os.path.join(REF, "ethylene_system.xml")).getroot()
    box = [float(root.find("PeriodicBoxVectors").find(t).get(a)) for t, a in (("A", "x"), ("B", "y"), ("C", "z"))]
    masses = [float(p.get("mass")) for p in root.find("Particles")]
    cons = [[int(c.get("p1")), int(c.get("p2")), float(c.get("d"))] for c in root.find("Constraints")]
    out = {"_source": "blues/tests/data/ethylene_system.xml + ethylene_structure.pdb (reference test fixtures)",
           "box": box, "masses": masses, "constraints": cons}
    for f in root.find("Forces"):
        t = f.get("type")
        if t == "HarmonicBondForce":
            out["bonds"] = [[int(b.get("p1")), int(b.get("p2")), float(b.get("d")), float(b.get("k"))] for b in f.find("Bonds")]
        elif t == "HarmonicAngleForce":
            out["angles"] = [[int(b.get("p1")), int(b.get("p2")), int(b.get("p3")), float(b.get("a")), float(b.get("k"))] for b in f.find("Angles")]
        elif t == "PeriodicTorsionForce":
            out["torsions"] = [[int(b.get("p1")), int(b.get("p2")), int(b.get("p3")), int(b.get("p4")), int(b.get("periodicity")), float(b.get("phase")), float(b.get("k"))] for b in f.find("Torsions")]
        elif t == "CustomNonbondedForce":
            out["custom_nonbonded"] = {"energy": f.get("energy"), "method": int(f.get("method")),
                                       "particles": [[float(p.get("param1")), float(p.get("param2")), float(p.get("param3")), float(p.get("param4"))] for p in f.find("Particles")],
                                       "set1": [int(p.get("index")) for p in f.find("InteractionGroups")[0].find("Set1")],
                                       "set2": [int(p.get("index")) for p in f.find("InteractionGroups")[0].find("Set2")]}
        elif t == "CustomCentroidBondForce":
            groups = [[[int(p.get("p")), p.get("weight")] for p in g] for g in f.find("Groups")]
            out["centroid_bond"] = {"energy": f.get("energy"), "groups": groups, "k": float(f.find("Bonds")[0].get("param1"))}
    pos = []
    for line in open(os.path.join(REF, "ethylene_structure.pdb")):
        if line.startswith(("ATOM", "HETATM")):
            pos.append([float(line[30:38]) * 0.1, float(line[38:46]) * 0.1, float(line[46:54]) * 0.1])
    out["positions_nm"] = pos
    # parameters of the test itself, blues/tests/test_ethylene.py:31-43, 79, 107-115, 152
    out["test"] = {"temperature": 200.0, "dt": 0.001, "friction": 1.0, "nIter": 100, "nstepsMD": 20, "nstepsNC": 20, "moveStep": 10,
                   "repeats": 5, "reportInterval": 5, "alchemical_atoms": [2, 3, 4, 5, 6, 7], "splitting": "H V R O R V H",
                   "distance_atoms": [0, 2], "distance_cut_nm": 0.49, "populations": [0.25, 0.75], "md_reset_temperature": 300.0}
    path = os.path.join(ROOT, "tests", "golden", "ethylene_system.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)


def make_oracle_vectors():
    """Regression vectors: oracle outputs on the committed toluene box (inputs = blues_amd/data/tol_box.npz).
    They are NOT reference values (the reference has none); they freeze the oracle so that
    a silent change of the checker itself is caught, and give the GPU suite committed numbers to hit."""
    s, v = systems.toluene_box()
    from blues_amd import integrators
    data = integrators.generateNCMCIntegrator(nstepsNC=20, dt=0.004, temperature=300.0, seed=7).to_data()
    o = oracle.Oracle(s, data); o.set_velocities(v)
    out = {"_note": "oracle (fp64 CPU restatement) outputs on blues_amd/data/tol_box.npz; generated by tests/golden/make_fixtures.py",
           "integrator": {"nstepsNC": 20, "dt": 0.004, "temperature": 300.0, "seed": 7}, "energies": [], "work_trace": []}
    sel = [0, 3, 7, 14, 15, 16, 500, 974]
    for (ls, le) in ((1.0, 1.0), (0.5, 0.3), (0.05, 0.0), (0.0, 0.0)):
        e, f, t = o.energy_forces(ls, le)
        out["energies"].append({"lambda_sterics": ls, "lambda_electrostatics": le, "total": e, "terms": t.tolist(),
                                "force_atoms": sel, "forces": f[sel].tolist(), "force_norm": float(np.linalg.norm(f))})
    for k in range(20):
        o.step(1); out["work_trace"].append(o.get_global("protocol_work"))
    x = o.get_positions()
    out["final_positions_atoms"] = sel; out["final_positions"] = x[sel].tolist()
    path = os.path.join(ROOT, "tests", "golden", "tol_box_oracle_vectors.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    make_known_answers()
    make_ethylene_fixture()
    if "--vectors-only" not in sys.argv:
        make_tol_box()
    make_oracle_vectors()
