"""T2 parity anchor: the softcore sterics / direct-space electrostatics / exception energies
of openmmtools' AbsoluteAlchemicalFactory, written as the Lepton strings that factory hands to OpenMM
(tests/golden/openmmtools_lepton_strings.json, each marked [recalled]), evaluated here by plain Python -- NOT by
blues_amd/lepton.py -- over a grid of (r, lambda_sterics, lambda_electrostatics, sigma, epsilon, chargeprod), against the
oracle's pair energies and against a central finite difference for the oracle's gradients.  This pins the oracle (and with it
every GPU parity test) to a TEXT a maintainer can diff against openmmtools 0.15.0, instead of to the builder's derivation.
Reference call site: blues/simulation.py:225-236, 300-316."""
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def evaluate(expression, variables):
    """Lepton surface syntax by plain Python: the first expression is the value, later `name = expr` definitions feed earlier ones."""
    parts = [p.strip() for p in expression.split(";") if p.strip()]
    env = {"sqrt": math.sqrt, "erfc": math.erfc, "erf": math.erf, "exp": math.exp, "step": lambda x: 0.0 if x < 0 else 1.0, "abs": abs, "min": min, "max": max}
    env.update(variables)
    for p in reversed(parts[1:]):
        name, _, rhs = p.partition("=")
        env[name.strip()] = eval(rhs.strip().replace("^", "**"), {"__builtins__": {}}, env)
    return eval(parts[0].replace("^", "**"), {"__builtins__": {}}, env)


@pytest.fixture(scope="module")
def strings():
    with open(os.path.join(HERE, "golden", "openmmtools_lepton_strings.json")) as fh:
        d = json.load(fh)
    assert all("[recalled]" in k for k in d["strings"] if k != "sterics_mixing_rules")     # the status is part of the fixture
    return d


GRID_R = (0.09, 0.15, 0.25, 0.33, 0.5, 0.77, 0.99)
GRID_L = (0.0, 1e-3, 0.25, 0.5, 0.9, 1.0)
PARAMS = ((0.3399669508423535, 0.359824, 0.0), (0.315075, 0.635968, -0.834 * 0.13), (0.25, 0.0657, 0.417 * -0.1191), (0.1, 0.0, 0.3))   # (sigma, epsilon, q1*q2)
ALPHA = 2.145966026289347


def test_softcore_sterics_and_direct_space_electrostatics(oracle_mod, strings):
    g = dict(strings["globals"])
    worst = 0.0
    for sigma, eps, qq in PARAMS:
        for r in GRID_R:
            for ls in GRID_L:
                for le in GRID_L:
                    v = dict(g, r=r, sigma=sigma, epsilon=eps, chargeprod=qq, lambda_sterics=ls, lambda_electrostatics=le, alpha_ewald=ALPHA)
                    want = evaluate(strings["strings"]["sterics_softcore [recalled]"], v) + evaluate(strings["strings"]["electrostatics_direct_space_pme [recalled]"], v)
                    got, dEdr = oracle_mod.pair_energy(r, qq, sigma, eps, alpha_ewald=ALPHA, alchemical=True, lambda_s=ls, lambda_e=le, softcore_alpha=g["softcore_alpha"])
                    assert got == pytest.approx(want, rel=1e-12, abs=1e-12), (sigma, eps, qq, r, ls, le)
                    h = 1e-6 * r

                    def U(rr):
                        w = dict(v, r=rr)
                        return evaluate(strings["strings"]["sterics_softcore [recalled]"], w) + evaluate(strings["strings"]["electrostatics_direct_space_pme [recalled]"], w)
                    fd = (U(r + h) - U(r - h)) / (2 * h)
                    assert dEdr == pytest.approx(fd, rel=2e-6, abs=1e-6 * max(1.0, abs(want)))
                    worst = max(worst, abs(got - want) / max(1.0, abs(want)))
    assert worst < 1e-12


def test_mixing_rules_and_exception_form(oracle_mod, strings):
    g = dict(strings["globals"])
    # Lorentz-Berthelot as the CustomNonbondedForce states it
    mix = strings["strings"]["sterics_mixing_rules"]
    env = {}
    for p in [q.strip() for q in mix.split(";") if q.strip()]:
        name, _, rhs = p.partition("=")
        env[name.strip()] = eval(rhs.strip(), {"__builtins__": {}}, {"sqrt": math.sqrt, "sigma1": 0.34, "sigma2": 0.315075, "epsilon1": 0.36, "epsilon2": 0.635968})
    assert env["sigma"] == pytest.approx(0.5 * (0.34 + 0.315075)) and env["epsilon"] == pytest.approx(math.sqrt(0.36 * 0.635968))
    # exceptions: softcore LJ + bare Coulomb (no erfc), both scaled -- alpha_ewald = 0 selects the unscreened form in the oracle
    for r in GRID_R:
        for ls in (0.0, 0.3, 1.0):
            for le in (0.0, 0.6, 1.0):
                v = dict(g, r=r, sigma=0.3, epsilon=0.2, chargeprod=0.05, lambda_sterics=ls, lambda_electrostatics=le)
                want = evaluate(strings["strings"]["sterics_softcore [recalled]"], v) + evaluate(strings["strings"]["electrostatics_exception [recalled]"], v)
                got, _ = oracle_mod.pair_energy(r, 0.05, 0.3, 0.2, alpha_ewald=0.0, alchemical=True, lambda_s=ls, lambda_e=le, softcore_alpha=g["softcore_alpha"])
                assert got == pytest.approx(want, rel=1e-12, abs=1e-12)
    # lambda = 1 is the unmodified pair (12-6 LJ + erfc Coulomb): what openmmtools must reproduce for the factory to be an identity there
    for r in GRID_R:
        v = dict(g, r=r, sigma=0.33, epsilon=0.4, chargeprod=-0.2, lambda_sterics=1.0, lambda_electrostatics=1.0, alpha_ewald=ALPHA)
        soft = evaluate(strings["strings"]["sterics_softcore [recalled]"], v) + evaluate(strings["strings"]["electrostatics_direct_space_pme [recalled]"], v)
        plain = 4 * 0.4 * ((0.33 / r) ** 12 - (0.33 / r) ** 6) + 138.935456 * -0.2 * math.erfc(ALPHA * r) / r
        assert soft == pytest.approx(plain, rel=1e-12)
        got, _ = oracle_mod.pair_energy(r, -0.2, 0.33, 0.4, alpha_ewald=ALPHA, alchemical=False)
        assert got == pytest.approx(plain, rel=1e-12)
