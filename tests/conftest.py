import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_guards_seen = [0]


@pytest.fixture(autouse=True)
def _device_buffer_guards(request):
    """After every GPU test: no kernel may have written past the end of a device buffer (the engines' buffers come from slabs with 256
    unowned bytes behind each block: include/blues_engine.h, blues_debug_check_guards)."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import ctypes
    from blues_amd import _lib
    lib = _lib.load()
    out = (ctypes.c_int64 * 8)()
    assert lib.blues_debug_check_guards(out) == 0
    marked = int(out[1])
    new, _guards_seen[0] = marked - _guards_seen[0], marked
    assert new <= 0, "a kernel wrote past the end of a device buffer during this test: %d guard(s) marked; first blocks (bytes, offset in guard): %s" % (new, [int(v) for v in out[2:8]])


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def tol_box():
    from blues_amd import systems
    return systems.toluene_box()


@pytest.fixture(scope="session")
def known_answers():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as fh:
        return json.load(fh)


class OracleBackedEngine:
    """TEST DOUBLE with NativeEngine's interface on top of the CPU oracle, so the host-side driver
    (context.py / simulation.py / moves.py) can be exercised without a GPU.  Never used by the product."""

    def __init__(self, system, integrator_data, device=0):
        from oracle import oracle
        self._o = oracle.Oracle(system, integrator_data)
        ex = getattr(system, "extras", None) or {}
        if ex.get("custom_pair_mode"):
            self._o.set_custom_pair_mode(ex["custom_pair_mode"])
        for cb in ex.get("centroid_bonds", []):
            self._o.add_centroid_bond(*cb)
        self.n = system.n_atoms
        self.system, self.integrator = system, integrator_data
        self._box = np.diag(np.asarray(system.box, dtype=float).reshape(-1)[:3]) if np.size(system.box) == 3 else np.asarray(system.box).reshape(3, 3)
        self._unpert = None
        self._edited = False

    def set_positions(self, x):
        o = self._o
        if 0 < o.get_global("step") < o.get_global("nsteps"):
            pass  # the oracle tracks unperturbed_pe itself
        o.set_positions(x)

    def set_velocities(self, v): self._o.set_velocities(v)
    def get_positions(self): return self._o.get_positions()
    def get_velocities(self): return self._o.get_velocities()
    def get_forces(self): return self._o.get_forces()
    def get_box(self): return self._box.copy()
    def set_box(self, b): self._box = np.diag(np.asarray(b, dtype=float).reshape(-1)[:3]) if np.size(b) == 3 else np.asarray(b).reshape(3, 3); self._o.set_box(np.diag(self._box))
    def set_velocities_to_temperature(self, T, seed=0): self._o.set_velocities_to_temperature(T, seed)
    def potential_energy(self): return self._o.potential_energy()
    def kinetic_energy(self): return self._o.kinetic_energy()
    def step(self, n=1): self._o.step(n)
    def run_switch(self, n, trace=False):
        w = []
        for _ in range(int(n)):
            self._o.step(1); w.append(self._o.get_global("protocol_work"))
        return np.array(w) if trace else None
    def get_global(self, name): return self._o.get_global(name)
    def set_global(self, name, value): self._o.set_global(name, value)
    def reset(self): self._o.reset()
    def close(self): pass


@pytest.fixture()
def oracle_backed_context(monkeypatch):
    """Routes blues_amd.context.NativeEngine to the oracle-backed test double."""
    from blues_amd import context
    monkeypatch.setattr(context, "NativeEngine", OracleBackedEngine)
    return context


@pytest.fixture()
def tune():
    """tune(field=value, ...) pins launch-policy fields of the engine (blues_amd.tuning, include/blues_engine.h: BluesTuning) for the
    engines and batches the test creates afterwards; the defaults come back when the test ends."""
    from blues_amd import tuning
    saved = tuning.current()
    yield tuning.set
    tuning._apply(saved)


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
