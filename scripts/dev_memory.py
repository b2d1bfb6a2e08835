"""Device memory per engine kind (torch.cuda.mem_get_info before / after): where a rank's HBM goes."""
import copy, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blues_amd import build, integrators, systems, tuning
build.build_engine()
from blues_amd.engine import NativeEngine, NativeBatch
def used(): torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 2.0 ** 20
tuning.set(assume_batch=16)
s, v = systems.s23k(mobile_atoms=275, frozen=True)
md_sys = copy.copy(systems.s23k(frozen=False)[0]); md_sys.alchemical_atoms = np.zeros(0, np.int32)
u0 = used()
m = NativeEngine(md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=2).to_data(precision=0)); m.set_positions(s.positions); m.set_velocities(v)
print("MD engine created: %.0f MiB" % (used() - u0)); u1 = used()
m.potential_energy(); print("  after an energy: +%.0f MiB" % (used() - u1)); u1 = used()
m.step(200); print("  after 200 steps: +%.0f MiB" % (used() - u1)); u1 = used()
m.step(1000); print("  after 1000 more steps: +%.0f MiB, resorts %d" % (used() - u1, m.stats()["resorts"])); u1 = used()
g = NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=1000, dt=0.004, temperature=300.0, seed=1).to_data(precision=0)); g.set_velocities(v)
print("NCMC engine created: %.0f MiB" % (used() - u1)); u1 = used()
g.step(100); print("  after 100 steps: +%.0f MiB mode %d" % (used() - u1, g.stats()["nonbonded_kernel"])); u1 = used()
g.reset(); g.set_positions(m.get_positions()); g.potential_energy(); print("  after the hand-over of an MD state: +%.0f MiB mode %d" % (used() - u1, g.stats()["nonbonded_kernel"])); u1 = used()
sn = [m.snapshot() for _ in range(4)]; print("4 snapshots: +%.0f MiB" % (used() - u1)); u1 = used()
