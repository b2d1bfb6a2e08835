"""MonteCarloBarostat of the MD leg (reference blues/simulation.py:603-626: SimulationFactory.addBarostat puts an
openmm.MonteCarloBarostat(pressure, temperature, frequency) into the MD System; the tutorial run behind BASELINE.md is NPT).

The move is OpenMM's [recalled: OpenMM 7.4.2 MonteCarloBarostatImpl / ReferenceMonteCarloBarostat -- not under /root/reference]:
every `frequency` steps, before the step, propose dV uniform in [-volumeScale, +volumeScale]; scale the box isotropically and move
every MOLECULE (connected component of bonds + constraints) rigidly so that its centre scales with the box; accept with
min(1, exp(-w / kT)), w = dU + P dV - N_mol kT ln(V'/V); on rejection restore box and coordinates.  volumeScale starts at 1 % of
the volume and is retuned every 10 attempts to keep the acceptance between 25 % and 75 %.

Host-proposed, device-evaluated: the two potential energies come from the engine (blues_get_energy after blues_set_box /
blues_set_positions re-tile the system for the new box); everything else is a handful of host operations every 25 steps.
Lone chains and the members of a replica batch alike (round 4): the attempt is made where the next chunk of steps is planned
(context.Simulation._plan_chunk), every member of a batch keeps its own box in the batch's argument records.
"""
import numpy as np

KB = 0.0083144626             # kJ/mol/K
BAR_NM3 = 0.0602214076        # 1 bar * 1 nm^3 in kJ/mol  (Avogadro * 1e-25)


def molecules_of(system):
    """Connected components of the bond + constraint graph (openmm.Context.getMolecules()), as a component id per atom."""
    n = system.n_atoms
    parent = np.arange(n)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]; a = parent[a]
        return a
    for a, b in list(np.asarray(system.bond_atoms).reshape(-1, 2)) + list(np.asarray(system.constraint_atoms).reshape(-1, 2)):
        ra, rb = find(int(a)), find(int(b))
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    roots = np.array([find(i) for i in range(n)])
    return np.unique(roots, return_inverse=True)[1]


class MonteCarloBarostat(object):
    def __init__(self, pressure_bar, temperature, frequency=25, seed=None):
        self.pressure = float(pressure_bar) * BAR_NM3          # kJ/mol/nm^3
        self.kT = KB * float(temperature)
        self.frequency = int(frequency)
        self.rng = np.random.RandomState(seed)
        self.volume_scale = None
        self.n_attempted = self.n_accepted = 0
        self.total_attempted = self.total_accepted = 0
        self._mol = None
        self.last = None

    def getFrequency(self):
        return self.frequency

    def scaled_positions(self, x, mol, scale):
        """Every molecule translated so that its centre (plain average of its atoms, as OpenMM takes it) scales with the box."""
        nmol = int(mol.max()) + 1
        cnt = np.bincount(mol, minlength=nmol).astype(np.float64)
        centre = np.stack([np.bincount(mol, weights=x[:, k], minlength=nmol) / cnt for k in range(3)], axis=1)
        return x + (centre * (scale - 1.0))[mol]

    def attempt(self, engine, system):
        """One volume move on `engine` (anything with get_box / set_box / get_positions / set_positions / potential_energy)."""
        if self._mol is None:
            self._mol = molecules_of(system)
        box = np.diag(np.asarray(engine.get_box(), dtype=np.float64).reshape(3, 3)).copy()
        volume = float(np.prod(box))
        if self.volume_scale is None:
            self.volume_scale = 0.01 * volume
        e0 = engine.potential_energy()
        x0 = engine.get_positions()
        dv = self.volume_scale * 2.0 * (self.rng.uniform() - 0.5)
        new_volume = volume + dv
        scale = (new_volume / volume) ** (1.0 / 3.0)
        engine.set_box(box * scale)
        engine.set_positions(self.scaled_positions(x0, self._mol, scale))
        e1 = engine.potential_energy()
        nmol = int(self._mol.max()) + 1
        w = e1 - e0 + self.pressure * dv - nmol * self.kT * np.log(new_volume / volume)
        accept = bool(w <= 0.0 or self.rng.uniform() < np.exp(-w / self.kT))
        if not accept:
            engine.set_box(box)
            engine.set_positions(x0)
        self.n_attempted += 1; self.total_attempted += 1
        self.n_accepted += int(accept); self.total_accepted += int(accept)
        if self.n_attempted >= 10:
            if self.n_accepted < 0.25 * self.n_attempted:
                self.volume_scale /= 1.1
                self.n_attempted = self.n_accepted = 0
            elif self.n_accepted > 0.75 * self.n_attempted:
                self.volume_scale = min(self.volume_scale * 1.1, volume * 0.3)
                self.n_attempted = self.n_accepted = 0
        self.last = {"accepted": accept, "w": float(w), "dV": float(dv), "volume": float(new_volume if accept else volume), "dU": float(e1 - e0)}
        return accept
