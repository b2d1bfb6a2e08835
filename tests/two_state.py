"""A two-state system the HIP engine can run, with populations known WITHOUT the oracle (tests/test_gpu_statistics.py, and its
CPU dress rehearsal tests/test_two_state_cpu.py): the spirit of the reference's only result-pinning test,
/root/reference/blues/tests/test_ethylene.py:107-163 (two states whose Boltzmann populations plain MD cannot equilibrate and NCMC moves
can; the populations come out right only if the whole BLUES iteration -- step program, protocol work, Metropolis test, velocity
re-draw, MD leg -- is right).

One mobile particle L (the "ligand": alchemical in the NCMC System, plain Lennard-Jones in the MD System) between two FROZEN
particles P and Q on the x axis, 2a apart, with different well depths, and a position restraint k |x - c|^2 to their midpoint c
(reference blues/simulation.py:347: the restraint BLUES itself adds).  Periodic box, cutoff 1.0 nm, no charges.  L sits in the part
of P's (or Q's) Lennard-Jones shell that faces c; the barrier between the two basins is ~5 kT, so plain MD almost never crosses, while
the NCMC move -- sterics off, point reflection through c at lambda = 0.5, sterics back on -- does.

U(x) = 4 e_P [(s/r_P)^12 - (s/r_P)^6] + 4 e_Q [(s/r_Q)^12 - (s/r_Q)^6] + k |x - c|^2     (each pair term zero beyond the cutoff)

The populations follow from this formula alone by quadrature (cylindrical symmetry about the axis): p_P = Z(x < c) / Z.
"""
import numpy as np

KB = 0.0083144626
BOX = 3.0
A_HALF = 0.6          # P = c - a, Q = c + a (nm)
SIGMA = 0.3
EPS_L, EPS_P, EPS_Q = 16.0, 16.0, 20.25     # Lorentz-Berthelot: pair well depths sqrt(16 * 16) = 16 and sqrt(16 * 20.25) = 18 kJ/mol
K_RESTR = 30.0        # kJ/mol/nm^2 (energy k d^2, as the reference's CustomExternalForce)
MASS_L = 12.0
CUTOFF = 1.0
N_PAD = 61            # inert frozen atoms (no charge, no epsilon) far from the action: the engine sees an ordinary 64-atom System


def geometry():
    c = np.array([BOX / 2] * 3)
    P, Q = c - np.array([A_HALF, 0, 0]), c + np.array([A_HALF, 0, 0])
    return c, P, Q


def system(alchemical=True):
    """SystemData of the two-state system: atom 0 = L, 1 = P, 2 = Q, then the inert padding."""
    from blues_amd import _abi
    c, P, Q = geometry()
    rmin = SIGMA * 2.0 ** (1.0 / 6.0)
    x = [P + np.array([rmin, 0, 0]), P, Q]
    # padding on a coarse lattice, at least 1.3 nm from c along some axis (it interacts with nothing anyway)
    pts = [np.array([i, j, k]) * (BOX / 5.0) + 0.05 for i in range(5) for j in range(5) for k in range(5)]
    pts = [p for p in pts if np.abs(p - c).max() > 1.1][:N_PAD]
    assert len(pts) == N_PAD
    x = np.array(x + pts)
    n = len(x)
    mass = np.zeros(n); mass[0] = MASS_L
    eps = np.zeros(n); eps[:3] = (EPS_L, EPS_P, EPS_Q)
    sig = np.full(n, SIGMA)
    return _abi.SystemData(box=np.array([BOX] * 3), mass=mass, charge=np.zeros(n), sigma=sig, epsilon=eps,
                           alchemical_atoms=np.array([0] if alchemical else [], np.int32),
                           restraint_atoms=np.array([0], np.int32), restraint_x0=c.reshape(1, 3), restraint_k=K_RESTR,
                           nonbonded_method=_abi.NB_PME_DIRECT, cutoff=CUTOFF, ewald_alpha=2.145966, positions=x,
                           residue_of_atom=np.arange(n, dtype=np.int32))


def potential(x):
    """U at positions x ((..., 3), nm) of L -- the formula above, nothing else."""
    c, P, Q = geometry()
    x = np.asarray(x, dtype=np.float64)
    u = K_RESTR * ((x - c) ** 2).sum(-1)
    for centre, e in ((P, np.sqrt(EPS_L * EPS_P)), (Q, np.sqrt(EPS_L * EPS_Q))):
        r = np.sqrt(((x - centre) ** 2).sum(-1))
        s6 = (SIGMA / np.maximum(r, 1e-9)) ** 6
        u = u + np.where(r < CUTOFF, 4.0 * e * (s6 * s6 - s6), 0.0)
    return u


def exact_populations(temperature, h=0.002, extent=1.3):
    """(p_P, p_Q) by quadrature of exp(-U / kT) over the half spaces x < c and x > c: cylindrical coordinates (x, rho) about the
    P-Q axis, midpoint rule, spacing h."""
    c, _, _ = geometry()
    kT = KB * temperature
    xs = (np.arange(int(2 * extent / h)) + 0.5) * h - extent
    rho = (np.arange(int(extent / h)) + 0.5) * h
    X, R = np.meshgrid(xs, rho, indexing="ij")
    pts = np.stack([c[0] + X, c[1] + R, np.full_like(X, c[2])], axis=-1)
    w = np.exp(-(potential(pts) - potential(pts).min()) / kT) * 2.0 * np.pi * R
    zp, zq = w[X < 0].sum(), w[X > 0].sum()
    return zp / (zp + zq), zq / (zp + zq)


def basin_of(x_L):
    """0: L is on P's side of the midpoint, 1: on Q's."""
    c, _, _ = geometry()
    return (np.asarray(x_L)[..., 0] > c[0]).astype(int)


def reflect(xyz):
    """The move: point reflection of L through the midpoint (an involution: its own reverse, proposal ratio 1)."""
    c, _, _ = geometry()
    return 2.0 * c - np.asarray(xyz, dtype=np.float64)


_MOVE_CLASS = {}


def make_move(moves_module):
    """A PointReflectionMove (one class per moves module: chains of a batch must carry moves of ONE class to share the batched scatter)."""
    if moves_module not in _MOVE_CLASS:
        class PointReflectionMove(moves_module.Move):
            def __init__(self):
                self.atom_indices = [0]
                self.positions = None

            def propose(self, ligand_positions):
                self.positions = ligand_positions
                return reflect(ligand_positions)

            def move(self, context):
                everything = context.getState(getPositions=True).getPositions(asNumpy=True)
                new = self.propose(everything[self.atom_indices]._value)
                for atom, xyz in zip(self.atom_indices, new):
                    everything[atom] = xyz
                context.setPositions(everything)
                return context
        _MOVE_CLASS[moves_module] = PointReflectionMove
    return _MOVE_CLASS[moves_module]()


def build_chain(context_module, r, nstepsNC, nstepsMD, temperature, dt, seed0=5000, precision="mixed", device=0, friction_md=10.0):
    """One BLUES chain on the two-state system: NCMC Simulation on the alchemical System, MD Simulation on the plain one (the
    reference's arrangement, blues/simulation.py:768-809, without the third `alch` context: the correction then comes from the NCMC
    engine at lambda = 1, which IS the MD potential here -- no reciprocal space, no charges)."""
    from blues_amd import integrators, moves, simulation
    ncmc_i = integrators.generateNCMCIntegrator(nstepsNC=nstepsNC, dt=dt, temperature=temperature, seed=seed0 + 2 * r)
    md_i = integrators.LangevinIntegrator(temperature, friction_md, dt, seed=seed0 + 2 * r + 1)
    kw = {} if precision is None else {"precision": precision, "device": device}
    ncmc = context_module.Simulation(None, system(True), ncmc_i, replica=r, **kw)
    md = context_module.Simulation(None, system(False), md_i, replica=r, **kw)
    md.context.setVelocitiesToTemperature(temperature, seed0 + 7 * r + 3)
    cfg = {"nIter": 1, "nstepsNC": nstepsNC, "nstepsMD": nstepsMD, "moveStep": nstepsNC // 2}
    chain = simulation.BLUESSimulation(simulation.SimulationSet(ncmc, md=md), cfg, moves.MoveEngine(make_move(moves)),
                                       rng=np.random.RandomState(seed0 + 11 * r + 5))
    return chain
