"""The Move plugin API (reference blues/moves.py:39-145, 313-410) and the flagship move,
RandomLigandRotationMove (reference blues/moves.py:148-310).

Moves are host-side geometry on a handful of atoms, executed once or thrice per switch;
they talk to the engine only through Context.getState / setPositions / setVelocities,
exactly as in the reference, so third-party Move subclasses work unchanged.
"""
import logging
import sys
import traceback

import numpy

from . import unit

logger = logging.getLogger(__name__)


class Move(object):
    """Base class: same five hooks as reference blues/moves.py:49-145."""

    def __init__(self):
        pass

    def initializeSystem(self, system, integrator):
        return system, integrator

    def beforeMove(self, context):
        return context

    def afterMove(self, context):
        return context

    def _error(self, context):
        return context

    def move(self, context):
        return context


def uniform_quaternion(random_state=None):
    """Uniformly distributed unit quaternion (Shoemake's method, as mdtraj.utils.uniform_quaternion)."""
    rs = random_state if isinstance(random_state, numpy.random.RandomState) else numpy.random.RandomState(random_state) if random_state is not None else numpy.random
    u = rs.uniform(0.0, 1.0, size=3)
    return numpy.array([numpy.sqrt(1 - u[0]) * numpy.sin(2 * numpy.pi * u[1]), numpy.sqrt(1 - u[0]) * numpy.cos(2 * numpy.pi * u[1]),
                        numpy.sqrt(u[0]) * numpy.sin(2 * numpy.pi * u[2]), numpy.sqrt(u[0]) * numpy.cos(2 * numpy.pi * u[2])])


def rotation_matrix_from_quaternion(q):
    a, b, c, d = q
    return numpy.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                        [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                        [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


class RandomLigandRotationMove(Move):
    """Random rotation of the ligand about its centre of mass (reference blues/moves.py:278-310).
    The reference derives atom_indices and masses from a parmed.Structure and a residue name;
    here they are given directly (the ParmEd selection machinery is out of scope)."""

    def __init__(self, atom_indices, masses, random_state=None, resname='LIG'):
        self.resname = resname
        self.random_state = random_state if not isinstance(random_state, (int, numpy.integer)) else numpy.random.RandomState(random_state)
        self.atom_indices = [int(i) for i in atom_indices]
        self.masses = numpy.asarray(masses, dtype=numpy.float64).reshape(-1, 1)
        self.totalmass = float(self.masses.sum())
        self.center_of_mass = None
        self.positions = None

    def getCenterOfMass(self, positions, masses):
        """reference blues/moves.py:253-270 (float32 coordinates, as parmed.geometry.center_of_mass is fed)"""
        coordinates = numpy.asarray(positions, numpy.float32)
        return (coordinates * masses).sum(0) / masses.sum()

    def propose(self, ligand_positions):
        """The geometry of the move on the ligand's own coordinates ((n_ligand, 3), nm) -> the rotated coordinates: random rotation
        about the centre of mass (reference blues/moves.py:293-302).  `move` applies it through the Context; a replica batch calls
        it for every chain between ONE gather and ONE scatter of all chains' ligand atoms (BatchedBLUESSimulation)."""
        self.positions = ligand_positions
        pivot = self.center_of_mass = self.getCenterOfMass(ligand_positions, self.masses)
        rotation = rotation_matrix_from_quaternion(uniform_quaternion(self.random_state))
        # row vectors times the matrix (x' = x R, not R x): the convention the reference applies its matrix in
        return pivot + (ligand_positions - pivot) @ rotation

    def move(self, context):
        """reference blues/moves.py:278-310: read the ligand out of the State's positions, rotate, assign, hand the positions back
        (on this engine the Quantity moves only the ligand's coordinates between device and host)."""
        everything = context.getState(getPositions=True).getPositions(asNumpy=True)
        rotated = self.propose(everything[self.atom_indices]._value)
        for atom, xyz in zip(self.atom_indices, rotated):
            everything[atom] = xyz
        context.setPositions(everything)
        self.positions = everything[self.atom_indices]._value
        return context


def _min_image(d, box):
    return d - box * numpy.round(d / box)


class WaterTranslationMove(Move):
    """Translates one (alchemical) water to a uniformly random point of a sphere around a reference group's centre of
    mass -- the hook behaviour of reference blues/moves.py:846-1083:
      beforeMove  picks a random water whose oxygen lies within `radius` of the centre and swaps positions AND velocities
                  with the alchemical water (the first water), or disables the move if there is none;
      move        re-places the alchemical water at a uniform random point of the sphere (if it is still inside);
      afterMove   forces a rejection through protocol_work = 999999 when the water ended up outside the sphere.
    The reference finds waters / protein atoms through ParmEd and MDTraj selections; here they are index lists."""

    def __init__(self, water_residues, protein_atoms, protein_masses, radius=2.3):
        self.radius = unit.value_in(radius, "nanometer")
        self.water_residues = [list(map(int, w)) for w in water_residues]
        self.atom_indices = self.water_residues[0]  # the alchemical water
        self.protein_atoms = [int(i) for i in protein_atoms]
        self.protein_masses = numpy.asarray(protein_masses, dtype=numpy.float64).reshape(-1, 1)
        self.go = True

    def _random_sphere_point(self, radius, origin):
        r = radius * (numpy.random.random() ** (1. / 3.))
        phi = numpy.random.uniform(0, 2 * numpy.pi)
        costheta = numpy.random.uniform(-1, 1)
        theta = numpy.arccos(costheta)
        return numpy.array([numpy.sin(theta) * numpy.cos(phi), numpy.sin(theta) * numpy.sin(phi), numpy.cos(theta)]) * r + origin

    def _centre(self, x):
        return (x[self.protein_atoms] * self.protein_masses).sum(0) / self.protein_masses.sum()

    def _distance(self, x, atom, centre, box):
        return float(numpy.linalg.norm(_min_image(x[atom] - centre, box)))

    def beforeMove(self, context):
        st = context.getState(getPositions=True, getVelocities=True)
        x = st.getPositions(asNumpy=True)._value.copy()
        v = st.getVelocities(asNumpy=True)._value.copy()
        box = numpy.diag(numpy.asarray(st.getPeriodicBoxVectors(asNumpy=True)._value))
        centre = self._centre(x)
        shuffled = [list(w) for w in self.water_residues]
        numpy.random.shuffle(shuffled)
        choice = None
        for w in shuffled:
            if self._distance(x, w[0], centre, box) <= self.radius:
                choice = w
                break
        if choice is None:
            self.go = False
            return context
        a = self.atom_indices
        xa, va = x[a].copy(), v[a].copy()
        x[a], v[a] = x[choice], v[choice]
        x[choice], v[choice] = xa, va
        context.setPositions(unit.Quantity(x, "nanometer"))
        context.setVelocities(unit.Quantity(v, "nanometer/picosecond"))
        self.go = True
        return context

    def move(self, context):
        if not self.go:
            return context
        st = context.getState(getPositions=True)
        x = st.getPositions(asNumpy=True)._value.copy()
        box = numpy.diag(numpy.asarray(st.getPeriodicBoxVectors(asNumpy=True)._value))
        centre = self._centre(x)
        if self._distance(x, self.atom_indices[0], centre, box) >= self.radius:
            return context  # the water drifted out of the region: no move
        target = self._random_sphere_point(self.radius, centre)
        x[self.atom_indices] = x[self.atom_indices] - (x[self.atom_indices[0]] - target)
        context.setPositions(unit.Quantity(x, "nanometer"))
        return context

    def afterMove(self, context):
        st = context.getState(getPositions=True)
        x = st.getPositions(asNumpy=True)._value
        box = numpy.diag(numpy.asarray(st.getPeriodicBoxVectors(asNumpy=True)._value))
        if self._distance(x, self.atom_indices[0], self._centre(x), box) > self.radius and self.go:
            context._integrator.setGlobalVariableByName("protocol_work", 999999)
        return context


class SideChainMove(Move):
    """Side-chain torsion move chosen the way the reference chooses it (blues/moves.py:418-844), with the bookkeeping the reference
    delegates to OpenEye re-derived from the bonded graph alone:

      * backbone atoms (getBackboneAtoms, :483-509): the atoms named N, CA, C, O (OEIsBackboneAtom's heavy set), when names are given;
      * query atoms (getTargetAtoms, :511-558): the non-backbone atoms of the residues in `residue_list`, water left out;
      * rotatable heavy bonds (findHeavyRotBonds, :560-601): bonds of a query atom whose two ends are heavy and that OEBond::IsRotor()
        would accept -- here: not in a ring (removing the bond disconnects its ends) and not terminal (each end has another heavy
        neighbour).  Bond orders are not in an Amber topology, so multiple bonds are recognised only by being terminal or in rings;
      * rotating atoms (getRotAtoms, :603-669): [axis1, axis2] followed by what the reference's bounded breadth-first walk from the
        non-backbone axis atoms collects (neighbours, and neighbours of heavy neighbours, never crossing axis2 or the backbone) --
        restated as written, quirks included (the walk starts from BOTH non-backbone axis atoms);
      * chooseBondandTheta (:703-729): uniform residue, uniform bond of it, theta uniform in [0, 2 pi);
      * move (:752-844): Rodrigues rotation (rotation_matrix, :731-750) of the target atoms about axis1 - axis2 through axis2.

    `atom_indices` is the reference's nested dict {residue: {bond: [atoms]}} (its quirk: SimulationFactory receives dict keys,
    SURVEY.md Appendix E); `alchemical_atoms` gives the flat list a System needs."""

    BACKBONE_NAMES = ("N", "CA", "C", "O")

    def __init__(self, system, residue_list, names=None, verbose=False, write_move=False, random_state=None):
        self.residue_list = list(residue_list)
        self.verbose, self.write_move = verbose, write_move
        self.random_state = random_state if isinstance(random_state, numpy.random.RandomState) else numpy.random.RandomState(random_state)
        n = system.n_atoms
        names = names if names is not None else system.names
        res = numpy.asarray(system.residue_of_atom if system.residue_of_atom is not None else numpy.zeros(n, int))
        heavy = numpy.asarray(system.mass) > 3.5   # (hydrogen mass repartitioning puts hydrogens at ~3 Da; frozen atoms have mass 0: names, when given, decide)
        if names is not None:
            heavy = numpy.array([not str(nm).strip().upper().startswith("H") for nm in names])
        nbr = [[] for _ in range(n)]
        pairs = [tuple(b) for b in numpy.asarray(system.bond_atoms).reshape(-1, 2)] + [tuple(c) for c in numpy.asarray(system.constraint_atoms).reshape(-1, 2)]
        for a, b in pairs:   # (HBonds constraints removed those bonds from the harmonic terms: both lists make up the graph)
            a, b = int(a), int(b)
            if b not in nbr[a]:
                nbr[a].append(b); nbr[b].append(a)
        self._nbr, self._heavy = nbr, heavy
        backbone = set(i for i in range(n) if names is not None and str(names[i]).strip() in self.BACKBONE_NAMES)
        self.backbone_atoms = sorted(backbone)
        # (a 3-atom residue with one heavy atom is water: the reference skips "HOH" by name)
        res_sizes = numpy.bincount(res)
        qry = [i for i in range(n) if i not in backbone and int(res[i]) in self.residue_list and not (res_sizes[res[i]] == 3 and int(heavy[res == res[i]].sum()) == 1)]
        self.qry_atoms = qry
        self.rot_bonds = {}
        for a in qry:
            for b in nbr[a]:
                key = (min(a, b), max(a, b))
                if key in self.rot_bonds or not (heavy[a] and heavy[b]):
                    continue
                if self._is_rotor(key[0], key[1]):
                    self.rot_bonds[key] = int(res[a])
        self.rot_atoms = self._rot_atoms()
        self.atom_indices = self.rot_atoms
        self.last_choice = None

    # ---- graph predicates standing in for OEBond::IsRotor()
    def _is_rotor(self, a, b):
        nbr, heavy = self._nbr, self._heavy
        if not any(heavy[c] for c in nbr[a] if c != b) or not any(heavy[c] for c in nbr[b] if c != a):
            return False   # terminal
        seen, stack = {a}, [a]   # in a ring iff b is reachable from a without the bond itself
        while stack:
            u = stack.pop()
            for w in nbr[u]:
                if (u == a and w == b) or w in seen:
                    continue
                if w == b:
                    return False
                seen.add(w); stack.append(w)
        return True

    def _rot_atoms(self):
        """reference blues/moves.py:603-669, statement for statement on the graph"""
        out = {}
        backbone = set(self.backbone_atoms)
        for (ax1, ax2), resnum in self.rot_bonds.items():
            idx_list = [ax1, ax2]
            query = []
            if ax1 not in backbone:
                query.append(ax1)
            if ax2 not in query and ax2 not in backbone:
                query.append(ax2)
            k = 0
            while k < len(query):   # the reference appends to the list it iterates over
                atom = query[k]; k += 1
                for cand in self._nbr[atom]:
                    if cand not in query and cand not in backbone and cand != ax2:
                        query.append(cand)
                        if self._heavy[cand]:
                            for nb in self._nbr[cand]:
                                if nb not in query and cand not in backbone and cand != ax2:
                                    query.append(nb)
            for y in query:
                if y not in idx_list:
                    idx_list.append(y)
            out.setdefault(resnum, {})[(ax1, ax2)] = idx_list
        return out

    @property
    def alchemical_atoms(self):
        return sorted({a for bonds in self.rot_atoms.values() for atoms in bonds.values() for a in atoms})

    def chooseBondandTheta(self):
        rs = self.random_state
        residues = list(self.rot_atoms.keys())
        res_choice = residues[rs.randint(len(residues))]
        bonds = list(self.rot_atoms[res_choice].keys())
        bond_choice = bonds[rs.randint(len(bonds))]
        return rs.uniform(0.0, 2 * numpy.pi), self.rot_atoms[res_choice][bond_choice], res_choice, bond_choice

    @staticmethod
    def rotation_matrix(axis, theta):
        """reference blues/moves.py:731-750 (Euler-Rodrigues parameters)"""
        axis = numpy.asarray(axis, dtype=float)
        axis = axis / numpy.sqrt(numpy.dot(axis, axis))
        a = numpy.cos(theta / 2.0)
        b, c, d = -axis * numpy.sin(theta / 2.0)
        aa, bb, cc, dd = a * a, b * b, c * c, d * d
        bc, ad, ac, ab, bd, cd = b * c, a * d, a * c, a * b, b * d, c * d
        return numpy.array([[aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac)], [2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab)],
                            [2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc]])

    def move(self, context, verbose=False):
        theta, target_atoms, res, bond = self.chooseBondandTheta()
        self.last_choice = (theta, res, bond)
        logger.info('Rotating bond: %s in resnum: %s by %.2f radians' % (bond, res, theta))
        q = context.getState(getPositions=True).getPositions(asNumpy=True)
        idx = [int(i) for i in target_atoms]
        x = numpy.array(q[idx]._value, dtype=float)          # only the touched atoms travel (DeviceQuantity)
        axis1, axis2 = x[0], x[1]
        rot = self.rotation_matrix(axis1 - axis2, theta)
        new = (x - axis2) @ rot.T + axis2
        for k, i in enumerate(idx):
            q[i] = new[k]
        context.setPositions(q)
        return context


class TorsionRotationMove(Move):
    """Rotates a set of atoms about a bond axis by a uniformly random angle -- the geometric core of the reference's
    SideChainMove.move (blues/moves.py:703-750, 752-844), without its OpenEye-based rotamer bookkeeping."""

    def __init__(self, axis_atoms, rotating_atoms, random_state=None):
        self.axis_atoms = (int(axis_atoms[0]), int(axis_atoms[1]))
        self.atom_indices = [int(i) for i in rotating_atoms]
        self.random_state = random_state if isinstance(random_state, numpy.random.RandomState) else numpy.random.RandomState(random_state)
        self.last_angle = None

    def move(self, context):
        x = context.getState(getPositions=True).getPositions(asNumpy=True)._value.copy()
        a, b = x[self.axis_atoms[0]], x[self.axis_atoms[1]]
        u = (b - a) / numpy.linalg.norm(b - a)
        th = self.random_state.uniform(0.0, 2 * numpy.pi)
        self.last_angle = th
        c, s_ = numpy.cos(th), numpy.sin(th)
        for i in self.atom_indices:  # Rodrigues rotation about the axis through b
            r = x[i] - b
            x[i] = b + r * c + numpy.cross(u, r) * s_ + u * numpy.dot(u, r) * (1 - c)
        context.setPositions(unit.Quantity(x, "nanometer"))
        return context


class MoveEngine(object):
    """reference blues/moves.py:313-410"""

    def __init__(self, moves, probabilities=None):
        self.moves = moves if isinstance(moves, list) else [moves]
        if probabilities is None:
            self.probabilities = [1. / len(self.moves) for _ in self.moves]
        else:
            prob_sum = float(sum(probabilities))
            self.probabilities = [x / prob_sum for x in probabilities]
        if len(self.moves) != len(self.probabilities):
            print('moves and probability list lengths need to match')
            raise IndexError
        self.selected_move = None
        self.move_name = None

    def selectMove(self):
        rand_num = numpy.random.choice(len(self.probabilities), p=self.probabilities)
        self.selected_move = self.moves[rand_num]
        self.move_name = self.selected_move.__class__.__name__

    def runEngine(self, context):
        try:
            new_context = self.selected_move.move(context)
        except Exception as e:
            print('Error: move not implemented correctly, printing traceback:')
            ex_type, ex, tb = sys.exc_info()
            traceback.print_tb(tb)
            print(e)
            raise SystemExit
        return new_context
