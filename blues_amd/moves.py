"""The Move plugin API (reference blues/moves.py:39-145, 313-410) and the flagship move,
RandomLigandRotationMove (reference blues/moves.py:148-310).

Moves are host-side geometry on a handful of atoms, executed once or thrice per switch;
they talk to the engine only through Context.getState / setPositions / setVelocities,
exactly as in the reference, so third-party Move subclasses work unchanged.
"""
import sys
import traceback

import numpy

from . import unit


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

    def move(self, context):
        """reference blues/moves.py:278-310, statement for statement: the State's positions are indexed and assigned through
        the Quantity (which, on this engine, moves only the ligand's coordinates between device and host)."""
        positions = context.getState(getPositions=True).getPositions(asNumpy=True)
        self.positions = positions[self.atom_indices]._value
        self.center_of_mass = self.getCenterOfMass(self.positions, self.masses)
        reduced_pos = self.positions - self.center_of_mass
        rand_quat = uniform_quaternion(self.random_state)
        rand_rotation_matrix = rotation_matrix_from_quaternion(rand_quat)
        rot_move = numpy.dot(reduced_pos, rand_rotation_matrix) + self.center_of_mass
        for index, atomidx in enumerate(self.atom_indices):
            positions[atomidx] = rot_move[index]
        context.setPositions(positions)
        self.positions = positions[self.atom_indices]._value
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
