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
        positions = context.getState(getPositions=True).getPositions(asNumpy=True)
        x = positions._value
        self.positions = x[self.atom_indices]
        self.center_of_mass = self.getCenterOfMass(self.positions, self.masses)
        reduced_pos = self.positions - self.center_of_mass
        rand_quat = uniform_quaternion(self.random_state)
        rand_rotation_matrix = rotation_matrix_from_quaternion(rand_quat)
        rot_move = numpy.dot(reduced_pos, rand_rotation_matrix) + self.center_of_mass
        for index, atomidx in enumerate(self.atom_indices):
            x[atomidx] = rot_move[index]
        context.setPositions(unit.Quantity(x, "nanometer"))
        self.positions = x[self.atom_indices]
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
