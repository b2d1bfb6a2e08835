"""A deliberately small stand-in for simtk.unit, which BLUES uses for every quantity that
crosses its API (reference blues/simulation.py:20-21) and which does not exist on the MI355X
box.  Values are stored in OpenMM's unit system (nm, ps, K, kJ/mol, amu, e); a Quantity
remembers its unit name, exposes `._value`, `.value_in_unit()` and enough arithmetic for
BLUES-style driver code (`work * (-1.0 / kT)`, `state['potential_energy']._value`).
Objects from a real simtk.unit are accepted wherever a number is expected (duck-typed).
"""
import numpy as np

# factor to OpenMM base units, dimension tag
_UNITS = {
    "nanometer": (1.0, "L"), "nanometers": (1.0, "L"), "angstrom": (0.1, "L"), "angstroms": (0.1, "L"),
    "picosecond": (1.0, "T"), "picoseconds": (1.0, "T"), "femtosecond": (1e-3, "T"), "femtoseconds": (1e-3, "T"),
    "nanosecond": (1e3, "T"), "nanoseconds": (1e3, "T"),
    "kelvin": (1.0, "K"),
    "kilojoule/mole": (1.0, "E"), "kilojoules_per_mole": (1.0, "E"), "kilocalorie/mole": (4.184, "E"), "kilocalories_per_mole": (4.184, "E"),
    "dalton": (1.0, "M"), "daltons": (1.0, "M"), "amu": (1.0, "M"),
    "1/picosecond": (1.0, "1/T"), "nanometer/picosecond": (1.0, "L/T"),
    "kilojoule/(nanometer*mole)": (1.0, "E/L"), "kilojoule/(nanometer**2*mole)": (1.0, "E/L2"),
    "kilocalorie/(angstrom**2*mole)": (418.4, "E/L2"),
    "dimensionless": (1.0, "1"),
}


class Quantity(object):
    __array_priority__ = 100

    def __init__(self, value, unit="dimensionless"):
        if unit not in _UNITS:
            raise ValueError("unknown unit %r" % unit)
        self._value = value
        self.unit = unit

    def value_in_unit(self, unit):
        f0, d0 = _UNITS[self.unit]
        f1, d1 = _UNITS[str(unit)]
        if d0 != d1:
            raise TypeError("cannot convert %s to %s" % (self.unit, unit))
        return self._value * (f0 / f1)

    def in_units_of(self, unit):
        return Quantity(self.value_in_unit(unit), str(unit))

    def _same(self, other):
        if isinstance(other, Quantity):
            return other.value_in_unit(self.unit)
        return other

    def __add__(self, o): return Quantity(self._value + self._same(o), self.unit)
    __radd__ = __add__
    def __sub__(self, o): return Quantity(self._value - self._same(o), self.unit)
    def __rsub__(self, o): return Quantity(self._same(o) - self._value, self.unit)
    def __neg__(self): return Quantity(-self._value, self.unit)

    def __mul__(self, o):
        if isinstance(o, _Reciprocal):
            return o.num * (self / o.q)
        if isinstance(o, Quantity):
            if _UNITS[o.unit][1] == "1":
                return Quantity(self._value * o._value, self.unit)
            raise TypeError("Quantity*Quantity is only supported with a dimensionless factor")
        return Quantity(self._value * o, self.unit)
    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, Quantity):
            if _UNITS[o.unit][1] == _UNITS[self.unit][1]:
                return self._value / o.value_in_unit(self.unit)  # same dimension -> plain number
            raise TypeError("unsupported Quantity division")
        return Quantity(self._value / o, self.unit)

    def __rtruediv__(self, o):  # number / Quantity, e.g. -1.0 / kT  ->  "per-energy" factor
        return _Reciprocal(o, self)

    def __getitem__(self, k): return Quantity(self._value[k], self.unit)
    def __setitem__(self, k, v): self._value[k] = self._same(v)
    def __len__(self): return len(self._value)
    def __float__(self): return float(self._value)
    def __eq__(self, o): return np.all(self._value == self._same(o))
    def __lt__(self, o): return self._value < self._same(o)
    def __gt__(self, o): return self._value > self._same(o)
    def __repr__(self): return "Quantity(value=%r, unit=%s)" % (self._value, self.unit)
    def __format__(self, spec): return format(self._value, spec) + " " + self.unit


class DeviceQuantity(Quantity):
    """A Quantity whose array still lives on the GPU (a State's positions or velocities).  Reading `_value` -- or anything
    that goes through it -- downloads it once; handed back untouched to Context.setPositions / setVelocities it never
    leaves the device.  What a Move typically does to positions -- read `q[indices]`, assign `q[i] = xyz` -- is served
    without the full array: the few atoms are fetched, the assignments are remembered and applied on the device when the
    Quantity is handed to setPositions.  Once `_value` has been read the host copy is authoritative."""

    def __init__(self, snapshot, kind, unit, wrapper=None):
        if unit not in _UNITS:
            raise ValueError("unknown unit %r" % unit)
        self.unit = unit
        self._snapshot, self._kind, self._host = snapshot, kind, None
        self._wrapper = wrapper if kind == 1 else None   # enforcePeriodicBox (context.PeriodicWrapper): applied to what reaches the host
        self._edits = {}   # atom index -> (3,) values in this Quantity's unit, while the array is still on the device

    @property
    def _value(self):
        if self._host is None:
            self._host = self._snapshot.read(self._kind).copy()
            if self._wrapper is not None:
                self._host = self._wrapper.wrap_all(self._host)
            for i, row in self._edits.items():
                self._host[i] = row
            self._edits = {}
        return self._host

    @_value.setter
    def _value(self, v):
        self._host = v
        self._edits = {}

    def on_device(self):
        """The untouched device copy, or None once the host has seen (and may have changed) the numbers."""
        return self._snapshot if self._host is None else None

    def pending_edits(self):
        """(indices, values) assigned while the array was still on the device."""
        idx = sorted(self._edits)
        return np.asarray(idx, dtype=np.int64), (np.array([self._edits[i] for i in idx], dtype=np.float64).reshape(-1, 3))

    @staticmethod
    def _atoms(k):
        """int / 1-D integer sequence -> (index array, is_scalar); None for anything else (slices, masks, tuples)."""
        if isinstance(k, (int, np.integer)):
            return np.array([int(k)], dtype=np.int64), True
        if isinstance(k, (list, np.ndarray)):
            a = np.asarray(k)
            if a.ndim == 1 and a.dtype.kind in "iu":
                return a.astype(np.int64), False
        return None

    def __getitem__(self, k):
        sel = self._atoms(k) if self._host is None else None
        if sel is None:
            return Quantity(self._value[k], self.unit)
        idx, scalar = sel
        n = self._snapshot.engine.n
        idx = [int(i) + (n if i < 0 else 0) for i in idx]
        if all(i in self._edits for i in idx):      # everything asked for was assigned here: no device round trip
            rows = np.array([self._edits[i] for i in idx], dtype=np.float64).reshape(-1, 3)
        else:
            rows = self._snapshot.read_atoms(self._kind, idx)
            if self._wrapper is not None:
                rows = rows + self._wrapper.shifts_for(idx, lambda need: self._snapshot.read_atoms(self._kind, [int(q) for q in need]))
            for j, i in enumerate(idx):
                if i in self._edits:
                    rows[j] = self._edits[i]
        return Quantity(rows[0] if scalar else rows, self.unit)

    def __setitem__(self, k, v):
        sel = self._atoms(k) if self._host is None else None
        if sel is None:
            self._value[k] = self._same(v)
            return
        idx, scalar = sel
        n = self._snapshot.engine.n
        vals = np.array(self._same(v), dtype=np.float64)
        if scalar:
            i = int(idx[0])
            self._edits[i + (n if i < 0 else 0)] = vals.reshape(3)
            return
        vals = np.broadcast_to(vals.reshape(-1, 3), (len(idx), 3))
        for j, i in enumerate(idx):
            self._edits[int(i) + (n if i < 0 else 0)] = vals[j].copy()

    def __len__(self):
        return self._snapshot.engine.n if self._host is None else len(self._host)


class _Reciprocal(object):
    """number / Quantity, so that `energy * (-1.0 / kT)` (reference blues/simulation.py:1116-1117) is a float."""

    def __init__(self, num, q):
        self.num, self.q = num, q

    def __rmul__(self, o):
        if isinstance(o, Quantity):
            return self.num * (o / self.q)
        raise TypeError("unsupported")
    __mul__ = __rmul__


def value_in(x, unit):
    """float | our Quantity | a simtk Quantity  ->  float in `unit`."""
    if isinstance(x, Quantity):
        return x.value_in_unit(unit)
    if hasattr(x, "value_in_unit"):  # simtk.unit.Quantity
        try:
            import simtk.unit as su  # pragma: no cover
            table = {"kelvin": su.kelvin, "picosecond": su.picosecond, "1/picosecond": su.picosecond ** -1,
                     "nanometer": su.nanometer, "kilojoule/mole": su.kilojoule_per_mole, "dalton": su.dalton}
            return x.value_in_unit(table[unit])
        except Exception:
            return x._value
    return float(x) if np.isscalar(x) else np.asarray(x, dtype=np.float64)


kelvin, picosecond, picoseconds, femtoseconds, nanometer, nanometers, angstroms = (
    "kelvin", "picosecond", "picoseconds", "femtoseconds", "nanometer", "nanometers", "angstroms")
kilojoules_per_mole, kilojoule_per_mole, daltons = "kilojoules_per_mole", "kilojoule/mole", "daltons"
