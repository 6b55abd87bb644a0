"""Mirror of src/d2d/multiopty_utils.py: the aircraft set and the multi-aircraft cost
plug-ins (same protocol as d2d.opty_utils; slices are lists, one per aircraft)."""
import numpy as np

import d2d.opty_utils as d2ou


class AircraftSet:
    def __init__(self, n=2):
        self.st = d2ou._Sym('t')
        self.nb_aicraft = n                    # (sic) the reference's attribute name
        self.aircraft = [d2ou.Aircraft(self.st, i) for i in range(n)]
        self._state_symbols = tuple(s for ac in self.aircraft for s in ac._state_symbols)
        self._input_symbols = tuple(s for ac in self.aircraft for s in ac._input_symbols)

    def get_eom(self, wind, g=9.81):
        """The stacked model of all aircraft (src/d2d/multiopty_utils.py:25-26)."""
        return d2ou.Eom(wind.sample_sym(self.st, None, None), g, tuple(ac._id for ac in self.aircraft))


def _scale(_p):
    return _p.obj_scale / _p.num_nodes / _p.acs.nb_aicraft


class CostNull:
    def cost(self, free, _p):
        return 0.

    def cost_grad(self, free, _p):
        return np.zeros_like(free)


class CostInput:
    """scale/N/n_ac * (kv sum (v-vsp)^2 + kphi sum phi^2) over all aircraft (src/d2d/multiopty_utils.py:56-71)."""

    def __init__(self, vsp=10., kv=1., kphi=1.):
        self.vsp, self.kv, self.kphi = vsp, kv, kphi

    def cost(self, free, _p):
        sp = sum(np.sum(free[s] ** 2) for s in _p._slice_phi)
        sv = sum(np.sum((free[s] - self.vsp) ** 2) for s in _p._slice_v)
        return _scale(_p) * (self.kv * sv + self.kphi * sp)

    def cost_grad(self, free, _p):
        g = np.zeros_like(free)
        for s in _p._slice_phi:
            g[s] = self.kphi * 2 * free[s]
        for s in _p._slice_v:
            g[s] = self.kv * 2 * (free[s] - self.vsp)
        return g * _scale(_p)


class CostAirvel(CostInput):
    """(src/d2d/multiopty_utils.py:33-43)."""

    def __init__(self, vsp=10.):
        CostInput.__init__(self, vsp, 1., 0.)


class CostBank(CostInput):
    """(src/d2d/multiopty_utils.py:45-54)."""

    def __init__(self):
        CostInput.__init__(self, 0., 0., 1.)


class CostObstacle:
    """Obstacle penalty on aircraft 0 only, scale without 1/n_ac (src/d2d/multiopty_utils.py:74-106)."""

    def __init__(self, c=(0, 0), r=10., kind=0):
        self.c, self.r, self.kind, self.k = c, r, kind, 2

    def _d(self, free, _p):
        return free[_p._slice_x[0]] - self.c[0], free[_p._slice_y[0]] - self.c[1]

    def cost(self, free, _p):
        return _p.obj_scale / _p.num_nodes * np.sum(d2ou._obstacle_field(*self._d(free, _p), self.r, self.kind, self.k))

    def cost_grad(self, free, _p):
        dx, dy = self._d(free, _p)
        e = d2ou._obstacle_field(dx, dy, self.r, self.kind, self.k)
        g = np.zeros_like(free)
        g[_p._slice_x[0]] = _p.obj_scale / _p.num_nodes * -2. * dx * e
        g[_p._slice_y[0]] = _p.obj_scale / _p.num_nodes * -2. * dy * e
        return g


class CostObstacles:
    def __init__(self, obss, kind=0):
        self.obss = [CostObstacle(c=(o[0], o[1]), r=o[2], kind=kind) for o in obss]

    def cost(self, free, _p):
        return np.sum([c.cost(free, _p) for c in self.obss])

    def cost_grad(self, free, _p):
        return np.sum([c.cost_grad(free, _p) for c in self.obss], axis=0)


class CostCollision:
    """Gaussian proximity penalty between aircraft 0 and 1 (src/d2d/multiopty_utils.py:120-153)."""

    def __init__(self, r=3., k=2.):
        self.r, self.k = r, k

    def _d(self, free, _p):
        return free[_p._slice_x[0]] - free[_p._slice_x[1]], free[_p._slice_y[0]] - free[_p._slice_y[1]]

    def cost(self, free, _p):
        dx, dy = self._d(free, _p)
        return _p.obj_scale / _p.num_nodes * np.sum(d2ou._obstacle_field(dx, dy, self.r, 1, self.k))

    def cost_grad(self, free, _p):
        dx, dy = self._d(free, _p)
        e = d2ou._obstacle_field(dx, dy, self.r, 1, self.k)
        f = _p.obj_scale / _p.num_nodes
        g = np.zeros_like(free)
        g[_p._slice_x[0]], g[_p._slice_y[0]] = f * -2. * dx * e, f * -2. * dy * e
        g[_p._slice_x[1]], g[_p._slice_y[1]] = f * 2. * dx * e, f * 2. * dy * e
        return g


class CostComposit:
    """CostInput [+ kobs * obstacles] [+ kcol * collision]; a NaN weight switches a term off
    (src/d2d/multiopty_utils.py:156-174)."""

    def __init__(self, kvel=1., kbank=1., kobs=float('Nan'), kcol=float('NaN'), vsp=10., obss=[], obs_kind=0, rcol=3.):
        self.kvel, self.kbank, self.kobs, self.kcol = kvel, kbank, kobs, kcol
        self.vsp, self.obss, self.obs_kind, self.rcol = vsp, obss, obs_kind, rcol
        self.ci = CostInput(vsp, kvel, kbank)
        self.cobs = CostObstacles(obss, obs_kind)      # always constructed, as in the reference (:160-161)
        self.ccol = CostCollision(r=rcol)

    def _extra(self):
        return [(k, c) for k, c in ((self.kobs, self.cobs), (self.kcol, self.ccol)) if not np.isnan(k)]

    def cost(self, free, _p):
        return self.ci.cost(free, _p) + sum(k * c.cost(free, _p) for k, c in self._extra())

    def cost_grad(self, free, _p):
        return self.ci.cost_grad(free, _p) + sum(k * c.cost_grad(free, _p) for k, c in self._extra())
