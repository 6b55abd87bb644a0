"""Mirror of src/d2d/opty_utils.py (reference): planner timing, wind, the aircraft symbol
holder, the single-aircraft cost plug-ins and the 'triangle' initial guess.

The cost classes keep the reference's plug-in protocol -- cost(free, planner) and
cost_grad(free, planner) on the node vector [x, y, psi, phi, v] -- so that user code and the
reference's own fake-planner checks (src/test/test_objective.py) keep working.  The planners
recognise these classes structurally and lower them to kernel parameters
(single_opt_planner.lower_cost); their gradients reproduce the reference's expressions,
including where those differ from the true derivative."""
import numpy as np


def planner_timing(t0, t1, hz):
    """Node count, step and rounded duration (src/d2d/opty_utils.py:8-14)."""
    num_nodes = int((t1 - t0) * hz) + 1
    time_step = 1. / hz
    duration = (num_nodes - 1) * time_step
    print(f'time_step: {time_step:.3f}s ({hz:.1f}hz), duration {duration:.1f}s -> {num_nodes} nodes')
    return num_nodes, time_step, duration


class WindField:
    def __init__(self, w=[0., 0.]):
        self.w = w

    def sample_sym(self, _t, _x, _y):
        return self.w

    def sample_num(self, _t, _x, _y):
        return self.w

    def __str__(self):
        return f'{self.w} m/s'


class _Sym:
    """Named placeholder standing where the reference holds a sympy Function / Symbol.  No computer algebra is needed on this
    side -- the model is fixed (the three kinematic equations below) and lives in the kernels -- but the reference's planner
    code builds its instance constraints and bounds out of these objects (`_g._sx(t0) - x0`, `bounds[_g._sphi(_g._st)] = ...`,
    src/single_opt_planner.py:46-57), so they support exactly that: calling (-> the function at a time), subtraction of a
    number (-> an instance constraint), hashing (-> keys of the bounds dictionary)."""

    def __init__(self, name):
        self.name = name

    def __call__(self, t=None):
        return _SymAt(self, t)

    def __repr__(self):
        return self.name


class _SymAt:
    """`x(t)`: a state / input function at a time (a number for instance constraints, the time symbol for bounds)."""

    def __init__(self, sym, t):
        self.sym, self.t = sym, t

    def __sub__(self, value):
        return InstanceConstraint(self.sym.name, self.t, float(value))

    def __hash__(self):
        return hash((self.sym.name, repr(self.t)))

    def __eq__(self, other):
        return isinstance(other, _SymAt) and (self.sym.name, repr(self.t)) == (other.sym.name, repr(other.t))

    def diff(self):
        return _SymAt(_Sym(self.sym.name + "'"), self.t)

    def __repr__(self):
        return f'{self.sym.name}({self.t})'


class InstanceConstraint:
    """`name(t) - value = 0` (what `_g._sx(t0) - x0` evaluates to)."""

    def __init__(self, name, t, value):
        self.name, self.t, self.value = name, float(t), value

    def __repr__(self):
        return f'{self.name}({self.t}) - {self.value}'


class Eom(tuple):
    """The symbolic model of the reference (src/d2d/opty_utils.py:38-50) as data: residual form, wind entering with a + sign
    (the reference's quirk: the plant, src/d2d/dynamic.py:18-19, has the opposite sign).  A tuple of printable equations that
    also carries what the solver needs: the wind vector and g."""

    def __new__(cls, wind, g=9.81, ids=('',)):
        eqs = []
        for i in ids:
            eqs += [f"x{i}' - v{i} cos(psi{i}) + {wind[0]}", f"y{i}' - v{i} sin(psi{i}) + {wind[1]}", f"psi{i}' - {g}/v{i} tan(phi{i})"]
        self = super().__new__(cls, eqs)
        self.wind, self.g, self.n_aircraft = (float(wind[0]), float(wind[1])), g, len(ids)
        return self


class Aircraft:
    """Symbol holder of one aircraft (src/d2d/opty_utils.py:31-50): states (x, y, psi), inputs (v, phi)."""

    def __init__(self, st=None, id=''):
        self._st = st or _Sym('t')
        self._id = id
        self._sx, self._sy, self._sv, self._sphi, self._spsi = (_Sym(f'{n}{id}') for n in ('x', 'y', 'v', 'phi', 'psi'))
        self._state_symbols = (self._sx(self._st), self._sy(self._st), self._spsi(self._st))
        self._input_symbols = (self._sv, self._sphi)

    def get_eom(self, atm, g=9.81):
        """xdot - v cos(psi) + wx, ydot - v sin(psi) + wy, psidot - g/v tan(phi)  (residual form, :42-44)."""
        return Eom(atm.sample_sym(self._st, self._sx(self._st), self._sy(self._st)), g, (self._id,))


# ---------------------------------------------------------------------------------------
# cost plug-ins.  s = obj_scale / num_nodes throughout.
# ---------------------------------------------------------------------------------------
class CostAirVel:
    """s * sum (v - vsp)^2  (src/d2d/opty_utils.py:55-66)."""

    def __init__(self, vsp=10.):
        self.vsp = vsp

    def cost(self, free, _p):
        return _p.obj_scale * np.sum((free[_p._slice_v] - self.vsp) ** 2) / _p.num_nodes

    def cost_grad(self, free, _p):
        g = np.zeros_like(free)
        g[_p._slice_v] = _p.obj_scale / _p.num_nodes * 2 * (free[_p._slice_v] - self.vsp)
        return g


class CostBank:
    """Mean (default) or max squared bank (src/d2d/opty_utils.py:68-82)."""
    use_mean = True

    def cost(self, free, _p):
        sq = free[_p._slice_phi] ** 2
        return _p.obj_scale * (np.sum(sq) / _p.num_nodes if self.use_mean else np.max(sq))

    def cost_grad(self, free, _p):
        g = np.zeros_like(free)
        ph = free[_p._slice_phi]
        if self.use_mean:
            g[_p._slice_phi] = _p.obj_scale / _p.num_nodes * 2 * ph
        else:
            i = np.argmax(ph ** 2)
            g[_p._slice_phi][i] = _p.obj_scale * 2 * ph[i]
        return g


class CostInput:
    """s * (kv sum (v-vsp)^2 + kphi sum phi^2)  (src/d2d/opty_utils.py:85-97)."""

    def __init__(self, vsp=10., kvel=1., kbank=1.):
        self.vsp, self.kv, self.kphi = vsp, kvel, kbank

    def cost(self, free, _p):
        return _p.obj_scale / _p.num_nodes * (self.kv * np.sum((free[_p._slice_v] - self.vsp) ** 2)
                                              + self.kphi * np.sum(free[_p._slice_phi] ** 2))

    def cost_grad(self, free, _p):
        g = np.zeros_like(free)
        g[_p._slice_phi] = self.kphi * 2 * free[_p._slice_phi]
        g[_p._slice_v] = self.kv * 2 * (free[_p._slice_v] - self.vsp)
        return g * (_p.obj_scale / _p.num_nodes)


def _obstacle_field(dx, dy, r, kind, k):
    if kind == 0:
        return np.clip(np.exp(r ** 2 - (dx ** 2 + dy ** 2)), 0., 1e3)
    return np.exp(-((dx / r * k) ** 2 + (dy / r * k) ** 2))


class CostObstacle:
    """Circular obstacle penalty, kind 0 (sharp) or 1 (Gaussian, k=2)  (src/d2d/opty_utils.py:99-134)."""

    def __init__(self, c=(30, 0), r=15., kind=0):
        self.c, self.r, self.kind, self.k = c, r, kind, 2.

    def _d(self, free, _p):
        return free[_p._slice_x] - self.c[0], free[_p._slice_y] - self.c[1]

    def cost1(self, free, _p):
        return _obstacle_field(*self._d(free, _p), self.r, self.kind, self.k)

    def cost(self, free, _p):
        return _p.obj_scale / _p.num_nodes * np.sum(self.cost1(free, _p))

    def cost_grad(self, free, _p):
        dx, dy = self._d(free, _p)
        e = _obstacle_field(dx, dy, self.r, self.kind, self.k)
        g = np.zeros_like(free)
        g[_p._slice_x] = _p.obj_scale / _p.num_nodes * -2. * dx * e      # reference's expression (:131-132)
        g[_p._slice_y] = _p.obj_scale / _p.num_nodes * -2. * dy * e
        return g


class CostObstacles:
    def __init__(self, obss, kind=0):
        self.obss = [CostObstacle(c=(o[0], o[1]), r=o[2], kind=kind) for o in obss]

    def cost(self, free, _p):
        return np.sum([c.cost(free, _p) for c in self.obss])

    def cost_grad(self, free, _p):
        return np.sum([c.cost_grad(free, _p) for c in self.obss], axis=0)


class CostComposit:
    """kobs * obstacles + kvel * air speed + kbank * bank (src/d2d/opty_utils.py:147-165);
    obss=None leaves the obstacle term out, as the reference's try/except does."""

    def __init__(self, obss, vsp=10., kobs=1., kvel=1., kbank=1., obs_kind=0):
        self.kobs, self.kvel, self.kbank = kobs, kvel, kbank
        if obss is not None:
            self.cobs = CostObstacles(obss, obs_kind)
        self.cvel = CostAirVel(vsp)
        self.cbank = CostBank()

    def _terms(self):
        t = [(self.kvel, self.cvel), (self.kbank, self.cbank)]
        if hasattr(self, 'cobs'):
            t.insert(0, (self.kobs, self.cobs))
        return t

    def cost(self, free, _p):
        return sum(k * c.cost(free, _p) for k, c in self._terms())

    def cost_grad(self, free, _p):
        return sum(k * c.cost_grad(free, _p) for k, c in self._terms())


def triangle(p0, p1, va, duration, num_nodes, go_left=1.):
    """Dog-leg initial guess of length va*duration (src/d2d/opty_utils.py:171-187) -> x, y, psi, phi, v."""
    p0, p1 = np.asarray(p0, dtype=float), np.asarray(p1, dtype=float)
    leg = p1 - p0
    d = np.linalg.norm(leg)
    nrm = np.array([-leg[1], leg[0]]) / d
    D = va * duration
    apex = p0 + leg / 2
    if D > d:
        apex = apex + np.sign(go_left) * np.sqrt(D ** 2 - d ** 2) / 2 * nrm
    n1 = int(num_nodes / 2); n2 = num_nodes - n1
    pts = np.vstack((np.linspace(p0, apex, n1), np.linspace(apex, p1, n2)))
    h0, h1 = apex - p0, p1 - apex
    psis = np.hstack((np.arctan2(h0[1], h0[0]) * np.ones(n1), np.arctan2(h1[1], h1[0]) * np.ones(n2)))
    return pts[:, 0], pts[:, 1], psis, np.zeros(num_nodes), va * np.ones(num_nodes)
