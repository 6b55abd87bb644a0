"""CPU: host-side mirror of the reference's interface (cost plug-ins, guesses, polynomial
primitives, scenario protocol, cost lowering) against the golden vectors."""
import types

import numpy as np
import pytest

import d2d.opty_utils as d2ou
import d2d.multiopty_utils as d2mou
import d2d.trajectory as ddt
import d2d.optyplan_scenarios as d2oscen
import single_opt_planner as sop
import multi_opt_planner as mop


class FakeSingle:                 # the reference's own fake-planner pattern (src/test/test_objective.py:11-16)
    def __init__(self, N, obj_scale):
        self.num_nodes, self.obj_scale = N, obj_scale
        self._slice_x, self._slice_y, self._slice_psi, self._slice_phi, self._slice_v = (
            slice(i * N, (i + 1) * N, 1) for i in range(5))


class FakeMulti:
    def __init__(self, N, n, obj_scale):
        self.num_nodes, self.obj_scale = N, obj_scale
        self.acs = types.SimpleNamespace(nb_aicraft=n)
        self._slice_x = [slice((0 + 3 * i) * N, (1 + 3 * i) * N, 1) for i in range(n)]
        self._slice_y = [slice((1 + 3 * i) * N, (2 + 3 * i) * N, 1) for i in range(n)]
        self._slice_psi = [slice((2 + 3 * i) * N, (3 + 3 * i) * N, 1) for i in range(n)]
        o = 3 * n * N
        self._slice_phi = [slice(o + i * N, o + (i + 1) * N, 1) for i in range(n)]
        o += n * N
        self._slice_v = [slice(o + i * N, o + (i + 1) * N, 1) for i in range(n)]


def test_single_cost_plugins(gold):
    g = gold('costs')
    N = int(g['s_N']); p = FakeSingle(N, float(g['s_obj_scale'])); f = g['s_free']; obss = [tuple(o) for o in g['obss']]
    bmax = d2ou.CostBank(); bmax.use_mean = False
    cases = {'s_airvel': d2ou.CostAirVel(12.0), 's_bank_mean': d2ou.CostBank(), 's_bank_max': bmax,
             's_input': d2ou.CostInput(12.0, 5.0, 1.5), 's_obst_k0': d2ou.CostObstacle((30.0, 0.0), 15.0, 0),
             's_obst_k1': d2ou.CostObstacle((30.0, 0.0), 15.0, 1), 's_obsts_k1': d2ou.CostObstacles(obss, 1),
             's_obsts_k0': d2ou.CostObstacles(obss, 0),
             's_composit_k1': d2ou.CostComposit(obss, 11.0, kobs=2.0, kvel=0.5, kbank=3.0, obs_kind=1),
             's_composit_none': d2ou.CostComposit(None, 11.0, kobs=0.0, kvel=0.1, kbank=10.0)}
    for k, c in cases.items():
        np.testing.assert_allclose(c.cost(f, p), g[k + '_cost'], rtol=1e-13, atol=1e-15, err_msg=k)
        np.testing.assert_allclose(c.cost_grad(f, p), g[k + '_grad'], rtol=1e-13, atol=1e-15, err_msg=k)


def test_three_obstacle_costs(gold):
    """The obstacle list of the reference's exp_6 / exp_7 (three discs) through the mirror's plug-ins, both kinds."""
    g = gold('costs_obs3')
    N = int(g['N']); p = FakeSingle(N, float(g['obj_scale'])); f = g['free']; obss = [tuple(o) for o in g['obss']]
    assert len(obss) == 3
    cases = {'obsts_k1': d2ou.CostObstacles(obss, 1), 'obsts_k0': d2ou.CostObstacles(obss, 0),
             'composit_k1': d2ou.CostComposit(obss, 12.0, kobs=0.5, kvel=10.0, kbank=1.0, obs_kind=1),
             'composit_k0': d2ou.CostComposit(obss, 12.0, kobs=0.5, kvel=10.0, kbank=1.0, obs_kind=0)}
    for k, c in cases.items():
        np.testing.assert_allclose(c.cost(f, p), g[k + '_cost'], rtol=1e-13, atol=1e-15, err_msg=k)
        np.testing.assert_allclose(c.cost_grad(f, p), g[k + '_grad'], rtol=1e-13, atol=1e-15, err_msg=k)
    # lowering: three and four obstacles land in the extension columns with their kind bits
    import d2dhip
    low = sop.lower_cost(cases['composit_k0'])
    row = sop.scen_row((0, 0, 0, 0, 12), (100, 0, 0, 0, 12), 12., low, 0.01, [0., 0.], (-0.6, 0.6), (9., 15.))
    assert tuple(row[d2dhip.SC_O0X:d2dhip.SC_O0X + 3]) == obss[0] and tuple(row[d2dhip.SC_O1X:d2dhip.SC_O1X + 3]) == obss[1]
    c2 = d2dhip.obs_col(2)
    assert c2 == d2dhip.SC_OEXT and tuple(row[c2:c2 + 3]) == obss[2] and not row[c2 + 3:].any() and row[d2dhip.SC_OKIND] == 0b111
    many = d2ou.CostComposit([(float(i), 2., 3.) for i in range(d2dhip.MAX_OBS + 1)], 12.0, kobs=0.5, kvel=10.0, kbank=1.0, obs_kind=1)
    with pytest.raises(NotImplementedError):
        sop.scen_row((0, 0, 0, 0, 12), (100, 0, 0, 0, 12), 12., sop.lower_cost(many), 0.01, [0., 0.], (-0.6, 0.6), (9., 15.))


def test_multi_cost_plugins(gold):
    g = gold('costs')
    N = int(g['s_N']); n = int(g['m_n']); p = FakeMulti(N, n, float(g['m_obj_scale'])); f = g['m_free']
    obss = [tuple(o) for o in g['obss']]; nan = float('NaN')
    cases = {'m_null': d2mou.CostNull(), 'm_airvel': d2mou.CostAirvel(12.0), 'm_bank': d2mou.CostBank(),
             'm_input': d2mou.CostInput(12.0, 5.0, 1.0), 'm_obst_k0': d2mou.CostObstacle((30.0, 0.0), 15.0, 0),
             'm_obst_k1': d2mou.CostObstacle((30.0, 0.0), 15.0, 1), 'm_obsts_k1': d2mou.CostObstacles(obss, 1),
             'm_collision': d2mou.CostCollision(r=10.0, k=2.0),
             'm_composit_nan': d2mou.CostComposit(kvel=70.0, kbank=1.0, kobs=nan, kcol=nan, vsp=12.0, obss=[], obs_kind=0, rcol=3.0),
             'm_composit_col': d2mou.CostComposit(kvel=70.0, kbank=1.0, kobs=nan, kcol=10.0, vsp=12.0, obss=[], obs_kind=0, rcol=10.0),
             'm_composit_all': d2mou.CostComposit(kvel=5.0, kbank=1.0, kobs=2.0, kcol=10.0, vsp=12.0, obss=obss, obs_kind=1, rcol=10.0)}
    for k, c in cases.items():
        np.testing.assert_allclose(c.cost(f, p), g[k + '_cost'], rtol=1e-13, atol=1e-15, err_msg=k)
        np.testing.assert_allclose(c.cost_grad(f, p), g[k + '_grad'], rtol=1e-13, atol=1e-15, err_msg=k)


def test_committed_solver_outputs(gold):
    """The reference's committed IPOPT outputs through the mirror's plug-ins."""
    g = gold('planner_goldens')
    N = len(g['exp14_time'])
    np.testing.assert_allclose(d2ou.CostAirVel(12.0).cost(g['exp14_free'], FakeSingle(N, 1.0)), 5.02972817, rtol=1e-8)
    Nm = len(g['stline_time'])
    cc = d2mou.CostComposit(kvel=70., kbank=1., kobs=float('NaN'), kcol=10., vsp=12., obss=[], obs_kind=0, rcol=10.)
    np.testing.assert_allclose(cc.cost(g['stline_free'], FakeMulti(Nm, 4, 1.0)), 0.2024378405, rtol=1e-9)
    np.testing.assert_allclose(np.linalg.norm(cc.cost_grad(g['stline_free'], FakeMulti(Nm, 4, 1.0))), g['stline_grad_norm'], rtol=1e-12)


def test_timing_triangle_guesses(gold, capsys):
    g = gold('guess_poly')
    for r, o in zip(g['timing_in'], g['timing_out']):
        np.testing.assert_allclose(d2ou.planner_timing(*r), o, rtol=1e-15)
    assert 'nodes' in capsys.readouterr().out           # the reference prints its timing line
    for i, r in enumerate(g['tri_in']):
        np.testing.assert_allclose(np.array(d2ou.triangle(r[0:2], r[2:4], r[4], r[5], int(r[6]), r[7])), g[f'tri_out_{i}'],
                                   rtol=1e-14, atol=1e-13)
    p = sop.Planner(d2oscen.exp_14, initialize=True)
    assert p.prob.num_free == 5 * 121
    np.testing.assert_allclose(p.get_initial_guess('tri'), g['single_exp14_tri'], rtol=1e-14, atol=1e-13)
    np.testing.assert_allclose(p.get_initial_guess('line'), g['single_exp14_line'], rtol=1e-14, atol=1e-13)
    scen = mop.trap_4
    scen.t1 = 7.0
    scen.p0s = tuple(map(tuple, g['multi_trap4_p0s'])); scen.p1s = tuple(map(tuple, g['multi_trap4_p1s']))
    mp = mop.Planner(scen, initialize=True)
    assert mp.num_nodes == int(g['multi_trap4_num_nodes']) and mp.prob.num_free == 5 * 4 * mp.num_nodes
    np.testing.assert_allclose(mp.get_initial_guess('tri'), g['multi_trap4_tri'], rtol=1e-14, atol=1e-13)
    rnd = mp.get_initial_guess('rnd')
    assert rnd.shape == (mp.prob.num_free,) and np.abs(rnd[mp._slice_y[0]]).max() <= 100


def test_polynomial_primitives(gold):
    g = gold('guess_poly')
    pol = ddt.PolynomialOne([0, .05, 0, 0], [1, .05, 0, 0], 10)
    np.testing.assert_allclose(pol.coefs, g['poly_ka_coefs'], rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose(pol.get(3.3), g['poly_ka_get33'], rtol=1e-13)
    for i in range(len(g['poly_T'])):
        p = ddt.PolynomialOne(g['poly_Y0'][i], g['poly_Y1'][i], g['poly_T'][i])
        np.testing.assert_allclose(p.coefs, g['poly_coefs'][i], rtol=1e-9, atol=1e-12)
        for j, t in enumerate(g['poly_t'][i]):
            np.testing.assert_allclose(p.get(t), g['poly_get'][i][j], rtol=1e-9, atol=1e-10)
        q = ddt.PolynomialOne.from_coefs(p.coefs[0], g['poly_T'][i])
        np.testing.assert_array_equal(q.coefs, p.coefs)
    # composite: segment lookup and wrap-around
    Y = np.zeros((3, 2, 4)); Y[1, :, 0] = [1.0, 2.0]; Y[2, :, 0] = [3.0, -1.0]; Y[:, :, 1] = 1.0
    ct = ddt.CompositeTraj([ddt.MinSnapPoly(Y[0], Y[1], 2.0), ddt.MinSnapPoly(Y[1], Y[2], 1.0)])
    assert ct.duration == 3.0
    np.testing.assert_allclose(ct.get(2.0)[0], [1.0, 2.0], atol=1e-12)
    np.testing.assert_allclose(ct.get(2.5 + 3.0), ct.get(2.5), atol=1e-12)


def test_cost_lowering_and_scenarios():
    nan = float('nan')
    assert sop.lower_cost(d2ou.CostAirVel(12.))[:4] == (12., 1., 0., 0.)
    assert sop.lower_cost(d2ou.CostInput(11., 5., 2.))[:3] == (11., 5., 2.)
    low = sop.lower_cost(d2ou.CostComposit([(1, 2, 3)], 10., kobs=2., kvel=.5, kbank=3., obs_kind=1))
    assert low[:5] == (10., .5, 3., 2., ((1, 2, 3),))
    low = sop.lower_cost(d2mou.CostComposit(kvel=70., kbank=1., kobs=nan, kcol=10., vsp=12., rcol=10.))
    assert low[:3] == (12., 70., 1.) and low[5] == 10. and low[6] == 10.
    bmax = d2ou.CostBank(); bmax.use_mean = False
    assert sop.lower_cost(bmax)[7:] == (0, 1) and sop.lower_cost(d2ou.CostBank())[7:] == (0, 0)
    low = sop.lower_cost(d2ou.CostComposit([(1, 2, 3), (4, 5, 6)], obs_kind=0))
    assert low[7:] == (0b11, 0) and low[4] == ((1, 2, 3), (4, 5, 6))
    assert sop.lower_cost(d2mou.CostComposit(kobs=1., obss=[(1, 2, 3)], obs_kind=0))[7] == 1
    with pytest.raises(NotImplementedError):
        sop.lower_cost(object())
    # scenario protocol
    for s in d2oscen.scens:
        for attr in ('name', 'desc', 't0', 't1', 'hz', 'p0', 'p1', 'wind', 'cost', 'obj_scale', 'x_constraint',
                     'y_constraint', 'phi_constraint', 'v_constraint', 'obstacles', 'vref', 'tol', 'max_iter', 'ncases'):
            assert hasattr(s, attr), (s.name, attr)
        s.set_case(0); s.label(0)
    assert 'exp0' in d2oscen.desc_all() and 'final state' in d2oscen.desc_one(0)
    d2oscen.exp_0_1.set_case(2)
    assert d2oscen.exp_0.t1 == 15.                       # set_case mutates the base class, as the reference
    d2oscen.exp_0.t1 = 10.
    row = sop.scen_row((0, 0, 0, 0, 10), (0, 30, np.pi, 0, 10), 12., sop.lower_cost(d2ou.CostAirVel(12.)), 0.01,
                       [1.0, -2.0], (-0.5, 0.5), (9., 14.))
    import d2dhip
    assert row[d2dhip.SC_WX] == -1.0 and row[d2dhip.SC_WY] == 2.0 and row[d2dhip.SC_PHIMAX] == 0.5
    assert row[d2dhip.SC_VMIN] == 9. and row[d2dhip.SC_VMAX] == 14. and row.shape == (d2dhip.SCEN_STRIDE,)
    row = sop.scen_row((0, 0, 0, 0, 10), (0, 30, np.pi, 0, 10), 12., sop.lower_cost(d2ou.CostComposit([(1, 2, 3)], obs_kind=0)),
                       0.01, [0., 0.], (-0.5, 0.5), (9., 14.))
    assert row[d2dhip.SC_OKIND] == 1 and row[d2dhip.SC_BANKMAX] == 0 and row[d2dhip.SC_O0R] == 3


def test_plan_csv_round_trip(tmp_path):
    """multi_opt_planner.export_csv writes the column format of src/07_multioptyplan.py:476-489 and
    full_sim.ExtractTrajData (src/11_full_sim_case1.py:206-217) reads it back; ExtendTraj_symm mirrors it."""
    import pandas as pd
    import full_sim as fs
    T, n = 11, 4
    p = types.SimpleNamespace(sol_time=np.linspace(0, 1, T))
    rng = np.random.default_rng(2)
    for k in ('x', 'y', 'psi', 'phi', 'v'):
        setattr(p, 'sol_' + k, [rng.normal(size=T) for _ in range(n)])
    # a symmetric pair structure: aircraft i ends where aircraft n-1-i starts
    for i in range(n):
        p.sol_x[i][-1], p.sol_y[i][-1] = p.sol_x[n - 1 - i][0], p.sol_y[n - 1 - i][0]
    f = tmp_path / 'plan.csv'
    mop.export_csv(p, f)
    df = pd.read_csv(f)
    assert list(df.columns)[:6] == ['time', 'x_1', 'y_1', 'psi_1', 'phi_1', 'v_1'] and len(df.columns) == 1 + 5 * n
    t, x, y, psi = fs.ExtractTrajData(df, n)
    np.testing.assert_allclose(x, np.stack(p.sol_x, 1), rtol=1e-14); np.testing.assert_allclose(psi, np.stack(p.sol_psi, 1), rtol=1e-14)   # (decimal text)
    t2, x2, y2, psi2 = fs.ExtendTraj_symm(n, x, y, psi, t)
    assert x2.shape == (2 * T, n) and t2[-1] == 2 * t[-1]
    for i in range(n):
        np.testing.assert_array_equal(x2[T:, i], x[:, n - 1 - i])
        np.testing.assert_array_equal(psi2[T:, i], y[:, n - 1 - i])      # (the reference extends psi with y, :238)


def test_trajectory_factory_and_catalogues_vs_reference(gold):
    """Demo trajectories (src/d2d/trajectory_factory.py) against the reference's traj.get(t) at seeded times, beyond one period of
    the composites; the planner scenario catalogues (single: src/d2d/optyplan_scenarios.py, 15 entries; multi:
    src/07_multioptyplan.py:170-435, 15 entries) against the reference's names and numeric attributes; the simulation scenarios
    whose start states are given explicitly (no device call in their construction)."""
    import d2d.trajectory_factory as ddtf
    import d2d.optyplan_scenarios as d2oscen
    import d2d.multioptyplan_scenarios as d2mscen
    import d2d.scenario as dds
    g = gold('traj_scen')
    for name in ('circle', 'two_lines', 'square', 'line_with_intro', 'demo_minsnap', 'slalom', 'sidemo'):
        traj, desc = ddtf.get(name)
        assert abs(traj.duration - float(g[f'traj_{name}_duration'])) < 1e-12
        Y = np.array([traj.get(t) for t in g[f'traj_{name}_t']])
        np.testing.assert_allclose(Y, g[f'traj_{name}_Y'], rtol=1e-12, atol=1e-10, err_msg=name)
    assert len(ddtf.list_available()) == 10 and len(dds.list_available()) == 13
    import importlib
    importlib.reload(d2oscen); importlib.reload(d2mscen)       # (other tests mutate base classes through set_case, as the protocol allows)
    assert len(d2oscen.scens) == 15 and len(d2mscen.scens) == 15
    for i, sc in enumerate(d2oscen.scens):
        assert sc.name == str(g[f'plan_{i}_name']), (i, sc.name)
        num = [sc.t0, sc.t1, sc.hz, sc.obj_scale, sc.vref, sc.ncases, len(sc.obstacles)] + list(sc.p0) + list(sc.p1) + list(sc.phi_constraint) + list(sc.v_constraint)
        np.testing.assert_allclose(num, g[f'plan_{i}_num'], rtol=0, atol=1e-15, err_msg=sc.name)
    for i, sc in enumerate(d2mscen.scens):
        assert sc.name == str(g[f'mplan_{i}_name']), (i, sc.name)
        num = ([sc.t0, sc.t1, sc.hz, sc.obj_scale, sc.vref, sc.ncases, len(sc.obstacles), len(sc.p0s)] + list(np.ravel(sc.p0s)) + list(np.ravel(sc.p1s))
               + list(sc.phi_constraint) + list(sc.v_constraint))
        np.testing.assert_allclose(num, g[f'mplan_{i}_num'], rtol=0, atol=1e-15, err_msg=sc.name)
    for sc in d2mscen.scens:                                   # (set_case mutates class attributes of base classes: after the comparison)
        for c in range(sc.ncases):
            sc.set_case(c); sc.label(c)
            sop.lower_cost(sc.cost)                            # every cost of the catalogue has a kernel lowering
    for name in ('line', 'line2', 'square', 'mucir', 'mucir2', 'patrol', 'patrol_2'):
        scen, _ = dds.get(name)
        np.testing.assert_allclose(np.array(scen.X0s, dtype=float), g[f'scen_{name}_X0s'], atol=1e-15)
        np.testing.assert_allclose([scen.time[0], scen.time[-1], len(scen.time)], g[f'scen_{name}_time'], atol=1e-12)
        np.testing.assert_allclose(scen.windfield.sample(0., [0., 0.]), g[f'scen_{name}_wind'])
        np.testing.assert_allclose(scen.extends, g[f'scen_{name}_extends'], atol=1e-12)
        Y = np.array([[traj.get(t) for traj in scen.trajs] for t in g[f'scen_{name}_ts']])
        np.testing.assert_allclose(Y, g[f'scen_{name}_Y'], rtol=1e-12, atol=1e-10, err_msg=name)
        nz = np.array([[i, j, k, p[j, k]] for i, p in enumerate(scen.perts) for j, k in zip(*np.nonzero(p))], dtype=float).reshape(-1, 4)
        np.testing.assert_allclose(nz, g[f'scen_{name}_pert_nz'])
