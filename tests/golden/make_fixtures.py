#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (this container only).

Run:  cd /root/repo && MPLBACKEND=Agg python3 -B tests/golden/make_fixtures.py

/root/reference never travels to the GPU box; only the numeric arrays written here do.
Two third-party modules the reference imports are absent from the image (`opty`,
`control`).  They are replaced IN MEMORY ONLY so that the reference's own modules
import:
  * opty.direct_collocation.Problem -> a dummy that only records num_free (the planner
    constructors need nothing else; nothing here calls .solve()).
  * control.lqr -> scipy.linalg.solve_continuous_are (the stabilising CARE solution is
    unique).  Every fixture that went through it is named *_carestandin.
No reference source text is stored: the files hold inputs and outputs only.
"""
import os
import sys
import types

import numpy as np
import scipy.linalg

REF = '/root/reference/src'
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
os.chdir(REF)
os.environ.setdefault('MPLBACKEND', 'Agg')
sys.dont_write_bytecode = True

# ---- in-memory stand-ins for the two absent third-party modules -------------------
opty = types.ModuleType('opty'); dc = types.ModuleType('opty.direct_collocation')


class Problem:
    def __init__(self, obj, obj_grad, eom, state_symbols, num_nodes, time_step, **kw):
        self.num_free = 5 * num_nodes * (len(state_symbols) // 3)


dc.Problem = Problem; opty.direct_collocation = dc
sys.modules['opty'] = opty; sys.modules['opty.direct_collocation'] = dc
control = types.ModuleType('control')


def _lqr(A, B, Q, R):
    P = scipy.linalg.solve_continuous_are(A, B, Q, R)
    K = np.linalg.solve(R, B.T @ P)
    return K, P, np.linalg.eigvals(A - B @ K)


control.lqr = _lqr; sys.modules['control'] = control

import contextlib, io                                                    # noqa: E402
import pandas as pd                                                      # noqa: E402
import d2d.opty_utils as d2ou                                            # noqa: E402
import d2d.multiopty_utils as d2mou                                      # noqa: E402
import d2d.dynamic as ddyn                                               # noqa: E402
import d2d.guidance as ddg                                               # noqa: E402
import d2d.trajectory as ddt                                             # noqa: E402
import d2d.utils as d2u                                                  # noqa: E402
import Controllers as tracking                                           # noqa: E402
import single_opt_planner, multi_opt_planner                             # noqa: E402

from oracle import fit as ofit                                           # noqa: E402


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


rng = np.random.default_rng(0)


# ----------------------------------------------------------------------------------
def fx_plant():
    out = {}
    n = 48
    X = np.stack([rng.uniform(-100, 100, n), rng.uniform(-100, 100, n), rng.uniform(-np.pi, np.pi, n),
                  rng.uniform(-0.7, 0.7, n), rng.uniform(8, 16, n)], 1)
    X[0, 2] = np.pi - 1e-3; X[1, 2] = -np.pi + 1e-3          # heading wrap cases
    U = np.stack([rng.uniform(-1.0, 1.0, n), rng.uniform(9, 16, n)], 1)
    U[0, 0] = 0.9; U[1, 0] = -0.9
    W = np.stack([rng.uniform(-3, 3, n), rng.uniform(-3, 3, n)], 1); W[: n // 2] = 0.0
    out['X'], out['U'], out['W'], out['dt'] = X, U, W, 0.05
    for tau in (0.01, 0.9667):
        ac = ddyn.Aircraft(); ac.tau_phi = tau
        Y = np.array([ac.disc_dyn(X[i], U[i], ddg.WindField(list(W[i])), 0.3, 0.05) for i in range(n)])
        out[f'Xnext_tau{tau}'] = Y
    ac = ddyn.Aircraft()
    AB = [ac.cont_jac(X[i], U[i], 0.0, None) for i in range(n)]
    out['A'] = np.array([a for a, _ in AB]); out['B'] = np.array([b for _, b in AB])
    out['cont_dyn'] = np.array([ac.cont_dyn(X[i], 0.0, U[i], ddg.WindField(list(W[i]))) for i in range(n)])
    out['norm_mpi_pi_in'] = np.array([-7.0, -np.pi, -3.0, 0.0, 3.0, np.pi, 7.0, 100.0])
    out['norm_mpi_pi_out'] = d2u.norm_mpi_pi(out['norm_mpi_pi_in'])
    out['known_answer_disc_dyn'] = ddyn.Aircraft().disc_dyn([20, 30, -np.pi / 2, 0, 10], [0.1, 15], ddg.WindField(), 0, 0.05)
    np.savez(os.path.join(OUT, 'plant.npz'), **out)


def fx_flatness_ctrl():
    out = {}
    n = 40
    Y = rng.uniform(-80, 80, (n, 2)); psi = rng.uniform(-np.pi, np.pi, n); sp = rng.uniform(9, 15, n)
    Yd = np.stack([sp * np.cos(psi), sp * np.sin(psi)], 1)
    Ydd = rng.uniform(-4, 4, (n, 2)); Yddd = rng.uniform(-1, 1, (n, 2)); Yddd[: n // 2] = 0.0
    W = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n)], 1); W[: n // 2] = 0.0
    out.update(Y=Y, Yd=Yd, Ydd=Ydd, Yddd=Yddd, W=W)
    ac = ddyn.Aircraft()
    res = [ddg.DiffFlatness.state_and_input_from_output(np.array([Y[i], Yd[i], Ydd[i], Yddd[i]]), W[i], ac) for i in range(n)]
    out['g_X'] = np.array([r[0] for r in res]); out['g_U'] = np.array([r[1] for r in res]); out['g_Xdot'] = np.array([r[2] for r in res])
    res = [tracking.DiffFlatness(list(W[i])).ComputeFlatness(0.0, Y[i], Yd[i], Ydd[i], Yddd[i]) for i in range(n)]
    out['c_X'] = np.array([r[0] for r in res]); out['c_U'] = np.array([r[1] for r in res])
    # ComputeGain (CARE stand-in)
    X = out['c_X'] + np.stack([rng.uniform(-25, 25, n), rng.uniform(-25, 25, n), rng.uniform(-1.5, 1.5, n),
                               rng.uniform(-1.0, 1.0, n), rng.uniform(-2, 2, n)], 1)
    X[0, 2] = out['c_X'][0, 2] + 2 * np.pi - 0.2              # wrap of dpsi
    out['X'] = X
    Xr, dX, Uc, K = [], [], [], []
    for i in range(n):
        ctrl = tracking.DiffController(list(W[i]))
        a, b, c = ctrl.ComputeGain(0.0, X[i].copy(), Y[i], Yd[i], Ydd[i], Yddd[i], ac)
        Xr.append(a); dX.append(b); Uc.append(c); K.append(ctrl.K[-1])
    out['gain_Xr_carestandin'] = np.array(Xr); out['gain_dX_carestandin'] = np.array(dX)
    out['gain_U_carestandin'] = np.array(Uc); out['gain_K_carestandin'] = np.array(K)
    np.savez(os.path.join(OUT, 'flatness_ctrl.npz'), **out)


def fx_guidance():
    out = {}
    n_ac = 4
    B = np.zeros((n_ac, n_ac - 1))
    for i in range(n_ac - 1):
        B[i, i] = -1; B[i + 1, i] = 1
    ncase = 24
    c = rng.uniform(-50, 50, (ncase, n_ac, 2)); p = rng.uniform(-120, 120, (ncase, 2, n_ac))
    zd = rng.uniform(-1, 1, (ncase, n_ac - 1)); zd[: ncase // 2] = 0.0
    Ur, eth = [], []
    dcf = ddg.DCFController()
    for i in range(ncase):
        u, e = dcf.get(n_ac, B, c[i], p[i], zd[i].copy(), 20.0)
        Ur.append(u[:, 0]); eth.append(e[:, 0])
    out.update(B=B, dcf_c=c, dcf_p=p, dcf_zdes=zd, dcf_Ur=np.array(Ur), dcf_etheta_deg=np.array(eth), dcf_kr=20.0)
    n = 40
    X = np.stack([rng.uniform(-100, 100, n), rng.uniform(-100, 100, n), rng.uniform(-np.pi, np.pi, n),
                  rng.uniform(-0.5, 0.5, n), rng.uniform(8, 16, n)], 1)
    cc = rng.uniform(-30, 30, (n, 2)); r = rng.uniform(30, 80, n)
    ke, kd = 4e-4, 25.0
    E, N, U, U1, U2 = [], [], [], [], []
    for i in range(n):
        e, nn, H = ddg.CircleTraj(cc[i]).get(X[i], r[i])
        u, u1, u2 = ddg.GVFcontroller(None, None, None).get(X[i], ke, kd, e, nn, H)
        E.append(e); N.append(nn); U.append(u); U1.append(u1); U2.append(u2)
    out.update(gvf_X=X, gvf_c=cc, gvf_r=r, gvf_ke=ke, gvf_kd=kd, gvf_e=np.array(E), gvf_n=np.array(N),
               gvf_U=np.array(U), gvf_U1=np.array(U1), gvf_U2=np.array(U2))
    np.savez(os.path.join(OUT, 'guidance.npz'), **out)


def fx_states_over_time():
    df = pd.read_csv(os.path.join(REF, 'states_over_time.csv'))
    a = df.values
    idx = sorted(set(range(0, 402)) | {i for m in range(4, 40) for i in (100 * m, 100 * m + 1)} | {3998, 3999})
    idx = np.array(idx)
    np.savez(os.path.join(OUT, 'states_over_time_sub.npz'), rows=idx, time=a[idx, 0],
             X=a[idx, 1:].reshape(len(idx), 4, 5),
             centres=np.array([[0, -20], [25, -40], [25, -80], [0, -100.0]]),
             r=60.0, v_c=15.0, ke=4e-4, kd=25.0, kr=20.0, dt=0.05, tau_phi=0.9667, tau_v=1.0)


class _FakeSingle:
    def __init__(self, N, obj_scale):
        self.num_nodes, self.obj_scale = N, obj_scale
        self._slice_x, self._slice_y, self._slice_psi, self._slice_phi, self._slice_v = (
            slice(i * N, (i + 1) * N, 1) for i in range(5))


class _FakeMulti:
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


def fx_costs():
    out = {}
    N = 50
    p = _FakeSingle(N, 0.7)
    free = np.concatenate([rng.uniform(-40, 60, N), rng.uniform(-40, 60, N), rng.uniform(-3, 3, N),
                           rng.uniform(-0.7, 0.7, N), rng.uniform(9, 15, N)])
    out['s_free'] = free; out['s_obj_scale'] = 0.7; out['s_N'] = N
    obss = [(30.0, 0.0, 15.0), (5.0, 25.0, 8.0)]
    out['obss'] = np.array(obss)
    cb_max = d2ou.CostBank(); cb_max.use_mean = False
    cases = {
        's_airvel': d2ou.CostAirVel(12.0), 's_bank_mean': d2ou.CostBank(), 's_bank_max': cb_max,
        's_input': d2ou.CostInput(12.0, 5.0, 1.5),
        's_obst_k0': d2ou.CostObstacle((30.0, 0.0), 15.0, 0), 's_obst_k1': d2ou.CostObstacle((30.0, 0.0), 15.0, 1),
        's_obsts_k1': d2ou.CostObstacles(obss, 1), 's_obsts_k0': d2ou.CostObstacles(obss, 0),
        's_composit_k1': d2ou.CostComposit(obss, 11.0, kobs=2.0, kvel=0.5, kbank=3.0, obs_kind=1),
        's_composit_none': d2ou.CostComposit(None, 11.0, kobs=0.0, kvel=0.1, kbank=10.0),
    }
    for k, cobj in cases.items():
        out[k + '_cost'] = cobj.cost(free, p); out[k + '_grad'] = cobj.cost_grad(free, p)
    n = 4
    pm = _FakeMulti(N, n, 1.3)
    fm = np.concatenate([np.concatenate([rng.uniform(-40, 60, N), rng.uniform(-40, 60, N), rng.uniform(-3, 3, N)]) for _ in range(n)]
                        + [rng.uniform(-0.7, 0.7, n * N), rng.uniform(9, 15, n * N)])
    # make aircraft 0 and 1 close so that the collision term is active
    fm[pm._slice_x[1]] = fm[pm._slice_x[0]] + rng.uniform(-6, 6, N)
    fm[pm._slice_y[1]] = fm[pm._slice_y[0]] + rng.uniform(-6, 6, N)
    out['m_free'] = fm; out['m_obj_scale'] = 1.3; out['m_n'] = n
    nan = float('NaN')
    mcases = {
        'm_null': d2mou.CostNull(), 'm_airvel': d2mou.CostAirvel(12.0), 'm_bank': d2mou.CostBank(),
        'm_input': d2mou.CostInput(12.0, 5.0, 1.0),
        'm_obst_k0': d2mou.CostObstacle((30.0, 0.0), 15.0, 0), 'm_obst_k1': d2mou.CostObstacle((30.0, 0.0), 15.0, 1),
        'm_obsts_k1': d2mou.CostObstacles(obss, 1), 'm_collision': d2mou.CostCollision(r=10.0, k=2.0),
        'm_composit_nan': d2mou.CostComposit(kvel=70.0, kbank=1.0, kobs=nan, kcol=nan, vsp=12.0, obss=[], obs_kind=0, rcol=3.0),
        'm_composit_col': d2mou.CostComposit(kvel=70.0, kbank=1.0, kobs=nan, kcol=10.0, vsp=12.0, obss=[], obs_kind=0, rcol=10.0),
        'm_composit_all': d2mou.CostComposit(kvel=5.0, kbank=1.0, kobs=2.0, kcol=10.0, vsp=12.0, obss=obss, obs_kind=1, rcol=10.0),
    }
    for k, cobj in mcases.items():
        out[k + '_cost'] = cobj.cost(fm, pm); out[k + '_grad'] = cobj.cost_grad(fm, pm)
    np.savez(os.path.join(OUT, 'costs.npz'), **out)


def fx_costs3():
    """Three static obstacles (the list of the reference's exp_6 / exp_7 scenarios, d2d/optyplan_scenarios.py) through the
    reference's CostObstacles / CostComposit, both obstacle kinds.  Own generator: the other fixtures stay byte-identical."""
    r3 = np.random.default_rng(33)
    N = 50
    p = _FakeSingle(N, 0.7)
    obss = [(25.0, 0.0, 15.0), (55.0, 7.5, 12.0), (80.0, -10.0, 12.0)]
    free = np.concatenate([r3.uniform(0, 100, N), r3.uniform(-30, 30, N), r3.uniform(-3, 3, N),
                           r3.uniform(-0.7, 0.7, N), r3.uniform(9, 15, N)])
    out = dict(free=free, obj_scale=0.7, N=N, obss=np.array(obss))
    cases = {'obsts_k1': d2ou.CostObstacles(obss, 1), 'obsts_k0': d2ou.CostObstacles(obss, 0),
             'composit_k1': d2ou.CostComposit(obss, 12.0, kobs=0.5, kvel=10.0, kbank=1.0, obs_kind=1),
             'composit_k0': d2ou.CostComposit(obss, 12.0, kobs=0.5, kvel=10.0, kbank=1.0, obs_kind=0)}
    for k, cobj in cases.items():
        out[k + '_cost'] = cobj.cost(free, p); out[k + '_grad'] = cobj.cost_grad(free, p)
    np.savez(os.path.join(OUT, 'costs_obs3.npz'), **out)


def fx_dfff_run():
    """The legacy simulation loop (run_simulation, src/05_test_simulation.py:21-34) with DFFFController on a composite
    minimum-snap trajectory of the reference's own classes, with a perturbation row, 3-state LQR through the CARE stand-in."""
    r = np.random.default_rng(21)
    ac = ddyn.Aircraft()
    wind = ddg.WindField([0.5, -0.3])
    J = [np.array([[0., 0.], [12., 0.], [0., 0.], [0., 0.]]).T]              # (axis, deriv) junction data
    for j in range(1, 4):
        p = np.array([40. * j, 12. * (-1) ** j]); v = np.array([12., 2. * (-1) ** (j + 1)])
        J.append(np.stack([p, v, r.normal(0, 0.5, 2), r.normal(0, 0.2, 2)], 1))
    T = 3.5
    steps = [ddt.MinSnapPoly(J[j], J[j + 1], T) for j in range(3)]
    traj = ddt.CompositeTraj(steps)
    dt = 0.05
    time = np.arange(0, 3 * T - 1e-9, dt)
    ctl = ddg.DFFFController(traj, ac, wind)
    n = len(time)
    perts = np.zeros((n, 5)); perts[40] = [0.8, -0.5, 0.05, 0.0, 0.3]
    X = np.zeros((n, 5)); U = np.zeros((n, 2))
    Yref = np.array([traj.get(t) for t in time])
    X[0] = [1.0, -2.0, 0.1, 0.0, 11.0]
    for i in range(1, n):
        U[i - 1] = ctl.get(X[i - 1].copy(), time[i - 1])
        X[i] = ac.disc_dyn(X[i - 1], U[i - 1], wind, time[i - 1], time[i] - time[i - 1])
        X[i] += perts[i]
    U[-1] = ctl.get(X[-1].copy(), time[-1])
    np.savez(os.path.join(OUT, 'dfff_run_carestandin.npz'), time=time, Yref=Yref, X=X, U=U, Xr=np.array(ctl.Xref), K=np.array(ctl.K),
             perts=perts, W=np.array(wind.w if hasattr(wind, 'w') else [0.5, -0.3]), tau_phi=ac.tau_phi, tau_v=ac.tau_v)


def fx_guess_poly():
    out = {}
    out['timing_in'] = np.array([[0, 10, 10], [0, 4.9, 10], [0, 7, 10], [0, 3, 10], [0, 12, 10], [0, 10, 50], [0.5, 6.0, 10]], float)
    out['timing_out'] = np.array([quiet(d2ou.planner_timing, *r) for r in out['timing_in']])
    tri_in, tri_out = [], []
    for (p0, p1, va, dur, N, gl) in [((0, 0), (0, 30), 12., 10., 101, 1.), ((0, 0), (0, 30), 12., 10., 101, -1.),
                                     ((0, 40), (75, 40), 12., 7.0, 71, -1.), ((3, -2), (50, 30), 12., 2.0, 21, 1.),
                                     ((-49.98, -58.14), (75, 40), 12., 12.0, 121, -1.), ((10, 5), (40, 45), 12., 4.9, 50, -1.)]:
        r = d2ou.triangle(np.array(p0, float), np.array(p1, float), va, dur, N, gl)
        tri_in.append([*p0, *p1, va, dur, N, gl]); tri_out.append(np.array(r))
    out['tri_in'] = np.array(tri_in)
    for i, r in enumerate(tri_out):
        out[f'tri_out_{i}'] = r
    # planner 'tri' guesses through the planner classes themselves
    import d2d.optyplan_scenarios as d2oscen
    sp = quiet(single_opt_planner.Planner, d2oscen.exp_14, True)
    out['single_exp14_tri'] = sp.get_initial_guess('tri')
    out['single_exp14_line'] = sp.get_initial_guess('line')
    scen = multi_opt_planner.trap_4
    scen.t1 = 7.0
    scen.p0s = ((0, 40, 0, 0, 12), (25, 20, 0, 0, 12), (25, -20, 0, 0, 12), (0, -40, 0, 0, 12))
    scen.p1s = ((75, 40, 0, 0, 12), (100, 20, 0, 0, 12), (100, -20, 0, 0, 12), (75, -40, 0, 0, 12))
    mp = quiet(multi_opt_planner.Planner, scen, True)
    out['multi_trap4_tri'] = mp.get_initial_guess('tri')
    out['multi_trap4_p0s'] = np.array(scen.p0s, float); out['multi_trap4_p1s'] = np.array(scen.p1s, float)
    out['multi_trap4_num_nodes'] = mp.num_nodes
    # polynomials
    pol = ddt.PolynomialOne([0, .05, 0, 0], [1, .05, 0, 0], 10)
    out['poly_ka_coefs'] = pol.coefs; out['poly_ka_get33'] = pol.get(3.3)
    Y0 = rng.uniform(-3, 3, (6, 4)); Y1 = rng.uniform(-3, 3, (6, 4)); T = rng.uniform(0.5, 3.0, 6); tt = rng.uniform(0, 1, (6, 5)) * T[:, None]
    out['poly_Y0'], out['poly_Y1'], out['poly_T'], out['poly_t'] = Y0, Y1, T, tt
    out['poly_coefs'] = np.array([ddt.PolynomialOne(Y0[i], Y1[i], T[i]).coefs for i in range(6)])
    out['poly_get'] = np.array([[ddt.PolynomialOne(Y0[i], Y1[i], T[i]).get(t) for t in tt[i]] for i in range(6)])
    np.savez(os.path.join(OUT, 'guess_poly.npz'), **out)


def fx_fit_cost():
    """Cost of polynomial trajectories computed by the REFERENCE's classes only:
    CompositeTraj/MinSnapPoly -> DiffFlatness -> CostInput + kobs*CostObstacles(kind 1)."""
    K, S = 50, 6
    _, _, dur = quiet(d2ou.planner_timing, 0.0, 4.9, 10.0)
    T = dur / S
    B = 12
    sc = ofit.set_scale(ofit.synth_scenarios(B, seed=7), 0.1, K)
    sc[0, ofit.SC_WX], sc[0, ofit.SC_WY] = 1.5, -0.7           # one case with wind
    ac = ddyn.Aircraft()
    out = dict(scen=sc, K=K, S=S, duration=dur)
    costs, frees, zs, wps = [], [], [], []
    for i in range(B):
        # random C^3 piecewise polynomial: junction data (pos + 3 derivatives per axis)
        J = np.zeros((S + 1, 2, 4))
        p0 = np.array([sc[i, ofit.SC_X0], sc[i, ofit.SC_Y0]]); p1 = np.array([sc[i, ofit.SC_X1], sc[i, ofit.SC_Y1]])
        for j in range(S + 1):
            J[j, :, 0] = p0 + (p1 - p0) * j / S + rng.normal(0, 2.0, 2)
            J[j, :, 1] = (p1 - p0) / dur + rng.normal(0, 1.5, 2)
            J[j, :, 2] = rng.normal(0, 2.0, 2); J[j, :, 3] = rng.normal(0, 2.0, 2)
        # the end knots meet the scenario's end conditions exactly (position; velocity = vref along psi0 / psi1,
        # src/single_opt_planner.py:46-49), so that the set lies in the fit's affine space z = Zp d + Z q and the GPU
        # can be evaluated AT this trajectory (tests/test_gpu_fit.py test_cost_vs_reference_classes_golden)
        vr = sc[i, ofit.SC_VREF]
        J[0, :, 0] = p0; J[0, :, 1] = vr * np.array([np.cos(sc[i, ofit.SC_PSI0]), np.sin(sc[i, ofit.SC_PSI0])])
        J[S, :, 0] = p1; J[S, :, 1] = vr * np.array([np.cos(sc[i, ofit.SC_PSI1]), np.sin(sc[i, ofit.SC_PSI1])])
        steps = [ddt.MinSnapPoly(J[j], J[j + 1], T) for j in range(S)]
        traj = ddt.CompositeTraj(steps)
        z = np.array([[st._polys[a].coefs[0] for st in steps] for a in range(2)])     # (2,S,8)
        W = [sc[i, ofit.SC_WX], sc[i, ofit.SC_WY]]
        t = np.linspace(0, dur, K)
        free = np.zeros(5 * K)
        for k in range(K):
            tk = min(t[k], dur * (1 - 1e-15)) if k == K - 1 else t[k]
            Ys = traj.get(tk)
            X, U, Xd = ddg.DiffFlatness.state_and_input_from_output(Ys, W, ac)
            free[0 * K + k], free[1 * K + k], free[2 * K + k], free[3 * K + k], free[4 * K + k] = X
        p = _FakeSingle(K, 0.1)
        obss = [(sc[i, ofit.SC_O0X], sc[i, ofit.SC_O0Y], sc[i, ofit.SC_O0R]), (sc[i, ofit.SC_O1X], sc[i, ofit.SC_O1Y], sc[i, ofit.SC_O1R])]
        c_in = d2ou.CostInput(sc[i, ofit.SC_VSP], sc[i, ofit.SC_KV], sc[i, ofit.SC_KPHI]).cost(free, p)
        c_ob = sc[i, ofit.SC_KOBS] * d2ou.CostObstacles(obss, 1).cost(free, p)
        wx, wy, _, _, _ = d2ou.triangle(p0, p1, sc[i, ofit.SC_VREF], dur, K, sc[i, ofit.SC_GOLEFT])
        costs.append([c_in, c_ob]); frees.append(free); zs.append(z); wps.append(np.stack([wx, wy]))
    out.update(cost_input_obst=np.array(costs), free=np.array(frees), z=np.array(zs), wp=np.array(wps))
    np.savez(os.path.join(OUT, 'fit_cost_golden.npz'), **out)


def fx_planner_goldens():
    """Committed solver outputs of the reference (values only) + the reference's cost on them."""
    out = {}
    d = np.load(os.path.join(REF, 'cache', 'optyplan_exp 14 - joining 2 points.npz'))
    N = len(d['sol_time'])
    free = np.concatenate([d['sol_x'], d['sol_y'], d['sol_psi'], d['sol_phi'], d['sol_v']])
    out['exp14_free'] = free; out['exp14_time'] = d['sol_time']
    out['exp14_cost_airvel12'] = d2ou.CostAirVel(12.0).cost(free, _FakeSingle(N, 1.0))
    df = pd.read_csv(os.path.join(REF, 'opt_states_st_line.csv'))
    n, N = 4, len(df)
    pm = _FakeMulti(N, n, 1.0)
    fm = np.zeros(5 * n * N)
    for i in range(n):
        fm[pm._slice_x[i]] = df[f'x_{i + 1}']; fm[pm._slice_y[i]] = df[f'y_{i + 1}']; fm[pm._slice_psi[i]] = df[f'psi_{i + 1}']
        fm[pm._slice_phi[i]] = df[f'phi_{i + 1}']; fm[pm._slice_v[i]] = df[f'v_{i + 1}']
    cc = d2mou.CostComposit(kvel=70., kbank=1., kobs=float('NaN'), kcol=10., vsp=12., obss=[], obs_kind=0, rcol=10.)
    out['stline_free'] = fm; out['stline_time'] = np.array(df['time'])
    out['stline_cost'] = cc.cost(fm, pm); out['stline_grad_norm'] = np.linalg.norm(cc.cost_grad(fm, pm))
    np.savez(os.path.join(OUT, 'planner_goldens.npz'), **out)
    # the other committed solver outputs: values only, with the reference's cost on them -- feasibility fixtures
    # (every one satisfies the backward-Euler collocation of the symbolic model to <= 4e-6) and known-answer costs
    feas = {}
    for tag, fn in (('exp0', 'optyplan_exp0.npz'), ('exp0_1_0', 'optyplan_exp0_1_0.npz'), ('exp0_1_1', 'optyplan_exp0_1_1.npz'),
                    ('exp0_1_2', 'optyplan_exp0_1_2.npz'), ('exp0_1_3', 'optyplan_exp0_1_3.npz'), ('exp0_1_4', 'optyplan_exp0_1_4.npz'),
                    ('exp13', 'optyplan_exp13 - some traj.npz')):
        d = np.load(os.path.join(REF, 'cache', fn))
        N = len(d['sol_time'])
        free = np.concatenate([d['sol_x'], d['sol_y'], d['sol_psi'], d['sol_phi'], d['sol_v']])
        feas[tag + '_W'] = np.stack([d['sol_x'], d['sol_y'], d['sol_psi'], d['sol_phi'], d['sol_v']])     # (5, N)
        feas[tag + '_time'] = d['sol_time']; feas[tag + '_wind'] = d['wind']
        feas[tag + '_cost_airvel12'] = d2ou.CostAirVel(12.0).cost(free, _FakeSingle(N, 1.0))
    for tag, fn in (('st_line', 'opt_states_st_line.csv'), ('simple_traj', 'opt_states_simple_traj.csv'), ('inf_traj_10s', 'inf_traj_10s.csv'),
                    ('opt_states', 'opt_states.csv'), ('opt_states_hf', 'opt_states_hf.csv')):
        df = pd.read_csv(os.path.join(REF, fn))
        n, N = 4, len(df)
        pm = _FakeMulti(N, n, 1.0)
        fm = np.zeros(5 * n * N)
        W = np.zeros((n, 5, N))
        for i in range(n):
            for c, nm in enumerate(('x', 'y', 'psi', 'phi', 'v')):
                W[i, c] = df[f'{nm}_{i + 1}']
            fm[pm._slice_x[i]] = W[i, 0]; fm[pm._slice_y[i]] = W[i, 1]; fm[pm._slice_psi[i]] = W[i, 2]
            fm[pm._slice_phi[i]] = W[i, 3]; fm[pm._slice_v[i]] = W[i, 4]
        feas[tag + '_W'] = W; feas[tag + '_time'] = np.array(df['time'])
        feas[tag + '_cost'] = cc.cost(fm, pm); feas[tag + '_grad_norm'] = np.linalg.norm(cc.cost_grad(fm, pm))
    np.savez_compressed(os.path.join(OUT, 'planner_feasibility_goldens.npz'), **feas)


def fx_tracking_trace():
    """100 steps of the phase-2/3 tracking loop body (src/11_full_sim_case1.py:272-290)
    on inf_traj_10s.csv, executed with the reference's own classes (CARE stand-in)."""
    df = pd.read_csv(os.path.join(REF, 'inf_traj_10s.csv'))
    n_ac = 4
    time = np.array(df['time'])
    x_ref = np.stack([df[f'x_{i + 1}'] for i in range(n_ac)], 1); y_ref = np.stack([df[f'y_{i + 1}'] for i in range(n_ac)], 1)
    dt = time[1] - time[0]
    T = len(time)
    F = []
    for j in range(n_ac):
        Fdx = np.gradient(x_ref[:, j], edge_order=2) / dt; Fddx = np.gradient(Fdx, edge_order=2) / dt
        Fdy = np.gradient(y_ref[:, j], edge_order=2) / dt; Fddy = np.gradient(Fdy, edge_order=2) / dt
        F.append((Fdx, Fdy, Fddx, Fddy))
    w = [0, 0]
    wind = ddg.WindField(w)
    ctrl = tracking.DiffController(w)
    acs = [ddyn.Aircraft() for _ in range(n_ac)]
    X = np.zeros((T, n_ac, 5)); U = np.zeros((T, n_ac, 2)); Xr = np.zeros((T, n_ac, 5)); dX = np.zeros((T, n_ac, 5)); K = np.zeros((T, n_ac, 2, 5))
    X[0] = np.array([[x_ref[0, j] + 1.0, y_ref[0, j] - 0.5, 0.05, 0.0, 12.0] for j in range(n_ac)])
    for i in range(1, T):
        for j in range(n_ac):
            Fdx, Fdy, Fddx, Fddy = F[j]
            a, b, c = ctrl.ComputeGain(time[i - 1], X[i - 1, j], [x_ref[i, j], y_ref[i, j]], [Fdx[i], Fdy[i]], [Fddx[i], Fddy[i]], [0, 0], acs[j])
            X[i, j] = acs[j].disc_dyn(X[i - 1, j], c, wind, time[i - 1], dt)
            U[i - 1, j] = c; dX[i - 1, j] = b; Xr[i - 1, j] = a; K[i - 1, j] = ctrl.K[-1]
    np.savez(os.path.join(OUT, 'tracking_trace_carestandin.npz'), time=time, x_ref=x_ref, y_ref=y_ref, X=X, U=U, Xr=Xr, dX=dX, K=K,
             tau_phi=acs[0].tau_phi, tau_v=acs[0].tau_v)


def fx_dfff():
    """DFFFController.get (src/d2d/guidance.py:62-91; 3-state LQR through the CARE stand-in) on seeded
    reference samples and perturbed states, with and without wind."""
    r = np.random.default_rng(7)
    n = 48
    Y = r.uniform(-80, 80, (n, 2)); psi = r.uniform(-np.pi, np.pi, n); sp = r.uniform(9, 15, n)
    Yd = np.stack([sp * np.cos(psi), sp * np.sin(psi)], 1)
    Ydd = r.uniform(-4, 4, (n, 2)); Yddd = r.uniform(-1, 1, (n, 2))
    W = np.stack([r.uniform(-2, 2, n), r.uniform(-2, 2, n)], 1); W[: n // 2] = 0.0
    ac = ddyn.Aircraft()

    class _Traj:                                   # the controller only needs .duration and .get(t)
        duration = 10.0

        def __init__(self, Ys): self.Ys = Ys
        def get(self, t): return self.Ys

    X = np.zeros((n, 5)); U = np.zeros((n, 2)); K = np.zeros((n, 2, 5)); Xr = np.zeros((n, 5))
    for i in range(n):
        Ys = np.array([Y[i], Yd[i], Ydd[i], Yddd[i]])
        ctl = ddg.DFFFController(_Traj(Ys), ac, ddg.WindField(list(W[i])))
        xr = ddg.DiffFlatness.state_and_input_from_output(Ys, W[i], ac)[0]
        X[i] = xr + np.array([r.uniform(-25, 25), r.uniform(-25, 25), r.uniform(-1.5, 1.5), r.uniform(-1.0, 1.0), r.uniform(-2, 2)])
        if i == 0:
            X[i, 2] = xr[2] + 2 * np.pi - 0.2     # wrap of dpsi
        U[i] = ctl.get(X[i].copy(), 0.5)
        K[i] = ctl.K[-1]; Xr[i] = ctl.Xref[-1]
    np.savez(os.path.join(OUT, 'dfff_carestandin.npz'), Y=Y, Yd=Yd, Ydd=Ydd, Yddd=Yddd, W=W, X=X, U=U, K=K, Xr=Xr,
             tau_phi=ac.tau_phi, tau_v=ac.tau_v)


def fx_traj_scen():
    """Demo trajectories (src/d2d/trajectory_factory.py) and simulation scenarios (src/d2d/scenario.py) of the reference: flat outputs
    traj.get(t) at seeded times, the scenarios' start states / time grids / winds / perturbations, the planner scenario catalogues'
    numeric attributes, and one run of the reference's run_simulation loop on scenario 'line2' (CARE stand-in)."""
    import importlib
    import d2d.trajectory_factory as ddtf
    import d2d.scenario as dds
    import d2d.optyplan_scenarios as d2oscen
    out = {}
    r = np.random.default_rng(11)
    for name in ('circle', 'two_lines', 'square', 'line_with_intro', 'demo_minsnap', 'slalom', 'sidemo'):
        traj, _ = quiet(ddtf.get, name)
        ts = np.sort(r.uniform(0, 2.2 * traj.duration, 40))
        out[f'traj_{name}_t'] = ts
        out[f'traj_{name}_Y'] = np.array([traj.get(t) for t in ts])
        out[f'traj_{name}_duration'] = traj.duration
    for name in ('line', 'line2', 'square', 'mucir', 'mucir2', 'patrol', 'patrol_2', 'patrol_3', 'circForm'):
        scen, _ = quiet(dds.get, name)
        out[f'scen_{name}_X0s'] = np.array(scen.X0s, dtype=float)
        out[f'scen_{name}_time'] = np.array([scen.time[0], scen.time[-1], len(scen.time)], dtype=float)
        out[f'scen_{name}_wind'] = np.array(scen.windfield.sample(0., [0., 0.]), dtype=float)
        out[f'scen_{name}_extends'] = np.array(scen.extends, dtype=float)
        ts = scen.time[::97][:30]
        out[f'scen_{name}_ts'] = ts
        out[f'scen_{name}_Y'] = np.array([[traj.get(t) for traj in scen.trajs] for t in ts])       # (nt, n, 4, 2)
        out[f'scen_{name}_pert_nz'] = np.array([[i, j, k, p[j, k]] for i, p in enumerate(scen.perts) for j, k in zip(*np.nonzero(p))],
                                                dtype=float).reshape(-1, 4)
    # planner catalogues: numeric attributes only (names and numbers, no code)
    sims = importlib.import_module('05_test_simulation')
    scen, _ = quiet(dds.get, 'line2')
    ctl = ddg.DFFFController(scen.trajs[0], scen.aircrafts[0], scen.windfield)
    X, U, Yref = sims.run_simulation(scen.time[:400], scen.aircrafts[0], scen.windfield, ctl, scen.X0s[0], scen.perts[0])
    out['run_line2_X'], out['run_line2_U'], out['run_line2_Yref'] = X, U, Yref
    for i, sc in enumerate(d2oscen.scens):
        out[f'plan_{i}_name'] = np.array(sc.name)
        out[f'plan_{i}_num'] = np.array([sc.t0, sc.t1, sc.hz, sc.obj_scale, sc.vref, sc.ncases, len(sc.obstacles)] + list(sc.p0) + list(sc.p1)
                                        + list(sc.phi_constraint) + list(sc.v_constraint), dtype=float)
    m07 = importlib.import_module('07_multioptyplan')
    for i, sc in enumerate(m07.scens):
        out[f'mplan_{i}_name'] = np.array(sc.name)
        out[f'mplan_{i}_num'] = np.array([sc.t0, sc.t1, sc.hz, sc.obj_scale, sc.vref, sc.ncases, len(sc.obstacles), len(sc.p0s)]
                                         + list(np.ravel(sc.p0s)) + list(np.ravel(sc.p1s)) + list(sc.phi_constraint) + list(sc.v_constraint),
                                         dtype=float)
    np.savez(os.path.join(OUT, 'traj_scen.npz'), **out)


if __name__ == '__main__':
    only = sys.argv[1:]
    for f in (fx_plant, fx_flatness_ctrl, fx_guidance, fx_states_over_time, fx_costs, fx_guess_poly,
              fx_fit_cost, fx_planner_goldens, fx_tracking_trace, fx_dfff, fx_costs3, fx_dfff_run, fx_traj_scen):
        if only and f.__name__ not in only:
            continue
        f(); print('wrote', f.__name__)
