"""GPU: the legacy simulation catalogue (src/d2d/scenario.py, src/d2d/trajectory_factory.py, src/05_test_simulation.py) and the
planner scenario catalogues (src/d2d/optyplan_scenarios.py, src/07_multioptyplan.py:170-435) through the device paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_trajectory_sampler_vs_reference_get(gold):
    """d2d_traj_sample (lines, circle arcs, slaloms, min-snap polynomials, composites) against the reference's traj.get(t) at the
    seeded times of the fixture -- beyond one period for the composites -- for all trajectories in ONE launch."""
    import d2dhip
    import d2d.trajectory as ddt
    import d2d.trajectory_factory as ddtf
    g = gold('traj_scen')
    ctx = d2dhip.default_context()
    names = ('circle', 'two_lines', 'square', 'line_with_intro', 'demo_minsnap', 'slalom')
    trajs = [ddtf.get(n)[0] for n in names]
    rows = [ddt.describe(t) for t in trajs]
    assert all(r is not None for r in rows) and ddt.describe(ddtf.get('sidemo')[0]) is None
    desc = ctx.dev(np.stack(rows))
    for j, n in enumerate(names):
        for t, Yg in zip(g[f'traj_{n}_t'][::3], g[f'traj_{n}_Y'][::3]):
            Y = ctx.traj_sample(desc, 1, float(t), 1.0).cpu().numpy()[0, :, j]        # x, y, xd, yd, xdd, ydd
            np.testing.assert_allclose(Y, Yg[:3].reshape(-1), rtol=1e-11, atol=1e-9, err_msg=f'{n} t={t}')
    # a grid: T samples of all of them, against the host classes
    T, dt = 200, 0.173
    Y = ctx.traj_sample(desc, T, 0.5, dt).cpu().numpy()
    for j, tr in enumerate(trajs):
        Yh = np.array([np.asarray(tr.get(0.5 + i * dt))[:3].reshape(-1) for i in range(T)])
        np.testing.assert_allclose(Y[:, :, j], Yh, rtol=1e-11, atol=1e-9, err_msg=names[j])


def test_scenarios_and_simulation_vs_reference(gold):
    """Every constructible scenario of the reference's registry: start states (incl. the ones computed through the flatness map),
    references, and scenario 'line2' flown by full_sim.test_simulation against the reference's own run_simulation (CARE stand-in)."""
    import d2d.scenario as dds
    import full_sim
    g = gold('traj_scen')
    for name in ('line', 'line2', 'square', 'mucir', 'mucir2', 'patrol', 'patrol_2', 'patrol_3', 'circForm'):
        scen, _ = dds.get(name)
        np.testing.assert_allclose(np.array(scen.X0s, dtype=float), g[f'scen_{name}_X0s'], atol=1e-10, err_msg=name)
        np.testing.assert_allclose([scen.time[0], scen.time[-1], len(scen.time)], g[f'scen_{name}_time'], atol=1e-9)
        Y = np.array([[traj.get(t) for traj in scen.trajs] for t in g[f'scen_{name}_ts']])
        np.testing.assert_allclose(Y, g[f'scen_{name}_Y'], rtol=1e-12, atol=1e-10, err_msg=name)
    scen, _ = dds.get('line2')
    scen.time = scen.time[:400]
    scen.perts = [p[:400] for p in scen.perts]
    Xs, Us, Yrefs = full_sim.test_simulation(scen)
    np.testing.assert_allclose(Yrefs[:, 0, :3], g['run_line2_Yref'][:, :3], rtol=1e-11, atol=1e-9)
    # per-step parity with the reference's loop: bounded by its LSODA integrator and accumulated over 400 steps
    assert np.abs(Xs[0] - g['run_line2_X']).max() < 2e-4 and np.abs(Us[0] - g['run_line2_U']).max() < 2e-4
    # all aircraft of a multi-trajectory scenario in ONE device loop (references sampled on the device): every aircraft against the
    # oracle's restatement of the reference's loop driven with the host classes' traj.get(t)
    from oracle import sim as S
    for name in ('patrol_2', 'mucir', 'patrol_3'):
        scen, _ = dds.get(name)
        scen.time = scen.time[:300]
        scen.perts = [p[:300] for p in scen.perts]
        Xs, Us, Yrefs = full_sim.test_simulation(scen)
        assert len(Xs) == len(scen.trajs)
        w = scen.windfield.sample(0., None)
        for j, tr in enumerate(scen.trajs):
            Ys = np.array([tr.get(t) for t in scen.time])
            np.testing.assert_allclose(Yrefs[:, j], Ys[:, :3], rtol=1e-11, atol=1e-9)
            Xo, Uo, _ = S.dfff_run(scen.time, Ys, scen.X0s[j], scen.perts[j], W=w)
            assert np.abs(Xs[j] - Xo).max() < 1e-6 and np.abs(Us[j] - Uo).max() < 1e-6, (name, j, np.abs(Xs[j] - Xo).max())


def test_every_planner_scenario_of_both_catalogues_plans():
    """All 15 single-aircraft scenarios (every case) through single_opt_planner and all 15 multi-aircraft scenarios through
    multi_opt_planner: a finite plan that starts and ends where the scenario says, on either backend for a sample of them."""
    import importlib
    import d2d.optyplan_scenarios as d2oscen
    import d2d.multioptyplan_scenarios as d2mscen
    import single_opt_planner as sop
    import multi_opt_planner as mop
    importlib.reload(d2oscen); importlib.reload(d2mscen)
    n_run = 0
    for sc in d2oscen.scens:
        for c in range(sc.ncases):
            sc.set_case(c)
            p = sop.Planner(sc, initialize=True)
            p.configure(sc.tol, 300)
            p.run(p.get_initial_guess('tri'))
            assert np.isfinite(p.solution).all(), (sc.name, c)
            np.testing.assert_allclose([p.sol_x[0], p.sol_y[0], p.sol_x[-1], p.sol_y[-1]], [sc.p0[0], sc.p0[1], sc.p1[0], sc.p1[1]], atol=1e-7)
            n_run += 1
    assert n_run >= 30
    importlib.reload(d2oscen)
    for sc in d2mscen.scens:
        for c in range(sc.ncases):
            sc.set_case(c)
            # (coupled groups at 50 Hz -- exp_2: 276 nodes, exp_5: 401 -- run their visits on the chunked persistent kernel)
            p = mop.Planner(sc, initialize=True, backend='fit')
            p.run(initial_guess=p.get_initial_guess('tri'), tol=sc.tol, max_iter=300)
            p.interpret_solution()
            assert np.isfinite(p.solution).all(), (sc.name, c)
            for i in range(len(sc.p0s)):
                np.testing.assert_allclose([p.sol_x[i][0], p.sol_y[i][0], p.sol_x[i][-1], p.sol_y[i][-1]],
                                           [sc.p0s[i][0], sc.p0s[i][1], sc.p1s[i][0], sc.p1s[i][1]], atol=1e-7)
    importlib.reload(d2mscen)
    # the collocation backend on a sample: hard bounds and feasibility
    for sc in (d2oscen.exp_1, d2oscen.exp_4_1, d2oscen.exp_13):
        p = sop.Planner(sc, initialize=True, backend='nlp')
        p.run()
        assert np.abs(p.sol_phi).max() <= sc.phi_constraint[1] + 1e-12 and p.sol_v.min() >= sc.v_constraint[0] - 1e-12
        if sc is not d2oscen.exp_13:                     # (exp_13 is infeasible: the reference's own IPOPT run did not converge)
            assert p.info['status'] == 1 and p.info['feas'] <= 1e-8, (sc.name, p.info)
