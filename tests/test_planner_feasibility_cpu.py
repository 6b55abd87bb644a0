"""The two catalogue cases no backend solves with hard bounds -- optyplan_scenarios.exp_0_3 and exp_3 (src/d2d/optyplan_scenarios.py:
56-63, 96-103: a turn-around in 20 s inside a 50 m / 40 m box) -- are infeasible as posed: 20 s at v >= 9 m/s are >= 180 m of path
with curvature <= g tan(30 deg) / v^2 <= 0.0699 1/m, the end condition psi(t1) = pi fixes the net rotation to +pi (an extra full
loop would end at 3 pi), and no such curve fits the box.  tools/dev_arcpaths.py searches the constant-speed arc sequences (the
extremals of that problem: v = v_min is the shortest path AND the tightest turn) by multi-start least squares; its smallest violation
stays at 2.99 (exp_0_3) / 4.50 (exp_3) from hundreds of starts with 6 and 9 arcs, while the same search solves the
same scenario with a shorter duration to 1e-10.  The collocation backend's verdict for both is D2D_ST_STALLED (tests/test_gpu_nlp.py)."""
import contextlib
import io
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


@pytest.fixture(scope='module')
def arc():
    argv, sys.argv = sys.argv, ['dev_arcpaths']
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            import dev_arcpaths
            import d2d.optyplan_scenarios as sc
    finally:
        sys.argv = argv
    return dev_arcpaths, sc


def test_search_finds_a_path_when_one_exists(arc):
    D, sc = arc

    class shorter(sc.exp_0_3):          # 11 s = 99 m: a straight, the half circle of radius 15 m, a straight
        t1 = 11.0
    best = D.search(shorter, K=6, starts=6, seed=1, verbose=False)
    assert best[0][0] < 1e-6


@pytest.mark.parametrize('name,floor', [('exp_0_3', 2.9), ('exp_3', 4.4)])
def test_box_turnarounds_are_infeasible_as_posed(arc, name, floor):
    D, sc = arc
    best = D.search(getattr(sc, name), K=6, starts=6, seed=2, verbose=False)
    assert best and best[0][0] >= floor, best[0][0]      # every local minimum of the violation is at least the global one
