"""CPU: the C-ABI library loads and exports every symbol include/d2d.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'd2d.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(d2d_[a-z_0-9]+)\s*\(', src)))


def test_header_symbols_exported():
    import d2dhip
    if not os.path.exists(d2dhip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(d2dhip.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/d2d.h but not exported'
    # the binding declares exactly the header's functions
    assert sorted(d2dhip.EXPORTS) == names
    lib.d2d_version.restype = ctypes.c_int
    hdr = open(os.path.join(ROOT, 'include', 'd2d.h')).read()
    assert lib.d2d_version() == int(re.search(r'#define D2D_VERSION (\d+)', hdr).group(1))      # the .so is the build of this header


def test_struct_layouts_match_header():
    import d2dhip
    assert ctypes.sizeof(d2dhip.GvfParams) == 4 * 4 + 9 * 8 + 2 * 4 + 3 * 8
    assert ctypes.sizeof(d2dhip.TrackParams) == 2 * 4 + 5 * 8 + 5 * 8 + 3 * 8 + 7 * 8
    # + mode, mp_finish, mp_ftol/xtol/gtol, slice, mp_slow; version 108: handout, prio_at, gs_ls, gs_ls_s0, gs_ls_r0, gs_prio_at, gs_pairs
    assert ctypes.sizeof(d2dhip.FitOpts) == 2 * 4 + 4 * 8 + 2 * 4 + 3 * 8 + 2 * 4 + 4 * 4 + 8 + 2 * 4
    assert ctypes.sizeof(d2dhip.FitPlanOpts) == 4 * 4
    assert ctypes.sizeof(d2dhip.NlpOpts) == 5 * 8 + 4 * 4 + 8 + 8
    assert d2dhip.SCEN_STRIDE == 80 and d2dhip.MAX_OBS == 16
    # defaults that the binding and the oracle repeat from the header
    hdr = open(os.path.join(ROOT, 'include', 'd2d.h')).read()
    from oracle import nlp as NL, fit as F, sim as S
    val = lambda name: float(re.search(r'#define %s ([-+0-9.eE]+)' % name, hdr).group(1))       # noqa: E731
    assert (d2dhip.NLP_INNER_MAX, d2dhip.NLP_OUTER_MAX) == (val('D2D_NLP_INNER_MAX'), val('D2D_NLP_OUTER_MAX')) == (NL.INNER_MAX, NL.OUTER_MAX)
    assert (F.GS_LS_SWEEP0, F.GS_LS_RATIO, F.GS_LS_FIRST_MAX, F.GS_LS_MAX) == tuple(val('D2D_GS_LS_' + k) for k in ('SWEEP0', 'RATIO', 'FIRST_MAX', 'MAX'))
    assert (S.GL_FAST_STAGES, S.GL_FAST_DPHI, S.GL_FAST_RATIO) == (val('D2D_GL_FAST_STAGES'), val('D2D_GL_FAST_DPHI'), val('D2D_GL_FAST_RATIO'))


def test_fit_opts_defaults_header_library_binding():
    """d2d_fit_opts_default (a host function: no GPU) = the binding's fit_opts() field by field = the header's D2D_* defaults."""
    import d2dhip
    lib = d2dhip.load()
    o = d2dhip.FitOpts()
    assert lib.d2d_fit_opts_default(ctypes.byref(o)) == 0
    b = d2dhip.fit_opts()
    for name, _ in d2dhip.FitOpts._fields_:
        assert getattr(o, name) == getattr(b, name), name
    hdr = open(os.path.join(ROOT, 'include', 'd2d.h')).read()
    val = lambda name: float(re.search(r'#define %s ([-+0-9.eE]+)' % name, hdr).group(1))       # noqa: E731
    assert (o.prio_at, o.gs_ls_s0, o.gs_ls_r0, o.gs_prio_at, o.mp_finish, o.mp_slow, o.slice) == tuple(
        val('D2D_' + k) for k in ('LM_PRIO_AT', 'GS_LS_SWEEP0', 'GS_LS_RATIO', 'GS_PRIO_AT', 'LM_MP_FINISH', 'LM_MP_SLOW', 'LM_SLICE'))
    assert o.handout == d2dhip.HANDOUT_PREDICTED == 1 and o.gs_ls == 1 and o.gs_pairs == 0 and o.mode == d2dhip.MODE_MINPACK
    assert (d2dhip.KERNEL_AUTO, d2dhip.KERNEL_SPLIT, d2dhip.KERNEL_FUSED, d2dhip.KERNEL_LONG, d2dhip.KERNEL_KNOT) == (-1, 0, 1, 2, 3)
    from d2dhip import handout
    assert (handout.NB, handout.ND, handout.X_LO, handout.X_HI) == tuple(val('D2D_HANDOUT_' + k) for k in ('NB', 'ND', 'X_LO', 'X_HI'))


def test_behaviour_is_not_selected_by_the_environment():
    """which kernel solves a plan, the group line search, slots ... are arguments of the ABI; getenv stays for
    stamps / debug dumps / ablation diagnostics only."""
    import glob
    names = set()
    for f in glob.glob(os.path.join(ROOT, 'drone-sim-python_amd', 'csrc', '*')):
        if f.endswith(('.hip', '.cpp', '.h')):
            names |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(f).read()))
    assert len(names) <= 8, sorted(names)
    assert all(any(t in n for t in ('STAMPS', 'DEBUG', 'DIAG', 'TIMES', 'CLOCK', 'ABLATE', 'GEOM')) for n in names), sorted(names)


def test_handout_prior_header_is_the_regression_of_the_committed_counts():
    """csrc/fit_handout_prior.h is generated (tools/fit_handout_prior.py) from tools/data/handout_calib.npz: regenerate the table
    and compare; the calibration seeds are disjoint from every bench / test batch (rank offsets 100 ..)."""
    import numpy as np
    from d2dhip import synth, handout
    cal = np.load(os.path.join(ROOT, 'tools', 'data', 'handout_calib.npz'))
    assert int(cal['ranks'].min()) >= 100
    B = int(cal['B'])
    dur = synth.planner_timing(0, 4.9, 10)[2]
    sc = np.concatenate([synth.synth_scenarios(B, seed=20241008, rank=int(r), obj_scale=0.1, K=50) for r in cal['ranks']])
    n = np.concatenate([cal[f'iters_{r}'] for r in cal['ranks']]).astype(np.float64)
    table = handout.fit_prior(sc, dur, n)
    txt = open(os.path.join(ROOT, 'drone-sim-python_amd', 'csrc', 'fit_handout_prior.h')).read()
    vals = np.array([float(v) for v in re.findall(r'(-?[0-9]+\.[0-9]+)f', txt)], dtype=np.float32).reshape(2, handout.NB, handout.ND)
    assert np.abs(vals - table).max() <= 0.0051
    # ... and it predicts: rank correlation with the counts it was regressed on
    from scipy.stats import spearmanr
    assert spearmanr(handout.key(sc[:32768], dur, vals)[0], n[:32768]).statistic >= 0.55


def test_missing_library_fails_loudly(monkeypatch):
    import d2dhip
    monkeypatch.setattr(d2dhip, '_lib', None)
    monkeypatch.setattr(d2dhip, 'LIB_PATH', '/nonexistent/libd2dhip.so')
    with pytest.raises(d2dhip.D2DError):
        d2dhip.load()


def test_host_code_under_asan_ubsan():
    """`make asan`: the CPU-side basis construction (csrc/fit_basis.cpp) built with -fsanitize=address,undefined and run over
    the plan shapes the planners use (K = 50 ... 501) and a rank-deficient one (sanitizers on the CPU build only)."""
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'drone-sim-python_amd', 'csrc')
    r = subprocess.run(['make', '-C', csrc, 'asan'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'asan_host: ok' in r.stdout and 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr
