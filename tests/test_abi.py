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
    assert ctypes.sizeof(d2dhip.FitOpts) == 2 * 4 + 4 * 8 + 2 * 4 + 3 * 8 + 2 * 4     # + mode, mp_finish, mp_ftol/xtol/gtol, slice, mp_slow
    assert d2dhip.SCEN_STRIDE == 80 and d2dhip.MAX_OBS == 16
    # defaults that the binding and the oracle repeat from the header
    hdr = open(os.path.join(ROOT, 'include', 'd2d.h')).read()
    from oracle import nlp as NL, fit as F, sim as S
    val = lambda name: float(re.search(r'#define %s ([-+0-9.eE]+)' % name, hdr).group(1))       # noqa: E731
    assert (d2dhip.NLP_INNER_MAX, d2dhip.NLP_OUTER_MAX) == (val('D2D_NLP_INNER_MAX'), val('D2D_NLP_OUTER_MAX')) == (NL.INNER_MAX, NL.OUTER_MAX)
    assert (F.GS_LS_SWEEP0, F.GS_LS_RATIO, F.GS_LS_FIRST_MAX, F.GS_LS_MAX) == tuple(val('D2D_GS_LS_' + k) for k in ('SWEEP0', 'RATIO', 'FIRST_MAX', 'MAX'))
    assert (S.GL_FAST_STAGES, S.GL_FAST_DPHI, S.GL_FAST_RATIO) == (val('D2D_GL_FAST_STAGES'), val('D2D_GL_FAST_DPHI'), val('D2D_GL_FAST_RATIO'))


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
