"""ctypes binding of libd2dhip.so (include/d2d.h) -- the only way host code reaches the
HIP kernels.  PyTorch-ROCm tensors are used purely as device buffers (data_ptr()) and
for the stream; no torch op takes part in the numerics.

There is no CPU fallback: importing this module without the built library raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('D2D_LIB') or os.path.join(os.path.dirname(_HERE), 'lib', 'libd2dhip.so')   # D2D_LIB: A/B builds

SCEN_STRIDE = 80
MAX_OBS = 16
(SC_X0, SC_Y0, SC_PSI0, SC_X1, SC_Y1, SC_PSI1, SC_VREF, SC_VSP, SC_KV, SC_KPHI, SC_KOBS, SC_S,
 SC_WWP, SC_WX, SC_WY, SC_GOLEFT, SC_O0X, SC_O0Y, SC_O0R, SC_O1X, SC_O1Y, SC_O1R, SC_WBND,
 SC_PHIMAX, SC_VMIN, SC_VMAX, SC_KCOL, SC_RCOL, SC_SCOL, SC_PMASK, SC_OKIND, SC_BANKMAX, SC_OEXT) = range(33)
SC_XMIN, SC_XMAX, SC_YMIN, SC_YMAX = range(SC_OEXT + 3 * (MAX_OBS - 2), SC_OEXT + 3 * (MAX_OBS - 2) + 4)   # soft position box


def obs_col(i):
    """First of the three columns (x, y, r) of static obstacle i in a scenario row (include/d2d.h D2D_SC_OEXT)."""
    return SC_O0X + 3 * i if i < 2 else SC_OEXT + 3 * (i - 2)


TRAJ_MAX_SEG, TRAJ_SEG_STRIDE = 8, 20
TRAJ_STRIDE = 4 + TRAJ_MAX_SEG * TRAJ_SEG_STRIDE
TRAJ_LINE, TRAJ_CIRCLE, TRAJ_SLALOM, TRAJ_POLY = 1, 2, 3, 4
ST_RUNNING, ST_CONVERGED, ST_MAXITER, ST_NONFINITE, ST_STALLED = range(5)


class D2DError(RuntimeError):
    pass


class GvfParams(C.Structure):
    _fields_ = [('n_form', C.c_int32), ('n_ac', C.c_int32), ('n_rows', C.c_int32), ('rec_stride', C.c_int32),
                ('dt', C.c_double), ('tau_phi', C.c_double), ('tau_v', C.c_double),
                ('ke', C.c_double), ('kd', C.c_double), ('kr', C.c_double), ('v_c', C.c_double),
                ('wx', C.c_double), ('wy', C.c_double), ('use_stop', C.c_int32), ('stop_hold', C.c_int32),
                ('stop_tol', C.c_double * 3)]


class TrackParams(C.Structure):
    _fields_ = [('n', C.c_int32), ('n_rows', C.c_int32), ('dt', C.c_double), ('tau_phi', C.c_double),
                ('tau_v', C.c_double), ('wx', C.c_double), ('wy', C.c_double),
                ('err_sats', C.c_double * 5), ('v_min', C.c_double), ('v_max', C.c_double),
                ('phi_lim', C.c_double), ('q_diag', C.c_double * 5), ('r_diag', C.c_double * 2)]


class NlpOpts(C.Structure):
    _fields_ = [('rho0', C.c_double), ('mub0', C.c_double), ('mub_min', C.c_double), ('feas_tol', C.c_double),
                ('opt_tol', C.c_double), ('inner_max', C.c_int32), ('outer_max', C.c_int32), ('serial', C.c_int32), ('slots', C.c_int32),
                ('bounds', C.c_void_p), ('order', C.c_void_p)]


class FitOpts(C.Structure):
    _fields_ = [('max_iter', C.c_int32), ('check_every', C.c_int32), ('ftol', C.c_double),
                ('gtol', C.c_double), ('xtol', C.c_double), ('so_lambda', C.c_double),
                ('mode', C.c_int32), ('mp_finish', C.c_int32), ('mp_ftol', C.c_double), ('mp_xtol', C.c_double),
                ('mp_gtol', C.c_double), ('slice', C.c_int32), ('mp_slow', C.c_int32),
                ('handout', C.c_int32), ('prio_at', C.c_int32), ('gs_ls', C.c_int32), ('gs_ls_s0', C.c_int32), ('gs_ls_r0', C.c_double),
                ('gs_prio_at', C.c_int32), ('gs_pairs', C.c_int32)]


class FitPlanOpts(C.Structure):
    _fields_ = [('kernel', C.c_int32), ('long_tables', C.c_int32), ('reserved', C.c_int32 * 2)]


SO_LAMBDA = 1e-4          # D2D_LM_SO_LAMBDA: damping below which the evaluations carry the second-order term (FAST mode)
NLP_INNER_MAX, NLP_OUTER_MAX = 20, 120     # include/d2d.h D2D_NLP_INNER_MAX / D2D_NLP_OUTER_MAX (tests/test_abi.py compares)
MODE_MINPACK, MODE_FAST = 0, 1     # D2D_LM_MODE_*: MINPACK's lmder path (what scipy least_squares('lm') follows) / rounds 1-2's loop
MP_FINISH = 3             # D2D_LM_MP_FINISH: calm lmder steps before the second-order finish (0 = pure lmder)
MP_SLOW = 8               # D2D_LM_MP_SLOW: stagnating lmder trials (cost change <= 1e-4 of itself) in a row before the finish (0 = never)
SLICE = 0                 # D2D_LM_SLICE: iterations a fit runs before it yields its wavefront to waiting fits (0 = never)
HANDOUT_INDEX, HANDOUT_PREDICTED = 0, 1   # D2D_HANDOUT_*: hand-out order of a batch larger than the resident wavefronts
PRIO_AT = 48              # D2D_LM_PRIO_AT
GS_LS_SWEEP0, GS_LS_RATIO, GS_PRIO_AT = 8, 0.8, 40     # D2D_GS_LS_SWEEP0 / D2D_GS_LS_RATIO / D2D_GS_PRIO_AT
KERNEL_AUTO, KERNEL_SPLIT, KERNEL_FUSED, KERNEL_LONG, KERNEL_KNOT = -1, 0, 1, 2, 3     # D2D_FIT_KERNEL_*
KERNELS = {'auto': KERNEL_AUTO, 'split': KERNEL_SPLIT, 'fused': KERNEL_FUSED, 'long': KERNEL_LONG, 'knot': KERNEL_KNOT}


def fit_opts(max_iter=200, check_every=8, ftol=1e-14, gtol=1e-9, xtol=1e-11, so_lambda=SO_LAMBDA, mode=MODE_MINPACK,
             mp_finish=MP_FINISH, mp_tol=1e-15, slice=SLICE, mp_slow=MP_SLOW, handout=HANDOUT_PREDICTED, prio_at=PRIO_AT,
             gs_ls=1, gs_ls_s0=GS_LS_SWEEP0, gs_ls_r0=GS_LS_RATIO, gs_prio_at=GS_PRIO_AT, gs_pairs=0):
    """d2d_fit_opts with the library's defaults (include/d2d.h; tests/test_abi.py compares them with d2d_fit_opts_default)."""
    return FitOpts(max_iter, check_every, ftol, gtol, xtol, so_lambda, mode, mp_finish, mp_tol, mp_tol, mp_tol, slice, mp_slow,
                   handout, prio_at, gs_ls, gs_ls_s0, gs_ls_r0, gs_prio_at, gs_pairs)


_P = C.c_void_p
_SIGS = {
    'd2d_version': (C.c_int, []),
    'd2d_last_error': (C.c_char_p, []),
    'd2d_ctx_create': (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    'd2d_ctx_destroy': (C.c_int, [_P]),
    'd2d_ctx_sync': (C.c_int, [_P]),
    'd2d_comm_available': (C.c_int, []),
    'd2d_comm_unique_id': (C.c_int, [_P]),
    'd2d_comm_create': (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    'd2d_comm_destroy': (C.c_int, [_P]),
    'd2d_comm_info': (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'd2d_allreduce_stats': (C.c_int, [_P, _P, _P]),
    'd2d_step': (C.c_int, [_P, C.c_int, _P, _P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    'd2d_sim_gvf_run': (C.c_int, [_P, C.POINTER(GvfParams)] + [_P] * 13),
    'd2d_ctrl_gain': (C.c_int, [_P, C.POINTER(TrackParams)] + [_P] * 6),
    'd2d_dfff_eval': (C.c_int, [_P, C.POINTER(TrackParams)] + [_P] * 5),
    'd2d_sim_dfff_run': (C.c_int, [_P, C.POINTER(TrackParams)] + [_P] * 7),
    'd2d_sim_track_run': (C.c_int, [_P, C.POINTER(TrackParams)] + [_P] * 10),
    'd2d_traj_sample': (C.c_int, [_P, C.c_int, C.c_int, C.c_double, C.c_double, _P, _P]),
    'd2d_dcf_eval': (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_double, _P, _P, _P, _P]),
    'd2d_gvf_eval': (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_double, C.c_double, _P]),
    'd2d_flatness': (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_double, C.c_double, C.c_double, C.c_double, _P, _P, _P]),
    'd2d_cont_jac': (C.c_int, [_P, C.c_int, _P, C.c_double, C.c_double, _P, _P]),
    'd2d_lqr': (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P]),
    'd2d_nlp_workspace_doubles': (C.c_int, [C.c_int]),
    'd2d_nlp_solve': (C.c_int, [_P, C.c_int, C.c_int, C.c_double, _P, C.POINTER(NlpOpts)] + [_P] * 8),
    'd2d_nlp_solve_groups': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_double, _P, C.POINTER(NlpOpts), C.c_int, C.c_double] + [_P] * 9),
    'd2d_fit_plan_create': (C.c_int, [_P, C.c_int, C.c_int, C.c_double, _P, C.POINTER(_P)]),
    'd2d_fit_plan_create_ex': (C.c_int, [_P, C.c_int, C.c_int, C.c_double, _P, C.POINTER(FitPlanOpts), C.POINTER(_P)]),
    'd2d_fit_opts_default': (C.c_int, [C.POINTER(FitOpts)]),
    'd2d_fit_plan_destroy': (C.c_int, [_P]),
    'd2d_fit_plan_get': (C.c_int, [_P] * 6),
    'd2d_fit_plan_kernel': (C.c_int, [_P]),
    'd2d_fit_init': (C.c_int, [_P, _P, C.c_int, _P, _P]),
    'd2d_fit_project': (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    'd2d_fit_eval': (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P]),
    'd2d_fit_rows': (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P]),
    'd2d_fit_jtj': (C.c_int, [_P, _P, C.c_int, _P]),
    'd2d_fit_solve': (C.c_int, [_P, _P, C.c_int, _P, _P, C.POINTER(FitOpts), _P, _P, _P, _P]),
    'd2d_fit_begin': (C.c_int, [_P, _P, C.c_int]),
    'd2d_fit_iterate': (C.c_int, [_P, _P, C.c_int, _P, _P, C.POINTER(FitOpts), C.c_int, C.POINTER(C.c_int32)]),
    'd2d_fit_finish': (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P, _P]),
    'd2d_fit_plan_set_order': (C.c_int, [_P, _P, C.c_int, _P]),
    'd2d_fit_plan_set_handout_prior': (C.c_int, [_P, _P, _P]),
    'd2d_fit_plan_get_order': (C.c_int, [_P, _P, C.c_int, _P]),
    'd2d_fit_plan_set_group_order': (C.c_int, [_P, _P, C.c_int, C.c_int]),
    'd2d_fit_group_report': (C.c_int, [_P, _P, C.c_int, _P, _P]),
    'd2d_fit_plan_set_groups': (C.c_int, [_P, C.c_int]),
    'd2d_fit_solve_groups': (C.c_int, [_P, _P, C.c_int, _P, _P, C.POINTER(FitOpts), C.c_int, C.c_int, C.c_double, _P,
                                       C.POINTER(C.c_int32), _P]),
    'd2d_fit_profile': (C.c_int, [_P, C.c_int]),
    'd2d_fit_profile_read': (C.c_int, [_P, _P]),
    'd2d_fit_coeffs': (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    'd2d_fit_sample': (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """dlopen libd2dhip.so and declare every signature; fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise D2DError(f'{LIB_PATH} not found: build it with __graft_entry__.build() '
                       f'(make -C drone-sim-python_amd/csrc); there is no CPU fallback')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise D2DError(f'libd2dhip error {rc}: {load().d2d_last_error().decode()}')


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise D2DError('no HIP device visible to PyTorch-ROCm; the d2d engine has no CPU fallback')
    return torch


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _hptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Comm:
    """d2d_comm: the RCCL communicator of the sharded solve's convergence exchange (d2d_allreduce_stats)."""

    def __init__(self, ctx, uid, rank, world):
        assert len(uid) == 128
        self.ctx = ctx
        h = _P()
        _check(ctx.lib.d2d_comm_create(ctx.h, C.c_char_p(uid), int(rank), int(world), C.byref(h)))
        self.h = h

    def info(self):
        r, w = C.c_int32(-1), C.c_int32(-1)
        _check(self.ctx.lib.d2d_comm_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def allreduce_stats(self, stats_dev):
        """stats_dev: device tensor of three doubles, in place: [0] sum, [1] max, [2] sum over the ranks; enqueued on the context's stream."""
        assert stats_dev.is_cuda and stats_dev.dtype == _torch().float64 and stats_dev.numel() == 3 and stats_dev.is_contiguous()
        _check(self.ctx.lib.d2d_allreduce_stats(self.ctx.h, self.h, _ptr(stats_dev)))

    def close(self):
        if getattr(self, 'h', None):
            self.ctx.lib.d2d_comm_destroy(self.h)
            self.h = None

    __del__ = close


class Context:
    """d2d_ctx bound to the current torch stream of `device`."""

    def __init__(self, device=0):
        torch = _torch()
        self.lib = load()
        self.device = torch.device('cuda', device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.current_stream(self.device)
        h = _P()
        _check(self.lib.d2d_ctx_create(device, C.c_void_p(self.stream.cuda_stream), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, 'h', None):
            self.lib.d2d_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def sync(self):
        _check(self.lib.d2d_ctx_sync(self.h))

    # -- multi-GPU convergence exchange through the C-ABI (RCCL, include/d2d.h d2d_comm_*) ------------
    def comm_available(self):
        """local, no communication: can this process load RCCL (d2d_comm_available)?  None when it can, else the reason."""
        rc = self.lib.d2d_comm_available()
        return None if rc == 0 else self.lib.d2d_last_error().decode()

    def comm_unique_id(self):
        """128 opaque bytes (ncclGetUniqueId): rank 0 creates them, the host hands them to the other ranks."""
        buf = C.create_string_buffer(128)
        _check(self.lib.d2d_comm_unique_id(buf))
        return buf.raw

    def comm_create(self, uid, rank, world):
        """d2d_comm of this context's device (ncclCommInitRank): every rank, same uid."""
        return Comm(self, uid, rank, world)

    # -- buffers --------------------------------------------------------------------
    def dev(self, a, dtype=None):
        """numpy array (or tensor) -> contiguous device tensor."""
        torch = _torch()
        if isinstance(a, torch.Tensor):
            t = a.to(self.device)
        else:
            t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        if dtype is not None:
            t = t.to(dtype)
        return t.contiguous()

    def empty(self, *shape, dtype=None):
        torch = _torch()
        return torch.empty(*shape, dtype=dtype or torch.float64, device=self.device)

    def zeros(self, *shape, dtype=None):
        torch = _torch()
        return torch.zeros(*shape, dtype=dtype or torch.float64, device=self.device)

    # -- plant / guidance -----------------------------------------------------------
    def step(self, X, U, W=(0.0, 0.0), tau_phi=0.01, tau_v=1.0, dt=0.05):
        """X dev [5][n], U dev [2][n] -> Xnext dev [5][n]   (Aircraft.disc_dyn, batched)."""
        n = X.shape[1]
        out = self.empty(5, n)
        _check(self.lib.d2d_step(self.h, n, _ptr(X), _ptr(U), W[0], W[1], tau_phi, tau_v, dt, _ptr(out)))
        return out

    def gvf_run(self, X0, centres, radius, n_ac, n_rows, dt, v_c, ke=4e-4, kd=25.0, kr=20.0,
                B=None, z_des=None, tau_phi=0.01, tau_v=1.0, W=(0.0, 0.0), X0f=None,
                stop_tol=(3.0, 3.0, np.deg2rad(0.5)), rec_stride=1, record=('X', 'U', 'Rr', 'eth'), out=None,
                etheta_tol_deg=None, stop_hold=0):
        """Circular-formation phase for N = n_form*n_ac drones.  X0 dev [5][N], centres dev
        [2][N], radius dev [N].  Returns dict of device tensors (plane-major).  out: the dictionary of an earlier
        call with the same shapes and `record` -- its buffers are written again instead of allocating new ones
        (without the stop rule every recorded row is rewritten; with it, rows behind the stop row keep their old
        content)."""
        torch = _torch()
        N = X0.shape[1]
        assert N % n_ac == 0
        n_form = N // n_ac
        if B is None:
            B = np.zeros((n_ac, max(n_ac - 1, 0)))
            for i in range(n_ac - 1):
                B[i, i] = -1.0; B[i + 1, i] = 1.0
        B = np.ascontiguousarray(B, dtype=np.float64)
        z_des = np.zeros(max(n_ac - 1, 0)) if z_des is None else np.ascontiguousarray(z_des, dtype=np.float64).reshape(-1)
        # stop rule: X0f -> the state rule of case 1; etheta_tol_deg -> the phase-error rule of cases 2 / 3 (+ stop_hold steps)
        use_stop = 2 if etheta_tol_deg is not None else (1 if X0f is not None else 0)
        if use_stop == 2:
            stop_tol = (float(etheta_tol_deg), 0.0, 0.0)
        p = GvfParams(n_form, n_ac, n_rows, rec_stride, dt, tau_phi, tau_v, ke, kd, kr, v_c, W[0], W[1],
                      use_stop, int(stop_hold), (C.c_double * 3)(*stop_tol))
        n_rec = (n_rows + rec_stride - 1) // rec_stride
        if out is None:
            out = {}
            out['X'] = self.zeros(n_rec, 5, N) if 'X' in record else None
            out['U'] = self.zeros(n_rec, 2, N) if 'U' in record else None
            out['Rr'] = self.zeros(n_rec, N) if 'Rr' in record else None
            out['eth'] = self.zeros(n_rec, n_form * max(n_ac - 1, 0)) if ('eth' in record and n_ac > 1) else None
            out['X_final'] = self.empty(5, N)
            out['stop_row'] = torch.empty(n_form, dtype=torch.int32, device=self.device)
            out['conv_row'] = torch.empty(n_form, dtype=torch.int32, device=self.device)
        else:
            assert out['X_final'].shape == (5, N) and all(out[k] is None or out[k].shape[0] == n_rec for k in ('X', 'U', 'Rr', 'eth'))
        _check(self.lib.d2d_sim_gvf_run(self.h, C.byref(p), _ptr(X0), _ptr(centres), _ptr(radius), _hptr(B), _hptr(z_des),
                                        _ptr(X0f if use_stop == 1 else None), _ptr(out['X']), _ptr(out['U']), _ptr(out['Rr']), _ptr(out['eth']),
                                        _ptr(out['X_final']), _ptr(out['stop_row']), _ptr(out.get('conv_row'))))
        return out

    # -- single evaluations behind the reference's per-call helpers --------------------
    def dcf_eval(self, centres, pos, n_ac, B, z_des, kr):
        """centres, pos dev [2][N] -> U_r dev [N], e_theta_deg dev [n_form*(n_ac-1)]."""
        N = pos.shape[1]; n_form = N // n_ac
        B = np.ascontiguousarray(B, dtype=np.float64); z = np.ascontiguousarray(z_des, dtype=np.float64).reshape(-1)
        Ur = self.empty(N); eth = self.empty(max(n_form * (n_ac - 1), 1))
        _check(self.lib.d2d_dcf_eval(self.h, n_form, n_ac, _hptr(B), _hptr(z), kr, _ptr(centres), _ptr(pos), _ptr(Ur), _ptr(eth)))
        return Ur, eth

    def gvf_eval(self, X, e, nvec, H, ke, kd):
        """X dev [5][n], e dev [n], nvec dev [2][n], H dev [4][n] -> dev [3][n] = U, U1, U2."""
        n = X.shape[1]
        U = self.empty(3, n)
        _check(self.lib.d2d_gvf_eval(self.h, n, _ptr(X), _ptr(e), _ptr(nvec), _ptr(H), ke, kd, _ptr(U)))
        return U

    def flatness(self, variant, Yref, W=(0.0, 0.0), tau_phi=0.01, tau_v=1.0):
        """Yref dev [8][n] -> X [5][n], U [2][n], Xdot [5][n]."""
        n = Yref.shape[1]
        X, U, Xd = self.empty(5, n), self.empty(2, n), self.empty(5, n)
        _check(self.lib.d2d_flatness(self.h, variant, n, _ptr(Yref), W[0], W[1], tau_phi, tau_v, _ptr(X), _ptr(U), _ptr(Xd)))
        return X, U, Xd

    def cont_jac(self, Xr, tau_phi=0.01, tau_v=1.0):
        n = Xr.shape[1]
        A, B = self.empty(25, n), self.empty(10, n)
        _check(self.lib.d2d_cont_jac(self.h, n, _ptr(Xr), tau_phi, tau_v, _ptr(A), _ptr(B)))
        return A, B

    def lqr(self, A, B, Q, R):
        """A dev [25][n], B dev [10][n]; Q (5,5), R (2,2) host -> K dev [10][n], P dev [25][n]."""
        n = A.shape[1]
        Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(25); R = np.ascontiguousarray(R, dtype=np.float64).reshape(4)
        K, P = self.empty(10, n), self.empty(25, n)
        _check(self.lib.d2d_lqr(self.h, n, _ptr(A), _ptr(B), _hptr(Q), _hptr(R), _ptr(K), _ptr(P)))
        return K, P

    @staticmethod
    def track_params(n, n_rows, dt, w=(0.0, 0.0), tau_phi=0.01, tau_v=1.0,
                     err_sats=(20, 20, np.pi / 3, np.pi / 4, 1), v_min=4.0, v_max=20.0,
                     phi_lim=np.deg2rad(60), Q=(1, 1, 0.1, 0.01, 0.01), R=(8, 1)):
        return TrackParams(n, n_rows, dt, tau_phi, tau_v, w[0], w[1], (C.c_double * 5)(*err_sats), v_min, v_max,
                           phi_lim, (C.c_double * 5)(*Q), (C.c_double * 2)(*R))

    def ctrl_gain(self, X, Yref, **kw):
        """X dev [5][n], Yref dev [8][n] -> Xr [5][n], dX [5][n], U [2][n], K [10][n]."""
        n = X.shape[1]
        p = self.track_params(n, 1, kw.pop('dt', 0.05), **kw)
        Xr, dX, U, K = self.empty(5, n), self.empty(5, n), self.empty(2, n), self.empty(10, n)
        _check(self.lib.d2d_ctrl_gain(self.h, C.byref(p), _ptr(X), _ptr(Yref), _ptr(Xr), _ptr(dX), _ptr(U), _ptr(K)))
        return Xr, dX, U, K

    def dfff_eval(self, X, Yref, w=(0.0, 0.0), tau_phi=0.01, tau_v=1.0):
        """DFFFController.get for n (state, reference sample) pairs: X dev [5][n], Yref dev [6][n]
        (x,y,xd,yd,xdd,ydd) -> Xr [5][n], U [2][n], K1 [6][n] (row-major 2x3)."""
        n = X.shape[1]
        p = self.track_params(n, 1, 0.01, w=w, tau_phi=tau_phi, tau_v=tau_v, phi_lim=np.deg2rad(45),
                              Q=(1, 1, 0.1, 0.0, 0.0), R=(8, 1))          # src/d2d/guidance.py:79,86
        Xr, U, K = self.empty(5, n), self.empty(2, n), self.empty(6, n)
        _check(self.lib.d2d_dfff_eval(self.h, C.byref(p), _ptr(X), _ptr(Yref), _ptr(Xr), _ptr(U), _ptr(K)))
        return Xr, U, K

    def traj_sample(self, desc, T, t_start, dt):
        """desc dev [n][TRAJ_STRIDE] trajectory descriptors (d2d.trajectory.describe) -> Yref dev [T][6][n]."""
        n = desc.shape[0]
        Y = self.empty(T, 6, n)
        _check(self.lib.d2d_traj_sample(self.h, n, T, float(t_start), float(dt), _ptr(desc), _ptr(Y)))
        return Y

    def dfff_run(self, Yref, X0, dt, perts=None, record=('X', 'U', 'Xr'), w=(0.0, 0.0), tau_phi=0.01, tau_v=1.0, out=None):
        """run_simulation of src/05_test_simulation.py with the legacy DFFFController for n aircraft: Yref dev [T][6][n]
        (x,y,xd,yd,xdd,ydd at the sample times), X0 dev [5][n], perts dev [T][5][n] or None -> dict of device histories
        X [T][5][n], U [T][2][n], Xr [T][5][n] and X_final."""
        T, _, n = Yref.shape
        p = self.track_params(n, T, dt, w=w, tau_phi=tau_phi, tau_v=tau_v, phi_lim=np.deg2rad(45),
                              Q=(1, 1, 0.1, 0.0, 0.0), R=(8, 1))          # src/d2d/guidance.py:79,86
        if out is None:
            out = {k: (self.zeros(T, c, n) if k in record else None) for k, c in (('X', 5), ('U', 2), ('Xr', 5))}
            out['X_final'] = self.empty(5, n)
        _check(self.lib.d2d_sim_dfff_run(self.h, C.byref(p), _ptr(Yref), _ptr(perts), _ptr(X0), _ptr(out['X']), _ptr(out['U']),
                                         _ptr(out['Xr']), _ptr(out['X_final'])))
        return out

    def nlp_solve(self, scen, W, h, partner=None, rho0=10.0, mub0=0.1, mub_min=1e-9, feas_tol=1e-9, opt_tol=1e-7, inner_max=NLP_INNER_MAX,
                  outer_max=NLP_OUTER_MAX, want_mult=False, serial=0, bounds=None, slots=0, order=None):
        """Direct-collocation NLP in node variables (d2d_nlp_solve): scen dev [B][SCEN_STRIDE], W dev [B][5][N] in/out (initial
        guess -> solution), partner dev [B][2][N] or None, bounds dev [B][4] = (phi_lo, phi_hi, psi_lo, psi_hi) or None (d2d_nlp_opts.bounds),
        order dev int32 [B]: the hand-out order of the persistent launch (a permutation; d2d_nlp_opts.order) or None.
        Returns dict(cost, feas, iters, status[, mult [B][3][N]]) of device tensors."""
        torch = _torch()
        B, _, N = W.shape
        assert W.is_contiguous() and scen.shape[0] == B and (partner is None or (partner.is_contiguous() and partner.shape == (B, 2, N)))
        work = self.empty(self.lib.d2d_nlp_workspace_doubles(N) * B)
        cost, feas = self.empty(B), self.empty(B)
        iters = torch.empty(B, dtype=torch.int32, device=self.device); status = torch.empty(B, dtype=torch.int32, device=self.device)
        mult = self.zeros(B, 3, N) if want_mult else None
        assert bounds is None or (bounds.is_contiguous() and tuple(bounds.shape) == (B, 4) and bounds.dtype == _torch().float64)
        assert order is None or (order.is_contiguous() and tuple(order.shape) == (B,) and order.dtype == torch.int32)
        o = NlpOpts(rho0, mub0, mub_min, feas_tol, opt_tol, inner_max, outer_max, serial, int(slots), None if bounds is None else bounds.data_ptr(),
                    None if order is None else order.data_ptr())
        _check(self.lib.d2d_nlp_solve(self.h, B, N, float(h), _ptr(scen), C.byref(o), _ptr(W), _ptr(partner), _ptr(work), _ptr(mult),
                                      _ptr(cost), _ptr(feas), _ptr(iters), _ptr(status)))
        out = dict(cost=cost, feas=feas, iters=iters, status=status, work=work)
        if want_mult:
            out['mult'] = mult
        return out

    def nlp_solve_groups(self, scen, W, h, n_ac, max_sweeps=12, tol=1e-7, rho0=10.0, mub0=0.1, mub_min=1e-9, feas_tol=1e-9, opt_tol=1e-7,
                         inner_max=NLP_INNER_MAX, outer_max=NLP_OUTER_MAX, serial=0, bounds=None):
        """The reference's multi-aircraft Problem for R scenarios in one launch (d2d_nlp_solve_groups): scen dev [R*n_ac][SCEN_STRIDE],
        W dev [R*n_ac][5][N] in/out, the aircraft of a scenario consecutive; CostCollision couples aircraft 0 and 1 (rows' KCOL > 0).
        Returns dict(cost, feas, iters, status per aircraft; sweeps, moved per scenario) of device tensors."""
        torch = _torch()
        B, _, N = W.shape
        assert W.is_contiguous() and scen.shape[0] == B and B % n_ac == 0
        R = B // n_ac
        work = self.empty((self.lib.d2d_nlp_workspace_doubles(N) * n_ac + 2 * N) * R)
        cost, feas, moved = self.empty(B), self.empty(B), self.empty(R)
        iters = torch.empty(B, dtype=torch.int32, device=self.device); status = torch.empty(B, dtype=torch.int32, device=self.device)
        sweeps = torch.empty(R, dtype=torch.int32, device=self.device)
        assert bounds is None or (bounds.is_contiguous() and tuple(bounds.shape) == (B, 4) and bounds.dtype == _torch().float64)
        o = NlpOpts(rho0, mub0, mub_min, feas_tol, opt_tol, inner_max, outer_max, serial, 0, None if bounds is None else bounds.data_ptr(), None)
        _check(self.lib.d2d_nlp_solve_groups(self.h, R, n_ac, N, float(h), _ptr(scen), C.byref(o), int(max_sweeps), float(tol), _ptr(W), _ptr(work),
                                             None, _ptr(cost), _ptr(feas), _ptr(iters), _ptr(status), _ptr(sweeps), _ptr(moved)))
        return dict(cost=cost, feas=feas, iters=iters, status=status, sweeps=sweeps, moved=moved, work=work)

    def track_run(self, x_ref, y_ref, X0, dt, record=('X', 'U', 'Xr', 'dX', 'Yd', 'Ydd'), out=None, **kw):
        """x_ref, y_ref dev [T][n]; X0 dev [5][n] -> dict of device histories (out: reuse the buffers of an earlier
        call with the same shapes and `record`)."""
        T, n = x_ref.shape
        p = self.track_params(n, T, dt, **kw)
        if out is None:
            out = {k: (self.zeros(T, c, n) if k in record else None)
                   for k, c in (('X', 5), ('U', 2), ('Xr', 5), ('dX', 5), ('Yd', 2), ('Ydd', 2))}
            out['X_final'] = self.empty(5, n)
        else:
            assert out['X_final'].shape == (5, n) and all(out[k] is None or out[k].shape[0] == T for k in ('X', 'U', 'Xr', 'dX', 'Yd', 'Ydd'))
        _check(self.lib.d2d_sim_track_run(self.h, C.byref(p), _ptr(x_ref), _ptr(y_ref), _ptr(X0), _ptr(out['X']),
                                          _ptr(out['U']), _ptr(out['Xr']), _ptr(out['dX']), _ptr(out['Yd']),
                                          _ptr(out['Ydd']), _ptr(out['X_final'])))
        return out


_default_ctx = None


def default_context():
    """Process-wide context on cuda:LOCAL_RANK used by the reference-named mirror classes."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get('LOCAL_RANK', 0)))
    return _default_ctx


class FitPlan:
    """Shared basis block + solver scratch for one (S, K, duration, wref)."""

    def __init__(self, ctx, S, K, duration, wref, kernel='auto', long_tables=-1):
        """kernel: 'auto' | 'knot' | 'fused' | 'long' | 'split' (d2d_fit_plan_opts.kernel: which kernel family serves d2d_fit_solve);
        long_tables: d2d_fit_plan_opts.long_tables (-1 = the segment formulation)."""
        self.ctx, self.S, self.K, self.duration = ctx, S, K, float(duration)
        self.nq = 4 * S
        w = np.ascontiguousarray(wref, dtype=np.float64)
        h = _P()
        po = FitPlanOpts(KERNELS[kernel], int(long_tables), (C.c_int32 * 2)(0, 0))
        _check(ctx.lib.d2d_fit_plan_create_ex(ctx.h, S, K, self.duration, _hptr(w), C.byref(po), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, 'h', None):
            self.ctx.lib.d2d_fit_plan_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def kernel(self):
        """'fused' | 'long' | 'split' | 'knot': the kernel d2d_fit_solve runs for this plan with the default solver (include/d2d.h
        D2D_FIT_KERNEL_*; 'knot' = the fused shape in knot coordinates, csrc/fit_knot.hip)."""
        return ('split', 'fused', 'long', 'knot')[self.ctx.lib.d2d_fit_plan_kernel(self.h)]

    def basis(self):
        """Host copies: G (3,K,nq), Gp (3,K,4), Z (8S,nq), Zp (8S,4), Pinit (nq,K)."""
        S, K, nq = self.S, self.K, self.nq
        G = np.empty((3, K, nq)); Gp = np.empty((3, K, 4)); Z = np.empty((8 * S, nq)); Zp = np.empty((8 * S, 4))
        P = np.empty((nq, K))
        _check(self.ctx.lib.d2d_fit_plan_get(self.h, _hptr(G), _hptr(Gp), _hptr(Z), _hptr(Zp), _hptr(P)))
        return G, Gp, Z, Zp, P

    def init(self, scen):
        B = scen.shape[0]
        q = self.ctx.empty(B, 2 * self.nq)
        _check(self.ctx.lib.d2d_fit_init(self.ctx.h, self.h, B, _ptr(scen), _ptr(q)))
        return q

    def project(self, scen, xy):
        """xy dev [B][2][K] node positions -> q0 dev [B][2nq]."""
        B = scen.shape[0]
        q = self.ctx.empty(B, 2 * self.nq)
        _check(self.ctx.lib.d2d_fit_project(self.ctx.h, self.h, B, _ptr(scen), _ptr(xy), _ptr(q)))
        return q

    def eval(self, scen, q, want_H=True):
        torch = _torch()
        B, n = scen.shape[0], 2 * self.nq
        cost, g = self.ctx.empty(B), self.ctx.empty(B, n)
        H = torch.zeros(B, n, n, dtype=torch.float32, device=self.ctx.device) if want_H else None
        _check(self.ctx.lib.d2d_fit_eval(self.ctx.h, self.h, B, _ptr(scen), _ptr(q), _ptr(cost), _ptr(g), _ptr(H)))
        return cost, g, H

    def solve(self, scen, q, max_iter=200, check_every=8, ftol=1e-14, gtol=1e-9, xtol=1e-11, so_lambda=SO_LAMBDA, **mode_kw):
        """In-place LM solve of q.  Returns cost, iters, status (device) and stats (numpy[4]).
        mode_kw: mode (MODE_MINPACK default / MODE_FAST), mp_finish, mp_tol, slice (fit_opts)."""
        torch = _torch()
        B = scen.shape[0]
        cost = self.ctx.empty(B)
        iters = torch.empty(B, dtype=torch.int32, device=self.ctx.device)
        status = torch.empty(B, dtype=torch.int32, device=self.ctx.device)
        stats = np.zeros(4)
        o = fit_opts(max_iter, check_every, ftol, gtol, xtol, so_lambda, **mode_kw)
        _check(self.ctx.lib.d2d_fit_solve(self.ctx.h, self.h, B, _ptr(scen), _ptr(q), C.byref(o), _ptr(cost),
                                          _ptr(iters), _ptr(status), _hptr(stats)))
        return cost, iters, status, stats

    # -- the same loop in parts (for a caller-side / cross-GPU convergence check) -------
    def begin(self, B):
        _check(self.ctx.lib.d2d_fit_begin(self.ctx.h, self.h, B))

    def iterate(self, scen, q, n_iters, max_iter=200, ftol=1e-14, gtol=1e-9, xtol=1e-11, so_lambda=SO_LAMBDA, **mode_kw):
        """Run n_iters more damped solves; returns the number of trajectories still running."""
        o = fit_opts(max_iter, n_iters, ftol, gtol, xtol, so_lambda, **mode_kw)
        running = C.c_int32(0)
        _check(self.ctx.lib.d2d_fit_iterate(self.ctx.h, self.h, scen.shape[0], _ptr(scen), _ptr(q), C.byref(o), n_iters,
                                            C.byref(running)))
        return running.value

    def finish(self, scen, q):
        torch = _torch()
        B = scen.shape[0]
        cost = self.ctx.empty(B)
        iters = torch.empty(B, dtype=torch.int32, device=self.ctx.device)
        status = torch.empty(B, dtype=torch.int32, device=self.ctx.device)
        stats = np.zeros(4)
        _check(self.ctx.lib.d2d_fit_finish(self.ctx.h, self.h, B, _ptr(scen), _ptr(q), _ptr(cost), _ptr(iters),
                                           _ptr(status), _hptr(stats)))
        return cost, iters, status, stats

    def order_from_iters(self, iters):
        """Scheduling hint: hand the fits of the next solves of this batch out longest-first (iters: device int32 [B] of a
        previous solve of the same scenarios)."""
        _check(self.ctx.lib.d2d_fit_plan_set_order(self.ctx.h, self.h, iters.shape[0], _ptr(iters)))

    def clear_order(self):
        _check(self.ctx.lib.d2d_fit_plan_set_order(self.ctx.h, self.h, 0, None))

    def set_handout_prior(self, table=None):
        """Install a hand-out prior (float32 [2][48][12] expected trial counts, d2dhip.handout.fit_prior) or, with None, the built-in one."""
        t = None if table is None else np.ascontiguousarray(table, dtype=np.float32).reshape(2, 48, 12)
        _check(self.ctx.lib.d2d_fit_plan_set_handout_prior(self.ctx.h, self.h, _hptr(t)))

    def learn_handout_prior(self, scen, iters):
        """Calibrate the prior on a finished solve of this workload: scen [B][SCEN_STRIDE], iters [B] (device or host)."""
        from . import handout
        sc = scen.cpu().numpy() if hasattr(scen, 'cpu') else np.asarray(scen)
        it = iters.cpu().numpy() if hasattr(iters, 'cpu') else np.asarray(iters)
        table = handout.fit_prior(sc, self.duration, it)
        self.set_handout_prior(table)
        return table

    def last_order(self, B):
        """The hand-out order the last solve launch used (int32 [B]); raises if it ran in index order."""
        o = np.zeros(B, np.int32)
        _check(self.ctx.lib.d2d_fit_plan_get_order(self.ctx.h, self.h, B, _hptr(o)))
        return o

    def group_order_from_last(self, R, enable=True):
        """Scheduling hint for solve_groups over R scenarios: start the scenarios that swept longest in the LAST solve_groups
        of this plan first (enable=False clears it)."""
        _check(self.ctx.lib.d2d_fit_plan_set_group_order(self.ctx.h, self.h, R, 1 if enable else 0))

    def group_report(self, R):
        """(sweeps int32 [R], last-sweep largest relative move float64 [R]) of the last solve_groups over R scenarios."""
        sw = np.zeros(R, np.int32); mv = np.zeros(R)
        _check(self.ctx.lib.d2d_fit_group_report(self.ctx.h, self.h, R, _hptr(sw), _hptr(mv)))
        return sw, mv

    def set_groups(self, n_ac):
        _check(self.ctx.lib.d2d_fit_plan_set_groups(self.h, n_ac))
        self.n_group = n_ac

    def solve_groups(self, scen, q, n_ac, max_sweeps=60, inner_iters=8, tol=1e-12, ftol=1e-14, gtol=1e-9, xtol=1e-11, **gs_kw):
        """Block Gauss-Seidel over the aircraft of every group (scen / q rows g*n_ac + i).  Returns the
        per-aircraft sub-problem costs (device), sweeps used and stats (numpy[4])."""
        if getattr(self, 'n_group', 1) != n_ac:
            self.set_groups(n_ac)
        B = scen.shape[0]
        assert B % n_ac == 0
        cost = self.ctx.empty(B)
        o = fit_opts(inner_iters, 1, ftol, gtol, xtol, 0.0, **gs_kw)      # gs_kw: gs_ls, gs_ls_s0, gs_ls_r0, gs_prio_at, gs_pairs
        sw = C.c_int32(0)
        stats = np.zeros(4)
        _check(self.ctx.lib.d2d_fit_solve_groups(self.ctx.h, self.h, B // n_ac, _ptr(scen), _ptr(q), C.byref(o), max_sweeps,
                                                 inner_iters, tol, _ptr(cost), C.byref(sw), _hptr(stats)))
        return cost, sw.value, stats

    def profile(self, enable):
        _check(self.ctx.lib.d2d_fit_profile(self.h, 1 if enable else 0))

    def rows(self, scen, q):
        """Residual rows at q: cost, J^T r (device); the fp32 row records stay in the plan's scratch for jtj()."""
        B, n = scen.shape[0], 2 * self.nq
        cost, g = self.ctx.empty(B), self.ctx.empty(B, n)
        _check(self.ctx.lib.d2d_fit_rows(self.ctx.h, self.h, B, _ptr(scen), _ptr(q), _ptr(cost), _ptr(g)))
        return cost, g

    def jtj(self, B, want_H=True):
        """J^T J from the records of the last rows(): the contraction-only launch.  H dev [B][2nq][2nq] or None."""
        torch = _torch()
        n = 2 * self.nq
        H = torch.zeros(B, n, n, dtype=torch.float32, device=self.ctx.device) if want_H else None
        _check(self.ctx.lib.d2d_fit_jtj(self.ctx.h, self.h, B, _ptr(H)))
        return H

    def profile_read(self):
        """(eval_ms, eval_launches, step_ms, step_launches, fused_lm_ms, fused_lm_launches, jtj_ms, jtj_launches) from HIP
        events."""
        out = np.zeros(8)
        _check(self.ctx.lib.d2d_fit_profile_read(self.h, _hptr(out)))
        return out

    def coeffs(self, scen, q):
        B = scen.shape[0]
        z = self.ctx.empty(B, 2, self.S, 8)
        _check(self.ctx.lib.d2d_fit_coeffs(self.ctx.h, self.h, B, _ptr(scen), _ptr(q), _ptr(z)))
        return z

    def sample(self, scen, q):
        B = scen.shape[0]
        Y, Xs = self.ctx.empty(B, 6, self.K), self.ctx.empty(B, 5, self.K)
        _check(self.ctx.lib.d2d_fit_sample(self.ctx.h, self.h, B, _ptr(scen), _ptr(q), _ptr(Y), _ptr(Xs)))
        return Y, Xs
