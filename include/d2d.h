/*
 * libd2dhip.so -- C ABI of the MI355X (gfx950) batched trajectory-fit and
 * guidance-simulation engine.
 *
 * The reference (rajashree-srikanth/drone-sim-python) is pure Python and has no FFI; each
 * entry point below names the Python interface it replaces (file:line relative to the
 * reference's repository root) and is what a ctypes binding on the reference side would
 * bind (INTEGRATION.md shows the stubs).
 *
 * Conventions
 *   - every function returns 0 on success or a negative D2D_E* code; d2d_last_error()
 *     returns a thread-local message for the last failure.  No exception crosses the ABI.
 *   - "dev" pointers are device (HBM) addresses owned by the caller (e.g.
 *     torch.Tensor.data_ptr() of a contiguous ROCm tensor); "host" pointers are ordinary
 *     host memory.  The library allocates only scratch tied to a context or a fit plan.
 *   - all work is enqueued on the hipStream_t given to d2d_ctx_create and is asynchronous
 *     unless stated; the caller synchronises the stream.
 *   - one context per GPU per host thread; calls are re-entrant across contexts.
 *   - device arrays are "plane-major" (structure of arrays): a state history is
 *     [row][component][drone] so that one wavefront writes 64 consecutive doubles.
 */
#ifndef D2D_H
#define D2D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D2D_VERSION 109

/* error codes */
#define D2D_OK 0
#define D2D_EINVAL (-1)   /* bad argument (shape, null pointer, unsupported size) */
#define D2D_EHIP (-2)     /* a HIP runtime call failed */
#define D2D_ENOMEM (-3)
#define D2D_ESTATE (-4)   /* call order / plan mismatch */

typedef struct d2d_ctx d2d_ctx;
typedef struct d2d_fit_plan d2d_fit_plan;

int d2d_version(void);
const char *d2d_last_error(void);

/* device = HIP device ordinal; stream = hipStream_t (NULL = the default stream). */
int d2d_ctx_create(int device, void *stream, d2d_ctx **out);
int d2d_ctx_destroy(d2d_ctx *ctx);
int d2d_ctx_sync(d2d_ctx *ctx);   /* hipStreamSynchronize on the context's stream */

/* Multi-GPU convergence exchange (one process per GPU; BASELINE configs[3]: trajectories shard by rank, the only collective is this
 * one).  The reference has no counterpart (it is single-process Python); SURVEY.md 8b asks for the entry point so that a host
 * without torch.distributed can drive the sharded solve.  RCCL is loaded at run time (dlopen of librccl.so.1): the library has
 * no link-time dependency on it.
 *   d2d_comm_available  local, no communication: D2D_OK when this process can load RCCL and resolve the symbols used below.
 *                       d2d_comm_create is itself a collective (ncclCommInitRank returns when ALL ranks have joined): a host
 *                       agrees on this flag over its own channel first and calls d2d_comm_create only if every rank reported
 *                       D2D_OK -- a rank that cannot get there would leave the others waiting inside it;
 *   d2d_comm_unique_id  rank 0 creates the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by whatever channel the
 *                       host has (a file, MPI, a TCP store);
 *   d2d_comm_create     every rank, same id: ncclCommInitRank on the context's device;
 *   d2d_allreduce_stats stats dev double[3] in place, enqueued on the context's stream as ONE grouped exchange:
 *                       [0] sum (cost), [1] max (|J^T r|_inf), [2] sum (trajectories still running) -- the three scalars
 *                       d2d_fit_finish reports in stats[0..2].  The caller synchronises the stream before reading them. */
typedef struct d2d_comm d2d_comm;
#define D2D_COMM_ID_BYTES 128
int d2d_comm_available(void);
int d2d_comm_unique_id(void *id_out);
int d2d_comm_create(d2d_ctx *ctx, const void *id, int rank, int world, d2d_comm **out);
int d2d_comm_destroy(d2d_comm *comm);
int d2d_comm_info(const d2d_comm *comm, int32_t *rank, int32_t *world);   /* ncclCommUserRank / ncclCommCount of the communicator (either may be NULL); D2D_ESTATE if they differ from what d2d_comm_create was given */
int d2d_allreduce_stats(d2d_ctx *ctx, d2d_comm *comm, double *stats);     /* D2D_EINVAL if stats is not device memory */

/* ------------------------------------------------------------------------------------
 * Plant and guidance (fp64).  State components: x, y, psi, phi, v; inputs: phi_c, v_c.
 * ------------------------------------------------------------------------------------ */

/* Gauss-Legendre Runge-Kutta quadrature of the plant step: D2D_GL_PANELS panels on
 * dt*[edge_p, edge_p+1], D2D_GL_STAGES stages each; phi and v (first-order lags under
 * a zero-order-hold input) are integrated exactly. */
#define D2D_GL_STAGES 4
#define D2D_GL_PANELS 5
/* ... and a step that starts within D2D_GL_FAST_DPHI (rad) of its bank command -- every step of a settled guidance loop -- takes ONE
 * panel of D2D_GL_FAST_STAGES stages over the whole step instead (the boundary layer it has to resolve is that small) */
#define D2D_GL_FAST_STAGES 6
#define D2D_GL_FAST_DPHI 0.02
#define D2D_GL_FAST_RATIO 6.0      /* ... and only while dt <= GL_FAST_RATIO * tau_phi: the one panel integrates exp(-t / tau_phi) over the whole step
                                      (error at the threshold: 2e-10 at dt = 5 tau_phi, 7e-8 at 10 tau_phi, 8e-6 at 20 tau_phi; the graded panels: <= 1e-9) */

/* One batched plant step.  Replaces Aircraft.disc_dyn(Xk, Uk, W, t, dt)
 * (src/d2d/dynamic.py:25-28; model :14-23, heading wrap :27 / src/d2d/utils.py:7).
 * X, Xout: dev [5][n]; U: dev [2][n]; wind (wx, wy) constant. */
int d2d_step(d2d_ctx *ctx, int n, const double *X, const double *U, double wx, double wy,
             double tau_phi, double tau_v, double dt, double *Xout);

typedef struct {
  int32_t n_form;      /* independent formations                                   */
  int32_t n_ac;        /* aircraft per formation, 1..64                            */
  int32_t n_rows;      /* rows of the time grid, len(arange(t_start,t_end,t_step)) */
  int32_t rec_stride;  /* history rows are kept for i % rec_stride == 0 (>=1)      */
  double dt, tau_phi, tau_v;
  double ke, kd, kr;   /* GVF / DCF gains (src/11_full_sim_case1.py:108-110)       */
  double v_c;          /* commanded airspeed (the `v` argument, :93)               */
  double wx, wy;       /* wind                                                     */
  int32_t use_stop;    /* 1: the state rule of case 1 (:140,170-175): every aircraft within stop_tol of X0f;
                          2: the phase-error rule of cases 2 / 3 (src/12_full_sim_case2.py:156-164,
                          src/12_full_sim_case3.py:163-178): every inter-vehicle phase error (degrees,
                          signed) <= stop_tol[0]                                      */
  int32_t stop_hold;   /* rule 2: the loop goes on for this many further steps on which the rule holds
                          (case 3 waits t_opt_comp = 0.7 s for the planner: 14 steps); 0 = case 2 */
  double stop_tol[3];  /* rule 1: |x|,|y|,|psi| tolerances, reference: 3, 3, 0.5 deg; rule 2: [0] = degrees */
} d2d_gvf_params;

/* Circular-formation phase: DCF phase consensus + GVF circle following + plant step, the
 * whole time loop on the device.  Replaces CircularFormationGVF(c, r, v, n_ac, X0f, ...)
 * (src/11_full_sim_case1.py:93-177; identical loop src/09_CircularFormation_diffcentre.py
 * :91-115) and the per-step calls it makes: DCFController.get (src/d2d/guidance.py:103-126),
 * CircleTraj.get (:137-146), GVFcontroller.get (:155-181), atan(U/9.81) (11_*:160),
 * Aircraft.disc_dyn (src/d2d/dynamic.py:25-28).
 * N = n_form*n_ac drones, drone d = formation d/n_ac, aircraft d%n_ac.
 *   X0       dev [5][N]   initial states (row 0 of the history)
 *   centres  dev [2][N]   circle centre of each drone
 *   radius   dev [N]      nominal radius r
 *   Bmat     host [n_ac][n_ac-1] incidence matrix (ConstructBMatrix, 11_*:81-91)
 *   z_des    host [n_ac-1] desired inter-vehicle angles
 *   X0f      dev [3][N] or NULL: x,y,psi targets of the stop rule
 *   X_hist   dev [n_rec][5][N] or NULL; U_hist dev [n_rec][2][N] or NULL (row i-1 holds the
 *            input applied during step i, as the reference writes U_array[i-1]);
 *   Rr_hist  dev [n_rec][N] or NULL (commanded radius, row i); eth_hist dev
 *            [n_rec][n_form*(n_ac-1)] or NULL (phase errors in degrees, row i)
 *            with n_rec = ceil(n_rows / rec_stride)
 *   X_final  dev [5][N]: state after the last executed step
 *   stop_row dev int32 [n_form]: rows kept by the reference's trim when its `break` fires (rule 1 breaks at the top of
 *            step i: i rows; rule 2 at the end of step i: i + 1 rows), or n_rows if the rule never fired;
 *   conv_row dev int32 [n_form] or NULL: rule 2, the reference's `index` = i - 1 of the first step on which the rule held
 *            (-1: never).
 * A formation that has stopped is frozen (its rows beyond stop_row are not written). */
int d2d_sim_gvf_run(d2d_ctx *ctx, const d2d_gvf_params *p, const double *X0,
                    const double *centres, const double *radius, const double *Bmat,
                    const double *z_des, const double *X0f, double *X_hist, double *U_hist,
                    double *Rr_hist, double *eth_hist, double *X_final, int32_t *stop_row, int32_t *conv_row);

typedef struct {
  int32_t n;           /* drones (independent)                                     */
  int32_t n_rows;      /* rows of the reference time grid                          */
  double dt, tau_phi, tau_v;
  double wx, wy;       /* wind handed to DiffController(w) and to the plant        */
  double err_sats[5];  /* src/Controllers.py:147                                   */
  double v_min, v_max, phi_lim;        /* :148-149                                 */
  double q_diag[5], r_diag[2];         /* :152                                     */
} d2d_track_params;

/* One batched evaluation of DiffController.ComputeGain (src/Controllers.py:159-186):
 * ComputeFlatness (:62-108), error wrap/clip (:165-169), Aircraft.cont_jac
 * (src/d2d/dynamic.py:32-43), control.lqr (:174; 5x5 continuous algebraic Riccati
 * equation solved on the device), U = clip(Ur - K dX) (:183-185).
 *   X dev [5][n]; Yref dev [8][n] = x,y,xd,yd,xdd,ydd,xddd,yddd
 *   outputs (dev, any may be NULL): Xr [5][n], dX [5][n], U [2][n], Kgain [10][n]
 *   (K row-major 2x5 per drone, plane-major across drones). */
int d2d_ctrl_gain(d2d_ctx *ctx, const d2d_track_params *p, const double *X, const double *Yref,
                  double *Xr, double *dX, double *U, double *Kgain);

/* One batched evaluation of the legacy DFFFController.get(X, t) (src/d2d/guidance.py:62-91):
 * DiffFlatness.state_and_input_from_output (:22-47) of the reference sample, error wrap (psi) and clip
 * (:70-72), Aircraft.cont_jac, control.lqr on the 3-state sub-system A[:3,:3], A[:3,3:] (:78-81; 3x3
 * Riccati equation solved on the device), U = clip(Ur - K dX) (:85-88).  Of d2d_track_params it reads n,
 * tau_phi, tau_v, wx, wy, err_sats, q_diag[0..2], r_diag, phi_lim, v_min, v_max.
 *   X dev [5][n]; Yref dev [6][n] = x,y,xd,yd,xdd,ydd (the sample traj.get(t) of every drone)
 *   outputs (dev, any may be NULL): Xr [5][n], U [2][n], Kgain [6][n] (K1 row-major 2x3; the phi and v
 *   columns of the reference's 2x5 K are zero). */
int d2d_dfff_eval(d2d_ctx *ctx, const d2d_track_params *p, const double *X, const double *Yref,
                  double *Xr, double *U, double *Kgain);

/* Trajectory-tracking phase, whole time loop on the device.  Replaces
 * implement_controller(n_ac, time, x_ref, y_ref, v, w, X0s)
 * (src/11_full_sim_case1.py:241-291) including ComputeDerivatives (:197-204; two passes
 * of numpy.gradient(edge_order=2)/dt per axis).
 *   x_ref, y_ref dev [n_rows][n]; X0 dev [5][n]
 *   X_hist dev [n_rows][5][n]; U_hist [n_rows][2][n]; Xr_hist, dX_hist [n_rows][5][n];
 *   Yd_hist, Ydd_hist [n_rows][2][n] (any history may be NULL); X_final dev [5][n].
 * Row conventions as the reference: step i uses reference sample i and state i-1 and
 * writes U, dX, Xr, Yd, Ydd at row i-1 and X at row i. */
int d2d_sim_track_run(d2d_ctx *ctx, const d2d_track_params *p, const double *x_ref,
                      const double *y_ref, const double *X0, double *X_hist, double *U_hist,
                      double *Xr_hist, double *dX_hist, double *Yd_hist, double *Ydd_hist,
                      double *X_final);

/* The legacy simulation loop run_simulation(time, aircraft, windfield, ctl, X0, perts)
 * (src/05_test_simulation.py:21-34) with DFFFController (src/d2d/guidance.py:52-91) for n independent aircraft:
 *   U[i-1] = ctl.get(X[i-1], t[i-1]); X[i] = disc_dyn(X[i-1], U[i-1], w, t[i-1], dt) + perts[i]; U[T-1] = ctl.get(X[T-1]).
 *   Yref dev [n_rows][6][n]: the trajectory's flat outputs at the sample times (x, y, xd, yd, xdd, ydd: what
 *   traj.get(t) returns, without the jerk the controller never reads); perts dev [n_rows][5][n] or NULL;
 *   X0 dev [5][n]; X_hist dev [n_rows][5][n], U_hist [n_rows][2][n], Xr_hist [n_rows][5][n] (any may be NULL);
 *   X_final dev [5][n] or NULL.  Controller constants as d2d_dfff_eval (p->q_diag[0..2], r_diag, err_sats, limits). */
int d2d_sim_dfff_run(d2d_ctx *ctx, const d2d_track_params *p, const double *Yref, const double *perts,
                     const double *X0, double *X_hist, double *U_hist, double *Xr_hist, double *X_final);

/* Reference trajectories of the legacy simulations, sampled on the device: Yref [T][6][n] (x, y, xd, yd, xdd, ydd at
 * t_start + i dt) -- the input of d2d_sim_dfff_run -- for n trajectories described by desc dev [n][D2D_TRAJ_STRIDE]:
 *   desc[0] = number of segments (1..D2D_TRAJ_MAX_SEG), desc[1] = the composite's t0, desc[2] = its total duration,
 *   desc[3] = 1: periodic composite (CompositeTraj.get, src/d2d/trajectory.py:202-208: fmod(t - t0, duration), first segment
 *   whose cumulative end is beyond the lapse) / 0: one plain segment evaluated at t;
 *   then per segment D2D_TRAJ_SEG_STRIDE doubles: type, the segment's t0, its cumulative end, parameters:
 *     D2D_TRAJ_LINE    p1x, p1y, unx, uny, v                      TrajectoryLine (src/d2d/trajectory.py:125-141)
 *     D2D_TRAJ_CIRCLE  cx, cy, r, omega = v/r, alpha0             TrajectoryCircle (:143-160)
 *     D2D_TRAJ_SLALOM  p1x, p1y, unx, uny, v, a, om, phi          TrajSlalom (src/d2d/trajectory_factory.py:113-137)
 *     D2D_TRAJ_POLY    coefs[0,:] of the x polynomial (8), of y (8)  MinSnapPoly (src/d2d/trajectory.py:166-187)
 * (d2d/trajectory.py `describe` builds these rows from the mirrored trajectory classes). */
#define D2D_TRAJ_MAX_SEG 8
#define D2D_TRAJ_SEG_STRIDE 20
#define D2D_TRAJ_STRIDE (4 + D2D_TRAJ_MAX_SEG * D2D_TRAJ_SEG_STRIDE)
enum { D2D_TRAJ_LINE = 1, D2D_TRAJ_CIRCLE = 2, D2D_TRAJ_SLALOM = 3, D2D_TRAJ_POLY = 4 };
int d2d_traj_sample(d2d_ctx *ctx, int n, int T, double t_start, double dt, const double *desc, double *Yref);

/* Single batched evaluations behind the reference's per-call helper methods (the time
 * loops above fuse them; these exist so that host code written against the reference's
 * classes reaches the same device functions).
 *  d2d_dcf_eval : DCFController.get(n_ac, B, c, p, z_des, kr) (src/d2d/guidance.py:103-126);
 *                 centres, pos dev [2][N]; U_r dev [N]; eth_deg dev [n_form*(n_ac-1)] or NULL.
 *  d2d_gvf_eval : GVFcontroller.get(X, ke, kd, e, n, H) (:155-181); X dev [5][n], e dev [n],
 *                 nvec dev [2][n], H dev [4][n] (row-major 2x2); U dev [3][n] = U, U1, U2.
 *  d2d_flatness : variant 0 = DiffFlatness.state_and_input_from_output (:22-47),
 *                 variant 1 = DiffFlatness.ComputeFlatness (src/Controllers.py:62-108);
 *                 Yref dev [8][n]; X dev [5][n], U dev [2][n], Xdot dev [5][n] or NULL.
 *  d2d_cont_jac : Aircraft.cont_jac (src/d2d/dynamic.py:32-43); A dev [25][n], B dev [10][n].
 *  d2d_lqr      : control.lqr(A, B, Q, R) (call sites src/Controllers.py:174,
 *                 src/d2d/guidance.py:76,80) for 5-state / 2-input systems; Q host [25],
 *                 R host [4]; K dev [10][n], P dev [25][n] or NULL. */
int d2d_dcf_eval(d2d_ctx *ctx, int n_form, int n_ac, const double *Bmat, const double *z_des, double kr,
                 const double *centres, const double *pos, double *U_r, double *eth_deg);
int d2d_gvf_eval(d2d_ctx *ctx, int n, const double *X, const double *e, const double *nvec, const double *H,
                 double ke, double kd, double *U);
int d2d_flatness(d2d_ctx *ctx, int variant, int n, const double *Yref, double wx, double wy, double tau_phi,
                 double tau_v, double *X, double *U, double *Xdot);
int d2d_cont_jac(d2d_ctx *ctx, int n, const double *Xr, double tau_phi, double tau_v, double *A, double *B);
int d2d_lqr(d2d_ctx *ctx, int n, const double *A, const double *B, const double *Q, const double *R, double *K,
            double *P);

/* ------------------------------------------------------------------------------------
 * Polynomial trajectory fit: S segments x 2 axes x 8 monomial coefficients in the layout
 * of PolynomialOne.coefs[0,:] (src/d2d/trajectory.py:47-72), K samples on
 * linspace(t0,t1,K) (planner_timing, src/d2d/opty_utils.py:8-14), flat outputs -> (va,phi)
 * through DiffFlatness.state_and_input_from_output (src/d2d/guidance.py:22-47), residual
 * rows from CostInput (src/d2d/opty_utils.py:85-97) and CostObstacle kind 1 (:99-134),
 * waypoints from triangle() (:171-187).  The C^3 junction conditions and the end
 * conditions (x,y,psi)(t0),(t1) (src/single_opt_planner.py:46-49, speed = vref) are
 * eliminated: z_axis = Zp d_axis + Z q_axis with nq = 4*S reduced unknowns per axis.
 * ------------------------------------------------------------------------------------ */

/* one scenario row: double[D2D_SCEN_STRIDE] per trajectory (dev [B][D2D_SCEN_STRIDE]) */
#define D2D_SCEN_STRIDE 80
#define D2D_MAX_OBS 16         /* static obstacles per trajectory (CostObstacles, src/d2d/opty_utils.py:136-147;
                                 the reference's largest scenario list, exp_5, holds 12)                   */
enum {
  D2D_SC_X0 = 0, D2D_SC_Y0, D2D_SC_PSI0, D2D_SC_X1, D2D_SC_Y1, D2D_SC_PSI1,
  D2D_SC_VREF,  /* end-condition speed and 'tri' waypoint speed                */
  D2D_SC_VSP,   /* CostInput.vsp                                               */
  D2D_SC_KV, D2D_SC_KPHI, D2D_SC_KOBS,
  D2D_SC_S,     /* obj_scale / K                                               */
  D2D_SC_WWP,   /* waypoint row weight                                         */
  D2D_SC_WX, D2D_SC_WY,   /* wind                                              */
  D2D_SC_GOLEFT,          /* triangle(go_left)                                 */
  D2D_SC_O0X, D2D_SC_O0Y, D2D_SC_O0R,   /* obstacle 0 (r<=0: absent)           */
  D2D_SC_O1X, D2D_SC_O1Y, D2D_SC_O1R,   /* obstacle 1                          */
  D2D_SC_WBND,  /* weight of the soft bound rows                                 */
  D2D_SC_PHIMAX, D2D_SC_VMIN, D2D_SC_VMAX,   /* bounds of those rows: |phi| <= PHIMAX,
                   VMIN <= va <= VMAX (phi_constraint / v_constraint of the scenario)   */
  D2D_SC_KCOL, D2D_SC_RCOL,  /* collision rows between the aircraft of one group: weight, radius */
  D2D_SC_SCOL,  /* their scale, obj_scale / K (src/d2d/multiopty_utils.py:132: no 1/n_ac)  */
  D2D_SC_PMASK, /* bit j set: coupled with aircraft j of the same group (stored as a double) */
  D2D_SC_OKIND, /* bit i set: obstacle i is CostObstacle kind 0, e = clip(exp(r^2 - d^2), 0, 1e3)
                   (src/d2d/opty_utils.py:108-111); clear: kind 1 (stored as a double)          */
  D2D_SC_BANKMAX, /* != 0: CostBank(use_mean=False): obj_scale*max(phi^2) instead of the mean (:72-73) */
  D2D_SC_OEXT,    /* obstacles 2 .. D2D_MAX_OBS-1: (x, y, r) of obstacle i at D2D_SC_OEXT + 3*(i-2) (r<=0: absent;
                     D2D_SC_OKIND bit i)                                                                        */
  D2D_SC_XMIN = D2D_SC_OEXT + 3 * (D2D_MAX_OBS - 2),
  D2D_SC_XMAX, D2D_SC_YMIN, D2D_SC_YMAX
                  /* x_constraint / y_constraint boxes of the scenario (src/single_opt_planner.py:56-57, src/multi_opt_planner.py:63-64): soft bound
                     rows w_b*dist(x, [XMIN, XMAX]), w_b*dist(y, [YMIN, YMAX]) with the weight D2D_SC_WBND of the
                     phi / v rows; an axis with MIN >= MAX (e.g. both zero) has no box.  The last two columns of
                     the row are reserved (zero).                                                                */
};
#define D2D_FIT_NROW 8        /* residual rows per sample */
#define D2D_FIT_MAX_S 6
#define D2D_FIT_NQ_MAX 24     /* 4*S */

/* Levenberg-Marquardt constants (oracle/fit.py LM_*) */
#define D2D_LM_LAMBDA0 1e-3
#define D2D_LM_LAMBDA_MIN 1e-12
#define D2D_LM_LAMBDA_MAX 1e12
#define D2D_LM_DIAG_FLOOR 1e-30
#define D2D_LM_SO_LAMBDA 1e-4     /* default of d2d_fit_opts.so_lambda */
/* A step whose gain ratio is not positive is shortened along its direction before the damping grows: fraction = minimiser of
 * the parabola through the cost at 0 (value, slope) and at 1, clipped to [BT_MIN, BT_MAX]; then BT_SHRINK of it, not below
 * BT_FLOOR.  A failed factorisation (indefinite exact Hessian) multiplies the damping by FAIL_MULT. */
#define D2D_LM_BT_MIN 0.1
#define D2D_LM_BT_MAX 0.5
#define D2D_LM_BT_SHRINK 0.25
#define D2D_LM_BT_FLOOR 0.02
#define D2D_LM_FAIL_MULT 8.0
/* block Gauss-Seidel over coupled aircraft: a scenario still sweeping after GS_PRIO_AT sweeps raises its wave's priority */
#define D2D_GS_PRIO_AT 40
/* ... and its slow sweeps are followed by a line search on the JOINT cost (own rows of every aircraft + every coupled pair once) along
 * the sweep's direction: from sweep GS_LS_SWEEP0 on, after a sweep that moved >= GS_LS_RATIO x the move of the sweep before it; first
 * length rho / (1 - rho) clipped to [1, GS_LS_FIRST_MAX] (GS_LS_FIRST_MAX when the moves grow), doubled while the joint cost falls (up
 * to GS_LS_MAX), one try at a quarter when the first does not lower it (oracle/fit.py bgs_solve; persistent kernel only) */
#define D2D_GS_LS_SWEEP0 8
#define D2D_GS_LS_RATIO 0.8
#define D2D_GS_LS_FIRST_MAX 8.0
#define D2D_GS_LS_MAX 64.0
enum { D2D_ST_RUNNING = 0, D2D_ST_CONVERGED = 1, D2D_ST_MAXITER = 2, D2D_ST_NONFINITE = 3, D2D_ST_STALLED = 4 };

/* Solver of d2d_fit_solve / d2d_fit_iterate (persistent LM kernel; oracle/fit.py states both on the CPU):
 *  D2D_LM_MODE_MINPACK (default)  MINPACK's lmder restated on the normal equations -- the path the CPU arbiter
 *      scipy.optimize.least_squares(method='lm') follows (unit scaling, factor 100; trust region on the Gauss-Newton model,
 *      lmpar's Newton iteration on the damping, lmder's ratio test and radius update), every decision taken from the Cholesky
 *      factor of J^T J + par I instead of the QR factors of J.  Once the trust region has been inactive for mp_finish accepted
 *      steps in a row (par = 0 and ratio >= 0.75: the basin is decided and Gauss-Newton crawls at its linear rate) the fit is
 *      handed to the second-order loop below, started at lambda0, for the quadratic finish; mp_finish = 0 never hands over
 *      (pure lmder, stopping on mp_ftol / mp_xtol / mp_gtol exactly as MINPACK does).  The same hand-over happens when lmder has
 *      STAGNATED: mp_slow trials in a row, accepted or not, each changed the cost by no more than D2D_LM_MP_SLOW_TOL of itself
 *      without meeting lmder's own stopping tests -- the zig-zag of Gauss-Newton on a large-residual fit (the full step
 *      overshoots, the radius shrinks, a damped step gains 1e-7 ...), which scipy follows for hundreds of evaluations (736 on one
 *      121-node bench scenario) and which is never "calm"; the exact Hessian ends it in a handful of iterations at the same minimum.
 *  D2D_LM_MODE_FAST  Nielsen's gain-ratio damping on (H + lam diag|H|) with shortened steps along a rejected direction and
 *      the second-order term once lam <= so_lambda (rounds 1-2): fewer factorisations per fit, but it reaches another local
 *      minimum than scipy on ~13 % of the synthetic bench scenarios. */
enum { D2D_LM_MODE_MINPACK = 0, D2D_LM_MODE_FAST = 1 };
#define D2D_LM_MP_FINISH 3       /* default of d2d_fit_opts.mp_finish */
#define D2D_LM_MP_SLOW 8         /* default of d2d_fit_opts.mp_slow */
#define D2D_LM_MP_SLOW_TOL 1e-4  /* a trial is "slow" when it changes the cost by no more than this fraction (|actred| of lmder) */
#define D2D_LM_SLICE 0           /* default of d2d_fit_opts.slice */
typedef struct {
  int32_t max_iter;     /* trial points (damped solves in FAST mode) per trajectory (default 200)      */
  int32_t check_every;  /* host convergence poll period in iterations (default 8)      */
  double ftol, gtol, xtol;   /* second-order / FAST loop: defaults 1e-14, 1e-9, 1e-11                  */
  double so_lambda;     /* FAST mode: once the damping has fallen to this value the next evaluation carries the
                           second-order term sum_i r_i Hessian(r_i) (exact Hessian of 0.5 sum r^2) beside
                           J^T J: Gauss-Newton's linear rate on these large-residual fits becomes
                           quadratic.  0 = Gauss-Newton only; default D2D_LM_SO_LAMBDA.  Persistent LM
                           kernels only; the launch-pair path ignores it.  (The finish of MINPACK mode always
                           carries the term.)                                                          */
  int32_t mode;         /* D2D_LM_MODE_*: both persistent kernels (K <= 64 fused, and the long-horizon / S != 6 one) honour it;
                           the launch-pair path and coupled groups run FAST                              */
  int32_t mp_finish;    /* MINPACK mode: calm steps before the second-order finish (default D2D_LM_MP_FINISH; 0 = never) */
  double mp_ftol, mp_xtol, mp_gtol;   /* lmder's ftol / xtol / gtol (default 1e-15 each: what bench.py's scipy leg uses) */
  int32_t slice;        /* scheduling of the persistent kernel, > 0: a fit that has run this many iterations while other fits
                           are waiting for a wavefront goes to the back of a device-wide ring (its state is saved, another
                           wavefront resumes it), so that every fit of a batch advances at the same rate whatever the order
                           they were handed out in.  Results are bit-identical.  Measured: equal shares end
                           a 4096-fit launch at (longest fit) + (the excess of the first rounds) -- 1.5 % sooner than running
                           every fit to its end in index order, 4 % later at 32 768 fits -- so the default is 0 = off     */
  int32_t mp_slow;      /* MINPACK mode: stagnating trials in a row before the second-order finish (default D2D_LM_MP_SLOW; 0 = never;
                           only with mp_finish > 0).  (Was `reserved`, always 0, up to version 106.)                          */
  /* ---- version 108: what used to be process-wide environment switches (D2D_LM_PRIO_AT, D2D_GROUPS_LS*, D2D_GROUPS_PRIO_AT,
   * D2D_GROUPS_PAIRS) is part of the call; d2d_fit_opts_default() fills every field with the defaults named here ---- */
  int32_t handout;      /* order in which the persistent kernels hand the fits of a batch to their wavefronts when the batch is larger
                           than the resident wavefronts (results never depend on it: the fits are independent):
                           D2D_HANDOUT_PREDICTED (default) longest-first by a PREDICTED trial count -- a key computed on the device from
                           each scenario row alone, inside the solve (d2d_fit_plan_set_handout_prior; nothing is known from earlier
                           solves); D2D_HANDOUT_INDEX index order.  An explicit d2d_fit_plan_set_order hint overrides both.        */
  int32_t prio_at;      /* a fit that has used this many trial points raises its wave's priority (s_setprio): the stragglers get the
                           SIMD's issue slots ahead of their co-resident wave (default D2D_LM_PRIO_AT; large = never)               */
  int32_t gs_ls;        /* d2d_fit_solve_groups: 1 (default) line search on the joint cost along slow sweeps, 0 plain block Gauss-Seidel */
  int32_t gs_ls_s0;     /* ... from this sweep on (default D2D_GS_LS_SWEEP0; values < 2 are raised to 2)                             */
  double gs_ls_r0;      /* ... after a sweep that moved >= this fraction of the move of the sweep before (default D2D_GS_LS_RATIO)   */
  int32_t gs_prio_at;   /* a scenario still sweeping after this many sweeps raises its wave's priority (default D2D_GS_PRIO_AT)      */
  int32_t gs_pairs;     /* d2d_fit_solve_groups on plans of the long-horizon kernel: 1 = every visit as launch pairs of the split
                           kernels (where their LDS image holds K) instead of ONE launch of the long kernel (default 0)              */
} d2d_fit_opts;
enum { D2D_HANDOUT_INDEX = 0, D2D_HANDOUT_PREDICTED = 1 };
#define D2D_LM_PRIO_AT 48        /* default of d2d_fit_opts.prio_at */
/* *o = the library's defaults (what a NULL opts pointer means).  Fill it first, then change fields: a struct built by hand from an
 * older header would leave the fields of later versions zero. */
int d2d_fit_opts_default(d2d_fit_opts *o);

/* Build the shared basis block on the host (fp64) and upload it.  wref[3] = weights of the
 * whitening metric sum_d wref[d] Phi_d^T Phi_d.  Synchronous. */
int d2d_fit_plan_create(d2d_ctx *ctx, int S, int K, double duration, const double *wref,
                        d2d_fit_plan **out);
/* The same with the kernel family chosen by the caller (version 108; before: D2D_FIT_KNOT / D2D_FIT_LONG / D2D_FIT_SPLIT /
 * D2D_FIT_LONG_SEG / D2D_FIT_LONG_TABLES in the environment of the process).  popts = NULL: the defaults. */
#define D2D_FIT_KERNEL_AUTO (-1)
typedef struct {
  int32_t kernel;        /* D2D_FIT_KERNEL_AUTO (default): knot for S = 6, K <= 64, long otherwise.  D2D_FIT_KERNEL_FUSED: the q-coordinate
                            fused kernel for the default solver too (S = 6, K <= 64 only); D2D_FIT_KERNEL_LONG: the long-horizon kernel
                            whatever K; D2D_FIT_KERNEL_SPLIT: the launch-pair path (K must fit its LDS image); D2D_FIT_KERNEL_KNOT: as AUTO
                            but D2D_EINVAL when the shape has no knot kernel.  A request the shape cannot serve is D2D_EINVAL.          */
  int32_t long_tables;   /* long-horizon kernel: -1 (default) the segment formulation of csrc/fit_seg.h; 0 / 1 / 2: the table kernels with
                            both tables in global memory / the fp64 block in the LDS / both in the LDS; 3: the table kernels, placement
                            chosen from the LDS footprint (development A/B: tools/bench_long.py, tools/dev_seg.py)                      */
  int32_t reserved[2];   /* zero */
} d2d_fit_plan_opts;
int d2d_fit_plan_create_ex(d2d_ctx *ctx, int S, int K, double duration, const double *wref, const d2d_fit_plan_opts *popts,
                           d2d_fit_plan **out);
int d2d_fit_plan_destroy(d2d_fit_plan *plan);
/* Which kernel d2d_fit_solve runs for this plan (uncoupled): the fused persistent LM kernel (S = 6, K <= 64, everything in
 * LDS), the long-horizon persistent kernel (K > 64: the segment formulation of csrc/fit_seg.h -- Legendre coefficients per segment
 * instead of basis tables, LDS footprint independent of K: any K, e.g. the reference's 101 .. 151-node scenarios and its 50 Hz
 * horizons of 351 .. 1501 nodes; since round 3 also every plan with S != 6: the kernel deals its lanes to any number of segments), or
 * the launch-pair path (d2d_fit_plan_opts.kernel = D2D_FIT_KERNEL_SPLIT).  d2d_fit_eval uses the launch-pair evaluation kernel
 * while the basis block fits the LDS (K <~ 229 at S = 6) and the segment formulation beyond; coupled groups run on the group
 * kernels up to that K and on the long-horizon kernel beyond; d2d_fit_rows / d2d_fit_jtj (the contraction-only pair of the
 * bench) keep the LDS limit (D2D_EINVAL beyond).
 * D2D_FIT_KERNEL_KNOT (round 5): the fused shape (S = 6, K <= 64) with the DEFAULT solver runs in knot coordinates -- the reference's
 * local parameterisation CompositeTraj([MinSnapPoly...]) (src/d2d/trajectory.py:166-208), J^T J block tridiagonal, one MFMA per sample
 * (csrc/fit_knot.hip, oracle/fit_knot.py); D2D_LM_MODE_FAST and the time-sliced hand-out of such a plan stay on the fused q kernel.
 * d2d_fit_plan_opts.kernel = D2D_FIT_KERNEL_FUSED at plan creation keeps the q kernel for the default solver too. */
enum { D2D_FIT_KERNEL_SPLIT = 0, D2D_FIT_KERNEL_FUSED = 1, D2D_FIT_KERNEL_LONG = 2, D2D_FIT_KERNEL_KNOT = 3 };
int d2d_fit_plan_kernel(const d2d_fit_plan *plan);

/* Copy the basis to host buffers (any may be NULL): G [3][K][nq], Gp [3][K][4],
 * Z [8S][nq], Zp [8S][4], Pinit [nq][K]. */
int d2d_fit_plan_get(const d2d_fit_plan *plan, double *G, double *Gp, double *Z, double *Zp,
                     double *Pinit);

/* q0 = projection of the 'tri' waypoints on the basis.  scen dev [B][32]; q dev [B][2*nq]. */
int d2d_fit_init(d2d_ctx *ctx, const d2d_fit_plan *plan, int B, const double *scen, double *q);

/* q0 = least-squares projection of caller-supplied node positions xy dev [B][2][K] (the x and
 * y blocks of an initial-guess vector as Planner.get_initial_guess builds it,
 * src/single_opt_planner.py:79-115) on the basis. */
int d2d_fit_project(d2d_ctx *ctx, const d2d_fit_plan *plan, int B, const double *scen, const double *xy,
                    double *q);

/* One evaluation at q: cost = sum r^2 (fp64), g = J^T r (fp64, dev [B][2nq]), H = J^T J
 * (fp32 via v_mfma_f32_16x16x4_f32, dev [B][2nq][2nq]).  Any output may be NULL. */
int d2d_fit_eval(d2d_ctx *ctx, const d2d_fit_plan *plan, int B, const double *scen,
                 const double *q, double *cost, double *g, float *H);

/* The two halves of d2d_fit_eval as separate launches.  d2d_fit_rows evaluates the residual rows at q -- cost and
 * g = J^T r as d2d_fit_eval (either may be NULL) -- and leaves the fp32 row records (four contracted rows per sample:
 * airspeed, bank and the two position rows) in the plan's scratch, [B][K+1][4] x 16 B in HBM.
 * d2d_fit_jtj contracts the records of the last d2d_fit_rows(B) into J^T J: ONE kernel whose whole duration is the
 * MFMA contraction (records HBM -> LDS, v_mfma_f32_16x16x4_f32, tiles -> HBM) -- the kernel bench.py prices against
 * the fp32 MFMA peak.  H dev [B][2nq][2nq] or NULL (NULL: the tiles stay in the plan's scratch).  Same cost
 * plug-ins as d2d_fit_eval (CostInput, CostObstacle: src/d2d/opty_utils.py:85-134); uncoupled plans only. */
int d2d_fit_rows(d2d_ctx *ctx, d2d_fit_plan *plan, int B, const double *scen, const double *q, double *cost, double *g);
int d2d_fit_jtj(d2d_ctx *ctx, d2d_fit_plan *plan, int B, float *H);

/* Full Levenberg-Marquardt solve from q (in/out).  cost dev [B], iters / status dev int32 [B]
 * (any may be NULL).  stats host double[4] (may be NULL): sum cost, max |J^T r|_inf,
 * trajectories still running, evaluations performed (in units of one Gauss-Newton evaluation = 200
 * contracted rows at K = 50; an evaluation with the second-order blocks contracts 1.5 times as many and counts
 * 1.5) -- the local contribution to the cross-GPU convergence all-reduce.  Synchronises the stream before returning. */
int d2d_fit_solve(d2d_ctx *ctx, const d2d_fit_plan *plan, int B, const double *scen, double *q,
                  const d2d_fit_opts *opts, double *cost, int32_t *iters, int32_t *status,
                  double *stats);

/* The same loop in three parts, for callers that interleave their own convergence check
 * (the cross-GPU all-reduce of the statistics): begin resets the per-trajectory LM state;
 * iterate runs up to n_iters more damped solves (never beyond opts->max_iter in total) and,
 * if n_running != NULL, synchronises and returns how many trajectories are still running (once max_iter is
 * reached it returns 0 without synchronising); finish refreshes cost / J^T r and exports as d2d_fit_solve.
 * Between begin and finish the solve owns q: every d2d_fit_iterate and the d2d_fit_finish of one solve take the SAME q buffer
 * (another pointer is D2D_EINVAL -- the persistent kernels keep the state of the fits that are still running against it, the knot
 * kernel in its own coordinates, so an edited or swapped buffer would be ignored) and options that select the same kernel (a change
 * of mode or slice that would move the solve to another kernel mid-way is D2D_ESTATE). */
int d2d_fit_begin(d2d_ctx *ctx, d2d_fit_plan *plan, int B);
int d2d_fit_iterate(d2d_ctx *ctx, d2d_fit_plan *plan, int B, const double *scen, double *q,
                    const d2d_fit_opts *opts, int n_iters, int32_t *n_running);
int d2d_fit_finish(d2d_ctx *ctx, d2d_fit_plan *plan, int B, const double *scen, const double *q,
                   double *cost, int32_t *iters, int32_t *status, double *stats);

/* Scheduling hint for d2d_fit_solve / d2d_fit_iterate on batches of B trajectories: iters dev int32 [B] = the iteration
 * counts a previous solve of the same scenarios returned (receding-horizon replanning, repeated solves of one batch).  The
 * persistent LM kernel then hands the fits out longest-first, so the launch does not end on one long fit that was drawn late.
 * Results do not depend on it (the fits are independent).  iters = NULL clears the hint; a batch of another size ignores it. */
int d2d_fit_plan_set_order(d2d_ctx *ctx, d2d_fit_plan *plan, int B, const int32_t *iters);
/* The hand-out prior behind D2D_HANDOUT_PREDICTED: how many trial points a fit is EXPECTED to need, as a function of the scenario
 * row alone.  What decides it on the SURVEY 8d workload is how far the two end headings are from the legs of the 'tri' dog-leg the
 * fit starts from (src/d2d/opty_utils.py:171-187) -- near two critical misalignments the start point sits on the watershed between
 * turning left and turning right, lmder stagnates there and the finish walks down a saddle -- and the chord length; the obstacles
 * barely matter (tools/fit_handout_prior.py: rank correlation with the measured trial counts 0.6 on held-out scenarios).  Key of a fit:
 *   table[0][bin(t0)][bin(x)] + table[1][bin(t1)][bin(x)],   x = |p1 - p0| / (vref * duration),
 *   t0 = wrap(psi0 - (beta + a)), t1 = wrap(psi1 - (beta - a)),  beta = heading of the chord, a = go_left * atan(2 h / |p1 - p0|) the
 *   angle of the dog-leg's legs against the chord (h = its apex height, sqrt(max(D^2 - d^2, 0)) / 2),
 * angles in D2D_HANDOUT_NB bins over [-pi, pi), x in D2D_HANDOUT_ND bins over [D2D_HANDOUT_X_LO, D2D_HANDOUT_X_HI) (clamped).  The device
 * computes the keys and a counting sort of them at the start of every solve whose batch exceeds the resident wavefronts (two small
 * launches on the solve's stream, inside whatever the caller times).  table: HOST float [2][D2D_HANDOUT_NB][D2D_HANDOUT_ND], in trial
 * points relative to any common offset; NULL restores the built-in table (csrc/fit_handout_prior.h: regressed on 196 608 synthetic
 * scenarios of OTHER seeds than any bench or test batch).  The built-in table was regressed on the fused shape (S = 6, K <= 64, default
 * solver) and says nothing about other horizons (rank correlation with the trial counts of 121- / 301-node fits: -0.13 / 0.06): plans of
 * the long-horizon kernel hand out in index order until the caller installs a table regressed on solves of their own workload (one
 * solve of a few thousand scenarios is enough: d2dhip/handout.py fit_prior, FitPlan.learn_handout_prior).  A prior only schedules:
 * results are bit-identical whatever it says. */
#define D2D_HANDOUT_NB 48
#define D2D_HANDOUT_ND 12
#define D2D_HANDOUT_X_LO 0.4
#define D2D_HANDOUT_X_HI 1.0
int d2d_fit_plan_set_handout_prior(d2d_ctx *ctx, d2d_fit_plan *plan, const float *table);
/* The hand-out order the LAST solve launch of this plan used over B trajectories: order HOST int32 [B] (position i took trajectory
 * order[i]); D2D_ESTATE if that launch ran in index order or over another batch size.  Synchronous (tests, diagnostics). */
int d2d_fit_plan_get_order(d2d_ctx *ctx, d2d_fit_plan *plan, int B, int32_t *order);
/* The same hint for d2d_fit_solve_groups over R scenarios: from_last != 0 orders the next solves by the sweep counts the
 * LAST d2d_fit_solve_groups of this plan (same R) recorded, longest first; 0 clears it.  Results do not depend on it (the
 * scenarios are independent).  D2D_ESTATE if the plan holds no sweep counts for R scenarios. */
int d2d_fit_plan_set_group_order(d2d_ctx *ctx, d2d_fit_plan *plan, int R, int from_last);

/* Coupled groups (BASELINE configs[2], multi_opt_planner): trajectories g*n_ac .. g*n_ac+n_ac-1 are
 * the aircraft of one scenario and repel each other through CostCollision rows
 * (src/d2d/multiopty_utils.py:120-153; D2D_SC_KCOL/RCOL/SCOL/PMASK select weight, radius, scale and
 * partners).  d2d_fit_plan_set_groups(n_ac <= 8) sizes the kernels' LDS for the extra rows;
 * d2d_fit_solve_groups runs block Gauss-Seidel over the aircraft index (all R groups in one batch
 * per visit, the others' sampled positions frozen, LM state restarted, at most inner_iters damped
 * solves per visit) until no unknown moved by more than tol*(1+|q|) in a sweep or max_sweeps.
 * scen dev [R*n_ac][32], q dev [R*n_ac][2nq] in/out, cost dev [R*n_ac] (per-aircraft sub-problem cost:
 * own rows + its collision rows) or NULL, sweeps_done host or NULL, stats host [4] or NULL
 * (sum of sub-problem costs, max |J^T r|, last sweep's largest relative move, evaluations).
 * Synchronises the stream. */
int d2d_fit_plan_set_groups(d2d_fit_plan *plan, int n_ac);
int d2d_fit_solve_groups(d2d_ctx *ctx, d2d_fit_plan *plan, int R, const double *scen, double *q,
                         const d2d_fit_opts *opts, int max_sweeps, int inner_iters, double tol,
                         double *cost, int32_t *sweeps_done, double *stats);
/* Per-scenario report of the last d2d_fit_solve_groups of this plan (persistent-kernel path): sweeps HOST int32 [R] = sweeps each
 * scenario used, moved HOST double [R] = its largest relative move in the last of them (<= tol: settled); either may be NULL.
 * D2D_ESTATE if the plan holds no such solve over R scenarios.  Synchronous. */
int d2d_fit_group_report(d2d_ctx *ctx, d2d_fit_plan *plan, int R, int32_t *sweeps, double *moved);

/* Per-launch timing of the LM loop with HIP events on the context's stream: enable = 1
 * starts a fresh recording, 0 stops (d2d_fit_eval's kernel launch is recorded too, as a fit_eval launch).
 * d2d_fit_profile_read waits for the recorded events:
 * out[8]: [0] = sum of fit_eval (J^T J) kernel ms, [1] = its launches, [2] = sum of fit_step kernel
 * ms, [3] = its launches, [4] = sum of fit_lm (fused persistent LM loop) kernel ms, [5] = its launches,
 * [6] = sum of fit_jtj (contraction-only, d2d_fit_jtj) kernel ms, [7] = its launches. */
int d2d_fit_profile(d2d_fit_plan *plan, int enable);
int d2d_fit_profile_read(d2d_fit_plan *plan, double *out);

/* Map q back to monomial coefficients z dev [B][2][S][8] (axis, segment, power). */
int d2d_fit_coeffs(d2d_ctx *ctx, const d2d_fit_plan *plan, int B, const double *scen,
                   const double *q, double *z);

/* Sample a coefficient set: flat outputs and states at the K plan nodes.
 * Y dev [B][6][K] (x,y,xd,yd,xdd,ydd) and Xs dev [B][5][K] (x,y,psi,phi,v); may be NULL. */
int d2d_fit_sample(d2d_ctx *ctx, const d2d_fit_plan *plan, int B, const double *scen,
                   const double *q, double *Y, double *Xs);

/* ------------------------------------------------------------------------------------
 * Direct-collocation NLP in the reference's own parameterisation: the solve behind
 * `opty.direct_collocation.Problem(obj, obj_grad, eom, state_symbols, num_nodes, time_step, known_parameter_map,
 * instance_constraints, bounds)` + `.solve(x0)` (src/single_opt_planner.py:62-71,124; src/multi_opt_planner.py:69-78,86).
 *   unknowns     node values (x, y, psi, phi, v)(t_i), i = 0..N-1 (the reference's free vector, src/single_opt_planner.py:35-39)
 *   equalities   backward-Euler collocation of the symbolic model (src/d2d/opty_utils.py:38-50, +wind sign quirk), i = 1..N-1:
 *                (x_i - x_{i-1})/h - v_i cos psi_i + wx = 0, (y_i - y_{i-1})/h - v_i sin psi_i + wy = 0,
 *                (psi_i - psi_{i-1})/h - g/v_i tan phi_i = 0;  end conditions (x, y, psi)(t0) = p0, (t1) = p1 (:46-49)
 *   bounds       HARD boxes on phi, v and optionally x, y (:53-57)
 *   objective    the cost plug-ins (CostInput / CostAirVel / CostBank mean / CostObstacle(s) / CostComposit,
 *                src/d2d/opty_utils.py:55-165; one CostCollision partner, src/d2d/multiopty_utils.py:120-153) with the gradient
 *                the reference hands to IPOPT (cost_grad, incl. the missing (k/r)^2 of CostObstacle kind 1: the solver therefore
 *                minimises the objective whose gradient that is; the cost REPORTED is the reference's cost()).
 * Solver (oracle/nlp.py is its CPU statement): equalities by an augmented Lagrangian, bounds by a primal-dual log barrier,
 * damped Newton steps on the block-tridiagonal (5x5 blocks) Lagrangian Hessian.  One problem per wavefront: merit function,
 * assembly, the elimination of (phi, v), ratio tests and updates run with lane = node; the 3x3 block recursion that remains is
 * serial in the nodes and runs from both ends of the horizon towards the middle in the two halves of the wave.
 * scen dev [B][D2D_SCEN_STRIDE]: the fit's scenario rows (end poses, VSP, KV, KPHI, KOBS, S, wind, obstacles + OKIND, PHIMAX,
 * VMIN/VMAX, the x/y box, KCOL/RCOL/SCOL); W dev [B][5][N] node values (problem, component, node), in: the initial guess (e.g.
 * Planner.get_initial_guess), out: the solution; partner dev [B][2][N] frozen positions of the CostCollision partner or NULL;
 * work dev double[d2d_nlp_workspace_doubles(N) * B] (scratch: the launch is persistent and uses one workspace per resident wavefront, the
 * first min(B, wave slots) of them; nothing in it is an output); mult dev [B][3][N] or NULL: out, scaled multiplier estimates (node 0 unused;
 * Lagrange multiplier = 2 rho mu);  cost dev [B] (the reference's cost() at the solution), feas dev [B] (largest collocation
 * residual, in the reference's form), iters / status dev int32 [B] or NULL (Newton steps; D2D_ST_CONVERGED / D2D_ST_MAXITER /
 * D2D_ST_STALLED: no feasible point found -- the violation stopped shrinking at the largest penalty).
 * Asynchronous on the context's stream. */
#define D2D_NLP_RHO0 10.0
#define D2D_NLP_RHO_MAX 1e8
#define D2D_NLP_RHO_GROW 10.0
#define D2D_NLP_MUB0 0.1
#define D2D_NLP_MUB_MIN 1e-9
#define D2D_NLP_GRAD_FLOOR 1e-11   /* x rho: rounding floor of the penalty gradient */
#define D2D_NLP_GATE_PROGRESS 1e-9  /* relative decrease of the merit function over inner_max steps below which an unsolved inner problem is left */
#define D2D_NLP_BANKMAX_VALUE_TOL 1e-7   /* CostBank max mode (D2D_SC_BANKMAX rows): converged in value -- feasible, barrier at its floor, a batch of
                                           inner_max steps lowered the merit function by no more than this fraction (the one-hot cost_grad has no zero) */
#define D2D_NLP_BANKMAX_BATCHES 3    /* ... over a window of this many batches of inner_max steps (a D2D_SC_BANKMAX row runs them as one batch) */
#define D2D_NLP_STALL_OUTERS 5      /* solved inner problems in a row that did not halve the violation: D2D_ST_STALLED */
/* Newton steps between two looks at the schedule (multipliers, penalty, barrier parameter) and the number of such looks.  A batch
 * of steps that leaves its inner problem unsolved is followed by another one under the same parameters only while it still lowers
 * the merit function (D2D_NLP_GATE_PROGRESS): with batches of 20 the rare problems whose inner iteration crawls are moved on 3 x
 * sooner than with the 60 of rounds 2-3 (4096 perturbed exp_14: longest problem 481 -> 260 steps, mean 152 unchanged, the same
 * problems converge to the same costs within 5e-10, the reference's 32 catalogue cases keep status and cost) -- and the
 * longest problem is what a batch launch waits for. */
#define D2D_NLP_INNER_MAX 20
#define D2D_NLP_OUTER_MAX 120
typedef struct {
  double rho0;       /* initial penalty (D2D_NLP_RHO0)                                                        */
  double mub0;       /* initial barrier parameter (D2D_NLP_MUB0)                                              */
  double mub_min;    /* final barrier parameter (D2D_NLP_MUB_MIN)                                             */
  double feas_tol;   /* largest collocation residual at convergence (default 1e-9; IPOPT's runs: tol 1e-5)    */
  double opt_tol;    /* barrier KKT error of the last inner problem (default 1e-7)                            */
  int32_t inner_max; /* Newton steps per outer iteration (D2D_NLP_INNER_MAX)                                    */
  int32_t outer_max; /* outer iterations: multiplier / barrier updates (D2D_NLP_OUTER_MAX)                     */
  int32_t serial;    /* solver of the reduced block-tridiagonal system of a Newton step: 0 (default) block cyclic reduction -- log2 N
                        levels of independent 3x3 eliminations, one lane per node; 1 the twisted serial block recursion of round 2
                        (N/2 dependent block pivots).  Same step to rounding.                                   */
  int32_t slots;     /* resident wavefronts (= workspaces) of the persistent launch; 0 (default) = every wave slot of the device
                        (version 108; before: D2D_NLP_SLOTS in the environment.  Was `reserved`, always 0.)                 */
  const double *bounds; /* dev [B][4] = (phi_lo, phi_hi, psi_lo, psi_hi) per problem, or NULL.  phi_lo < phi_hi replaces the symmetric
                           |phi| <= D2D_SC_PHIMAX of the scenario row (opty's bounds dict may hold any interval,
                           src/single_opt_planner.py:53); psi_lo < psi_hi adds a box on the heading of the free nodes (none
                           otherwise; the end headings are equalities).                                             */
  const int32_t *order; /* dev [B]: a permutation of 0 .. B-1, the order in which the persistent launch hands the problems to its
                           resident wavefronts (longest first is what shortens a launch that holds a few problems per wavefront:
                           e.g. the argsort of the previous solve's `iters`, descending), or NULL: index order.  Scheduling only:
                           every problem's result is the same whatever the order.  An entry outside 0 .. B-1 is skipped (the
                           problem it should have named is not solved); an index named twice is the caller's error (two
                           wavefronts would write the same W).  4096 perturbed exp_14: 168 k -> 205 k problems/s.  (version 109) */
} d2d_nlp_opts;
int d2d_nlp_workspace_doubles(int N);
/* A problem whose row is unusable -- PHIMAX <= 0, VMIN <= 0 or VMIN >= VMAX (the model divides by v and the barrier needs an
 * interior), an inverted position box, non-finite end poses -- is refused at once: status D2D_ST_NONFINITE, cost = feas = NaN. */
int d2d_nlp_solve(d2d_ctx *ctx, int B, int N, double h, const double *scen, const d2d_nlp_opts *opts, double *W,
                  const double *partner, double *work, double *mult, double *cost, double *feas, int32_t *iters,
                  int32_t *status);

/* The reference's multi-aircraft Problem (src/multi_opt_planner.py:41-78,86: one NLP over the n_ac aircraft of a scenario) for R
 * scenarios in ONE launch: problems r*n_ac .. r*n_ac + n_ac - 1 (rows of scen, W, cost, ...) are the aircraft of scenario r.  The
 * aircraft are coupled through the objective only -- CostCollision on the pair (0, 1), src/d2d/multiopty_utils.py:120-153
 * (D2D_SC_KCOL / RCOL / SCOL of rows 0 and 1; KCOL = 0 on the scenario's first row: uncoupled) -- so a fixed point of block
 * Gauss-Seidel over the aircraft, each block the full collocation solve of one aircraft against the partner's frozen node
 * positions, is a KKT point of the joint problem.  A workgroup takes a scenario, wavefront a its aircraft a: every aircraft is solved
 * uncoupled first (concurrently), then aircraft 0 and 1 take turns until neither moved by more than tol (metres) in a sweep or
 * max_sweeps; no host round trips.  A pair that has not settled reports D2D_ST_MAXITER.
 * work dev double[(d2d_nlp_workspace_doubles(N) * n_ac + 2 * N) * R]; sweeps dev int32 [R], moved dev [R] (largest move of the last sweep)
 * or NULL; the other arguments as d2d_nlp_solve with B = R * n_ac.  Asynchronous on the context's stream. */
int d2d_nlp_solve_groups(d2d_ctx *ctx, int R, int n_ac, int N, double h, const double *scen, const d2d_nlp_opts *opts, int max_sweeps,
                         double tol, double *W, double *work, double *mult, double *cost, double *feas, int32_t *iters, int32_t *status,
                         int32_t *sweeps, double *moved);

#ifdef __cplusplus
}
#endif
#endif /* D2D_H */
