/*
 * uvs_solver.h -- C ABI of the MI355X-native sliding-window back-end.
 *
 * This header is the drop-in boundary for ONE path of url-kaist/UV-SLAM: the
 * nonlinear least-squares solve inside Estimator::optimization()
 * (reference vins_estimator/src/estimator.cpp:761-1233).  In the reference
 * that function builds a ceres::Problem and calls ceres::Solve; a maintainer
 * replaces that body by "fill a uvs_window, call uvs_solve_window()" -- see
 * INTEGRATION.md for the exact stub.
 *
 * Conventions (all taken from the reference, file:line cited per field):
 *   - every scalar is IEEE double (the reference path is FP64 end to end);
 *   - pose block = (px,py,pz, qx,qy,qz,qw)              estimator.cpp:530-537
 *   - speed/bias block = (v[3], ba[3], bg[3])            estimator.cpp:539-549
 *   - residual row order of an IMU block = (P,R,V,BA,BG) parameters.h:59-66
 *   - parameter blocks are addressed by INDEX (frame id, landmark id), not by
 *     pointer as in Ceres (SURVEY.md section 8b "Parameter memory").
 *
 * Plain pointers and sizes only; no C++ / torch types cross this boundary.
 * All pointers are HOST pointers unless a function says otherwise.
 */
#ifndef UVS_SOLVER_H
#define UVS_SOLVER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UVS_ABI_VERSION 7

#define UVS_WINDOW_SIZE 10                    /* parameters.h:12 WINDOW_SIZE  */
#define UVS_NUM_FRAMES (UVS_WINDOW_SIZE + 1)  /* frames 0..WINDOW_SIZE        */
#define UVS_SIZE_POSE 7                       /* parameters.h:51 SIZE_POSE    */
#define UVS_SIZE_SPEEDBIAS 9                  /* parameters.h:52              */
#define UVS_SIZE_LINE 4                       /* parameters.h:54              */
#define UVS_MAX_ITER 64                       /* capacity of the per-iteration trace */
#define UVS_MAX_PRIOR_BLOCKS 16               /* 10 poses + speedbias + ex + td (+slack) */
#define UVS_MAX_PRIOR_DIM 96                  /* n <= 76 in the reference (a9) */

/* ---- status codes (the reference has no error path: estimator.cpp:993-997
 *      discards ceres::Solver::Summary; we return one and never abort) ---- */
enum {
    UVS_OK = 0,
    UVS_ERR_INVALID_ARG = 1,    /* null pointer / index out of range / bad count */
    UVS_ERR_UNSUPPORTED = 2,    /* a combination this path does not take: relocalization blocks in a landmark-sharded solve over SEVERAL ranks;
                                 * RCCL not found for a multi-rank communicator */
    UVS_ERR_NO_DEVICE = 3,      /* no HIP device / extension cannot run (never falls back to CPU) */
    UVS_ERR_HIP = 4,            /* a HIP runtime call failed; see uvs_last_error() */
    UVS_ERR_CAPACITY = 5,       /* window larger than the handle was created for */
    UVS_ERR_NUMERIC = 6         /* non-finite cost / Cholesky breakdown reported by the device */
};

/* ---- termination reasons, named after Ceres ---- */
enum {
    UVS_TERM_NO_CONVERGENCE = 0,       /* max_num_iterations reached           */
    UVS_TERM_GRADIENT_TOL = 1,
    UVS_TERM_PARAMETER_TOL = 2,
    UVS_TERM_FUNCTION_TOL = 3,
    UVS_TERM_MIN_RADIUS = 4,
    UVS_TERM_INVALID_STEPS = 5,        /* max_num_consecutive_invalid_steps    */
    UVS_TERM_NUMERIC_FAILURE = 6,
    UVS_TERM_MAX_TIME = 7              /* max_solver_time_in_seconds reached (Ceres: NO_CONVERGENCE, "Maximum solver time reached") */
};

/* Solver options == the globals Estimator::optimization() reads
 * (parameters.h:11-47, estimator.cpp:982-991) plus the Ceres defaults it
 * relies on.  uvs_default_options() fills the EuRoC
 * values of config/euroc/euroc_config.yaml. */
typedef struct uvs_options {
    int32_t max_num_iterations;        /* NUM_ITERATIONS, euroc_config.yaml:56 (10)        */
    int32_t estimate_extrinsic;        /* ESTIMATE_EXTRINSIC (0): Ex_Pose constant; != 0: free 6-dof block (estimator.cpp:784-788) */
    int32_t estimate_td;               /* ESTIMATE_TD (0); 1: ProjectionTdFactor + para_Td (estimator.cpp:790-797,853-858) */
    int32_t function_tol_keeps_candidate; /* 0 = Ceres order: tolerance checks before accept (App. B.4) */
    double focal_length;               /* FOCAL_LENGTH = fx, parameters.cpp:60 (461.6)     */
    double point_sqrt_info;            /* FOCAL_LENGTH/1.6, estimator.cpp:17               */
    double line_factor;                /* LINE_FACTOR (300)  euroc_config.yaml:86          */
    double vp_factor;                  /* VP_FACTOR (10)     euroc_config.yaml:87          */
    double loss_point;                 /* CauchyLoss(1.0)  estimator.cpp:765               */
    double loss_line;                  /* CauchyLoss(0.1)  estimator.cpp:768               */
    double loss_vp;                    /* CauchyLoss(1.0)  estimator.cpp:772               */
    double gravity[3];                 /* G = (0,0,g_norm) parameters.cpp:12,79            */
    /* Ceres trust-region defaults (not set by the reference => defaults apply) */
    double initial_trust_region_radius;   /* 1e4  */
    double max_trust_region_radius;       /* 1e16 */
    double min_trust_region_radius;       /* 1e-32 */
    double min_relative_decrease;         /* 1e-3 */
    double min_lm_diagonal;               /* 1e-6 */
    double max_lm_diagonal;               /* 1e32 */
    double function_tolerance;            /* 1e-6 */
    double gradient_tolerance;            /* 1e-10 */
    double parameter_tolerance;           /* 1e-8 */
    int32_t max_consecutive_invalid_steps;/* 5 */
    int32_t jacobi_scaling;               /* 1 */
    double max_solver_time_in_seconds;    /* options.max_solver_time_in_seconds = SOLVER_TIME or 0.8 SOLVER_TIME (estimator.cpp:987-991); checked at the top of
                                           * every LM iteration with the GPU's 100 MHz wall clock.  0 (default) = no cap: the iteration count is then
                                           * deterministic, which the parity runs need */
} uvs_options;

/* One IMU pre-integration block == the fields of IntegrationBase that
 * IMUFactor::Evaluate reads (integration_base.h:188-203, imu_factor.h:19-182).
 * Links frame i = index, frame j = index+1 (estimator.cpp:811-818).
 * jacobian / covariance are 15x15 ROW-major here (Eigen's are column-major;
 * the host shim transposes on copy). */
typedef struct uvs_imu_block {
    double sum_dt;
    double delta_p[3];
    double delta_q[4];                 /* (x,y,z,w) */
    double delta_v[3];
    double linearized_ba[3];
    double linearized_bg[3];
    double jacobian[15 * 15];
    double covariance[15 * 15];
    int32_t frame_i;                   /* j = frame_i + 1 */
    int32_t skip;                      /* 1 when sum_dt > 10.0 (estimator.cpp:814) */
} uvs_imu_block;

/* Marginalization prior == what MarginalizationFactor::Evaluate reads from
 * MarginalizationInfo (marginalization_factor.cpp:333-381, .h:64-69).
 * n rows; kept block b has global size block_size[b] (7,9,1), column offset
 * block_idx[b] (already minus m), linearization point x0 at x0[x0_off[b]..].
 * linearized_jacobians is n x n ROW-major. */
enum { UVS_BLOCK_POSE = 0, UVS_BLOCK_SPEEDBIAS = 1, UVS_BLOCK_EX_POSE = 2, UVS_BLOCK_TD = 3 };
#define UVS_PRIOR_X0_LEN (UVS_MAX_PRIOR_BLOCKS * 9)
typedef struct uvs_prior {
    int32_t n;                         /* 0 => no prior (last_marginalization_info == nullptr) */
    int32_t n_blocks;
    int32_t block_kind[UVS_MAX_PRIOR_BLOCKS];   /* UVS_BLOCK_*                                */
    int32_t block_frame[UVS_MAX_PRIOR_BLOCKS];  /* frame index for POSE / SPEEDBIAS, else 0   */
    int32_t block_size[UVS_MAX_PRIOR_BLOCKS];   /* keep_block_size (global size 7/9/1)        */
    int32_t block_idx[UVS_MAX_PRIOR_BLOCKS];    /* keep_block_idx - m (local column offset)   */
    int32_t x0_off[UVS_MAX_PRIOR_BLOCKS];       /* offset of keep_block_data in x0[]          */
    double x0[UVS_PRIOR_X0_LEN];
    double linearized_residuals[UVS_MAX_PRIOR_DIM];
    double linearized_jacobians[UVS_MAX_PRIOR_DIM * UVS_MAX_PRIOR_DIM];  /* row-major, leading dim n */
} uvs_prior;

/* The sliding window handed to the solver: exactly what optimization() feeds
 * Ceres after vector2double() (estimator.cpp:800), flattened to SoA.
 *
 * Point residual blocks (estimator.cpp:823-866): one entry per
 * ProjectionFactor(pts_i, pts_j) with blocks Pose[fi], Pose[fj], Ex_Pose,
 * Feature[lm].  Entries of one landmark must be contiguous and lm must be
 * non-decreasing (this is the order the reference loop emits them in).
 *
 * Line residual blocks (estimator.cpp:868-927): one entry per
 * LineProjectionFactor(ric,tic,sp,ep) with blocks Pose[fj], Ortho[lm]; when
 * has_vp != 0 the same entry also carries VPProjectionFactor(...,vp)
 * (estimator.cpp:920-925, added iff vp(2)==1).  Same contiguity rule. */
typedef struct uvs_window {
    /* frame states: para_Pose / para_SpeedBias / para_Ex_Pose (estimator.h:114-121) */
    double pose[UVS_NUM_FRAMES][UVS_SIZE_POSE];
    double speedbias[UVS_NUM_FRAMES][UVS_SIZE_SPEEDBIAS];
    double ex_pose[UVS_SIZE_POSE];
    double td;                         /* para_Td (unused unless estimate_td) */

    /* point landmarks: para_Feature[l][0] = inverse depth (feature_manager.cpp:290-306) */
    int32_t n_points;
    int32_t n_point_obs;
    const double *inv_depth;           /* [n_points]            */
    const int32_t *pt_lm;              /* [n_point_obs] feature_index            */
    const int32_t *pt_fi;              /* [n_point_obs] imu_i (anchor frame)     */
    const int32_t *pt_fj;              /* [n_point_obs] imu_j                    */
    const double *pt_pi;               /* [n_point_obs][3] pts_i                 */
    const double *pt_pj;               /* [n_point_obs][3] pts_j                 */

    /* line landmarks: para_Ortho_plucker[l] = (psi_x,psi_y,psi_z,phi) (feature_manager.cpp:308-331) */
    int32_t n_lines;
    int32_t n_line_obs;
    const double *line_orth;           /* [n_lines][4]          */
    const int32_t *ln_lm;              /* [n_line_obs] line_feature_index        */
    const int32_t *ln_fj;              /* [n_line_obs] imu_j                     */
    const double *ln_sp;               /* [n_line_obs][3] start_point            */
    const double *ln_ep;               /* [n_line_obs][3] end_point              */
    const int32_t *ln_has_vp;          /* [n_line_obs] 1 iff vp(2)==1            */
    const double *ln_vp;               /* [n_line_obs][3] vp                     */

    /* IMU factors (estimator.cpp:811-818): up to WINDOW_SIZE blocks */
    int32_t n_imu;
    const uvs_imu_block *imu;          /* [n_imu] */

    /* marginalization prior (estimator.cpp:803-809); may be NULL / n==0 */
    const uvs_prior *prior;

    /* ProjectionTdFactor inputs (projection_td_factor.cpp:3-16,51-52), read only when options.estimate_td != 0, else may be NULL:
     * image-plane feature velocities of the anchor / current observation and their time offsets.  The rolling-shutter term is
     * folded in by the caller: pt_td_i = cur_td_i - TR / ROW * (row_i - ROW / 2), so that
     *   pts_i_td = pts_i - (td - pt_td_i) * (vel_i, 0)          (same for j). */
    const double *pt_vel_i;            /* [n_point_obs][2] */
    const double *pt_vel_j;            /* [n_point_obs][2] */
    const double *pt_td_i;             /* [n_point_obs]    */
    const double *pt_td_j;             /* [n_point_obs]    */

    /* Relocalization residual blocks (estimator.cpp:944-978), n_relo_obs == 0 when relocalization_info == 0.  One entry per matched
     * feature: ProjectionFactor(pts_i, pts_j) on the blocks Pose[start_frame], relo_Pose, Ex_Pose, Feature[lm] with pts_i = the
     * feature's FIRST observation (feature_per_frame[0].point) and pts_j = (match_point.x, match_point.y, 1).  relo_Pose is a free
     * 7-dof block with PoseLocalParameterization (estimator.cpp:947-948); it starts at para_Pose[relo_frame_local_index]
     * (Estimator::setReloFrame, estimator.cpp:1361-1379).  relo_lm must be strictly increasing and every such landmark needs at least
     * one ordinary observation (its anchor frame imu_i is taken from there).  With estimate_td the blocks stay plain ProjectionFactors (no
     * dependence on td, estimator.cpp:967-970).  With estimate_extrinsic the 6 + 6 (+ 1) free dofs beside the frames no longer fit the spare rows of
     * the reduced system: relo_Pose is then eliminated at a second level (same exact solve of the damped system) -- by uvs_solve_window, the batch
     * entry points and, on one rank, the uvs_large_* forms. */
    int32_t n_relo_obs;
    double relo_pose[UVS_SIZE_POSE];
    const int32_t *relo_lm;            /* [n_relo_obs] feature_index */
    const double *relo_pi;             /* [n_relo_obs][3] pts_i      */
    const double *relo_pj;             /* [n_relo_obs][3] pts_j      */
} uvs_window;

/* Solver output == the para_* arrays after ceres::Solve and BEFORE
 * double2vector() (estimator.cpp:999); caller allocates inv_depth[n_points]
 * and line_orth[n_lines*4]. */
typedef struct uvs_state {
    double pose[UVS_NUM_FRAMES][UVS_SIZE_POSE];
    double speedbias[UVS_NUM_FRAMES][UVS_SIZE_SPEEDBIAS];
    double ex_pose[UVS_SIZE_POSE];
    double td;
    double *inv_depth;                 /* [n_points]   */
    double *line_orth;                 /* [n_lines][4] */
    double relo_pose[UVS_SIZE_POSE];   /* relo_Pose after the solve (estimator.cpp:671-685 reads it); the input value when n_relo_obs == 0 */
} uvs_state;

/* Replaces ceres::Solver::Summary (discarded by the reference) with the trace
 * needed for parity diffing. Entry 0 of the arrays is
 * the initial evaluation; entry k>=1 is LM iteration k. */
typedef struct uvs_report {
    int32_t status;                    /* UVS_OK / UVS_ERR_NUMERIC */
    int32_t termination;               /* UVS_TERM_* */
    int32_t num_iterations;            /* LM iterations executed (successful or not) */
    int32_t num_successful;
    double initial_cost;
    double final_cost;
    double cost[UVS_MAX_ITER + 1];             /* cost of x after iteration k             */
    double candidate_cost[UVS_MAX_ITER + 1];
    double model_cost_change[UVS_MAX_ITER + 1];
    double relative_decrease[UVS_MAX_ITER + 1];
    double radius[UVS_MAX_ITER + 1];           /* trust region radius after iteration k   */
    double step_norm[UVS_MAX_ITER + 1];
    double gradient_max_norm[UVS_MAX_ITER + 1];
    int32_t accepted[UVS_MAX_ITER + 1];        /* 1 accepted, 0 rejected, -1 invalid step */
} uvs_report;

/* Per-residual-block evaluation dump (uvs_evaluate): what
 * cost_function->Evaluate + the Ceres corrector would hand the linear solver,
 * in LOCAL (tangent) column size.  Used by the parity tests to compare the
 * hand-derived HIP Jacobians element-wise with the oracle's Jets.
 *   point block : r[2], J = [d/dPose_i (2x6) | d/dPose_j (2x6) | d/dEx (2x6) | d/dlambda (2x1)]  -> 2x19 row-major
 *   line block  : r[2], J = [d/dPose_j (2x6) | d/dline (2x4)]                                  -> 2x10
 *   vp block    : r[1], J = [d/dPose_j (1x6) | d/dline (1x4)]                                  -> 1x10 (zeros when !has_vp)
 *   imu block   : r[15], J = [Pose_i (15x6) | SB_i (15x9) | Pose_j (15x6) | SB_j (15x9)]         -> 15x30
 *   prior       : r[n]  (J is the constant linearized_jacobians)
 * robust!=0 applies the Cauchy corrector (a10); cost = sum over blocks of 0.5*rho(s). */
typedef struct uvs_eval {
    double *pt_r;      /* [n_point_obs][2]   */
    double *pt_J;      /* [n_point_obs][2*19]*/
    double *ln_r;      /* [n_line_obs][2]    */
    double *ln_J;      /* [n_line_obs][2*10] */
    double *vp_r;      /* [n_line_obs][1]    */
    double *vp_J;      /* [n_line_obs][10]   */
    double *imu_r;     /* [n_imu][15]        */
    double *imu_J;     /* [n_imu][15*30]     */
    double *prior_r;   /* [prior n]          */
    double cost;       /* out */
    double *pt_Jtd;    /* [n_point_obs][2] d r / d td of ProjectionTdFactor (only with estimate_td; may be NULL) */
} uvs_eval;

typedef struct uvs_solver uvs_solver;  /* opaque: device buffers, stream, workspaces */

/* ---- lifecycle ---- */
int uvs_abi_version(void);
void uvs_default_options(uvs_options *opts);
/* device = HIP device ordinal. max_* size the per-window workspaces
 * (reference capacities NUM_OF_F = NUM_OF_LF = 1000, parameters.h:14-16).
 * Fails with UVS_ERR_NO_DEVICE when no GPU is present: there is no CPU path. */
int uvs_create(const uvs_options *opts, int device, int max_batch, int max_points, int max_point_obs,
               int max_lines, int max_line_obs, uvs_solver **out);
void uvs_destroy(uvs_solver *s);
const char *uvs_last_error(const uvs_solver *s);
const char *uvs_status_string(int status);

/* ---- the hot path: replaces problem build + ceres::Solve (estimator.cpp:763-997) ---- */
int uvs_solve_window(uvs_solver *s, const uvs_window *w, uvs_state *out, uvs_report *rep);

/* Batch of independent windows (BASELINE configs[2]). upload: host->HBM once;
 * solve: device-resident, re-runnable (reads the uploaded initial state, writes
 * separate outputs); elapsed_ms (may be NULL) = HIP-event time of the solve
 * kernels on the solver's stream; download: HBM->host. */
int uvs_batch_upload(uvs_solver *s, int n, const uvs_window *const *ws);
int uvs_batch_solve(uvs_solver *s, float *elapsed_ms);
int uvs_batch_download(uvs_solver *s, int n, uvs_state *states, uvs_report *reps);
/* A stream of n_batches batches of per_batch windows each (ws[k * per_batch + b] = window b of batch k), END TO END: host packing, upload, solve
 * and download of consecutive batches overlap on three buffer sets (no reference equivalent: offline replay of recorded windows, estimator.cpp:992
 * once per window).  states / reps: n_batches * per_batch entries (either may be NULL); wall_ms: wall time of the whole call.  Results are those
 * of uvs_batch_upload / uvs_batch_solve / uvs_batch_download batch by batch.  On an error in batch k (e.g. a malformed window) the batches before k have been
 * delivered, batch k and the later ones have not been touched, and the code of the first error is returned.  The call leaves NO resident batch behind:
 * uvs_batch_solve / uvs_batch_download after it need their own uvs_batch_upload. */
int uvs_batch_stream(uvs_solver *s, int n_batches, int per_batch, const uvs_window *const *ws, uvs_state *states, uvs_report *reps, double *wall_ms);

/* One evaluation of every residual block at the window's state (no solve).  Relocalization blocks (n_relo_obs) are solve-only: they are
 * neither evaluated here nor marginalized below (the reference's marginalization does not add them, estimator.cpp:1002-1228), and the
 * per-observation outputs keep the caller's numbering. */
int uvs_evaluate(uvs_solver *s, const uvs_window *w, int robust, uvs_eval *out);

/* Diagnostic (parity tests only): reduced system of the FIRST LM iteration of `w`:
 * S_lower[176*176] row-major = damped, landmark-Schur-reduced frame system in the padded index space
 * (16*frame + dof, dof 15 = dummy pivot), g/hd/dd/step[176], scal[UVS_DEBUG_SCAL_LEN] = {cost, gmax, chol_ok, model_cost_change,
 * step_norm^2, -, -, -, per-phase shader cycles[16], sub-timers[8], -...}: the caller's buffer must hold UVS_DEBUG_SCAL_LEN doubles. */
#define UVS_DEBUG_SCAL_LEN 40
int uvs_debug_first_iteration(uvs_solver *s, const uvs_window *w, double *S_lower, double *g, double *hd, double *dd,
                              double *step, double *scal);

/* Diagnostic, host only (no device is touched): packs `w` the way uvs_batch_upload() does and reports the layout:
 * info[12] = {blob bytes, workspace doubles, landmark chunks, packed point observations (incl. relocalization blocks), relocalization
 * blocks, doubles per point record, extra Schur slots per point landmark, LDS doubles of the fullest chunk, LDS staging capacity,
 * largest split of a pose block, compact prior-image entries, pose blocks the prior touches}.  Same status codes as the upload. */
int uvs_debug_pack_layout(const uvs_options *opts, const uvs_window *w, int32_t *info);

/* ---- marginalization (estimator.cpp:1002-1228, marginalization_factor.cpp) ----
 * flag 0 = MARGIN_OLD, 1 = MARGIN_SECOND_NEW. `w` carries the POST-solve state
 * (the reference calls vector2double() again at :1004). Output prior is already
 * re-indexed for the next window (addr_shift, estimator.cpp:1139-1153). */
int uvs_marginalize(uvs_solver *s, const uvs_window *w, int flag, uvs_prior *out);
/* Same, for the usual sequence "solve window w, then marginalize it" (Estimator::optimization()): the caller promises that the
 * residual blocks of `w` are the ones of the LAST upload of this handle (uvs_solve_window / uvs_batch_upload with n = 1) and that
 * only the state fields (pose, speedbias, ex_pose, td, inv_depth, line_orth) changed; the resident factors are reused and only the
 * state is sent to the device.  Returns UVS_ERR_INVALID_ARG when the block counts do not match the resident window. */
int uvs_marginalize_resident(uvs_solver *s, const uvs_window *w, int flag, uvs_prior *out);
/* The same call in two halves (round 4): uvs_marginalize_resident_begin() hands the whole marginalization -- sub-window packing, the device linearization, the
 * elimination of the departing frame and the n x n factorization -- to a worker thread of the handle and returns at once; uvs_marginalize_wait() blocks until the
 * prior is there and returns what uvs_marginalize_resident() would have returned.  The prior is first needed by the NEXT uvs_solve_window(), so the caller's own
 * work between two frames (window slide, IMU integration, feature bookkeeping: estimator.cpp:123-222) runs beside it.  Contract: between begin and wait the handle
 * must not be used for anything else, and `w` as well as every array it points to (incl. w->prior) must stay valid and unchanged.  uvs_destroy() waits by itself. */
int uvs_marginalize_resident_begin(uvs_solver *s, const uvs_window *w, int flag);
int uvs_marginalize_wait(uvs_solver *s, uvs_prior *out);
/* The marginalization of a BATCH of independent windows (ABI v7, round 6): out[b] = what uvs_marginalize(s, ws[b], flags[b], &out[b]) returns, for b = 0 .. n - 1, with the
 * cubic work of all windows in two launches -- the sub-windows of the MARGIN_OLD windows (flag 0) are linearized by one launch (assembly A = sum J^T J, b = sum J^T r and elimination
 * of the dropped landmark blocks, marginalization_factor.cpp:232-276), then ONE launch eliminates every window's dropped frame block, forms the Schur complement and factors it
 * (J0 = sqrt(S) V^T, r0 = sqrt(S^-1) V^T b with the eps = 1e-8 cut, :278-291; a parallel cyclic Jacobi per window, csrc/uvs_marg_kernel.h).  MARGIN_SECOND_NEW windows (flag 1)
 * read their old prior only and join the second launch.  Host work per window (sub-window packing, the block tables) runs on the handle's packing threads.  status (may be NULL)
 * receives the per-window code; the return value is the first one that is not UVS_OK.  A window the device path does not take (a landmark or frame block the reference's eps cut
 * would touch, N = dropped + kept frame dofs > 96) is sent through uvs_marginalize() -- which, like every one-window call, may replace the handle's resident batch.  Synchronous; not
 * to be called while a uvs_marginalize_resident_begin() is in flight.  What a batched closed-loop replay calls between two uvs_batch_solve(). */
int uvs_marginalize_batch(uvs_solver *s, int n, const uvs_window *const *ws, const int *flags, uvs_prior *out, int *status);

/* ---- ONE large window spread over the GPU and, with an all-reduce between the steps, over several GPUs (BASELINE configs[3]) ----
 * Landmarks shard (rank r holds the landmarks k with k % G == r; frames / IMU / prior are replicated); the only exchanged data are
 * the pose-block partials returned by uvs_large_reduced() (SUM, except entry [n-7] which is a MAX) and the n = 6 scalars of
 * uvs_large_scalars() (SUM; the sixth is this rank's vote that options.max_solver_time_in_seconds is used up: reduced with the rest, so that
 * every rank ends the solve at the same iteration).  Both are DEVICE pointers so that RCCL can reduce them in place.  On one GPU skip the all-reduces
 * or call uvs_large_solve().  Loop: begin; while (!done) { if (need_linearize) { linearize; allreduce(reduced) } step; allreduce(scalars); decide } finish.
 * Relocalization blocks (n_relo_obs > 0) are taken on ONE rank only: a landmark shard cannot tell from its own observations whether relo_Pose is a
 * free block of the window, so in a solve over several ranks no rank may pass them (uvs_large_solve_fused answers UVS_ERR_UNSUPPORTED). */
int uvs_large_begin(uvs_solver *s, const uvs_window *w);
int uvs_large_need_linearize(const uvs_solver *s);
int uvs_large_linearize(uvs_solver *s);
double *uvs_large_reduced(uvs_solver *s, int *n);
int uvs_large_exchange_host(uvs_solver *s, int which, double *buf, int set);   /* host-staged get/set of the two vectors (no GPU-aware transport) */
int uvs_large_step(uvs_solver *s);
double *uvs_large_scalars(uvs_solver *s, int *n);
int uvs_large_decide(uvs_solver *s);
int uvs_large_done(const uvs_solver *s);
double uvs_large_local_x2(const uvs_solver *s);
void uvs_large_set_landmark_x2(uvs_solver *s, double all_ranks_x2);
int uvs_large_finish(uvs_solver *s, uvs_state *out, uvs_report *rep);
int uvs_large_solve(uvs_solver *s, const uvs_window *w, uvs_state *out, uvs_report *rep);

/* ---- fused multi-GPU loop: the library owns the RCCL communicator (SURVEY.md 8b "library owns ... RCCL comm") and keeps the
 * trust-region control on the device, so that one solve is one stream of launches with two in-place all-reduces per iteration and no
 * host round trip.  Usage on every rank (one process per GPU, one handle per process):
 *     rank 0: uvs_large_comm_unique_id(&id); broadcast `id` (128 bytes) to the other ranks by any host transport;
 *     uvs_large_comm_init(s, nranks, rank, &id);            (nranks <= 8; nranks == 1 needs no id and no RCCL)
 *     uvs_large_solve_fused(s, shard_of_this_rank, &state, &report, &ms);    w = the landmarks k with k % nranks == rank, frames / IMU /
 *                                                                            prior replicated; frames of `state` identical on all ranks
 *     uvs_large_comm_destroy(s);                             (also done by uvs_destroy)
 * RCCL is looked up at run time (a copy already loaded by the process is reused; UVS_RCCL_LIB overrides): UVS_ERR_UNSUPPORTED if absent.
 *
 * WITHOUT a communicator (no uvs_large_comm_init, or nranks == 1) uvs_large_solve_fused() is also the LOW-LATENCY form for ONE window of
 * any size on an otherwise idle GPU (the online case): the landmark chunks of every LM iteration run on many compute units, the frame
 * terms in a workgroup beside them, the reduced solve in one workgroup; one upload, one stream of launches, one download, one wait.
 * On MI355X a canonical 10-keyframe window takes 1.2 ms per call this way against 1.7 ms through uvs_solve_window() (which keeps the
 * whole solve on one compute unit and is what a BATCH of windows uses per window).  Same LM controller, same results to rounding
 * (tests/test_fused_single.py); relocalization blocks are taken on one rank (tests/test_relo.py). */
typedef struct uvs_rccl_id { char internal[128]; } uvs_rccl_id;      /* == ncclUniqueId */
int uvs_large_comm_unique_id(uvs_rccl_id *id);
int uvs_large_comm_init(uvs_solver *s, int nranks, int rank, const uvs_rccl_id *id);
void uvs_large_comm_destroy(uvs_solver *s);
int uvs_large_solve_fused(uvs_solver *s, const uvs_window *w, uvs_state *out, uvs_report *rep, float *loop_ms);

/* ---- size helpers for callers that serialise windows ---- */
int uvs_reduced_dim(const uvs_options *opts);  /* 165 (+6 if estimate_extrinsic) */

#ifdef __cplusplus
}
#endif
#endif /* UVS_SOLVER_H */
