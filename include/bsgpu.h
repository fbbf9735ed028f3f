/*
 * bsgpu.h — C-ABI of the MI355X-native fixed-lag-smoother solve path.
 *
 * This is the drop-in boundary for the ONE call the reference's optimizer makes
 * on its hot path:
 *
 *     summary_ = graph_->optimize(params_.solver_options);
 *         (reference: bs_optimizers/src/fixed_lag_smoother.cpp:281, and the ten
 *          other optimize()/optimizeFor() call sites listed in SURVEY.md §3.4)
 *
 * In the reference that call walks a fuse_core::Graph, builds a ceres::Problem
 * (AddParameterBlock / SetParameterBlockConstant / AddResidualBlock) and runs
 * ceres::Solve.  Nothing like a C interface exists there (it is C++ virtuals all
 * the way down), so the entry points below are what a fuse_core::Graph
 * implementation would bind to hand the flattened problem to the GPU:
 *
 *   bsgpu_set_blocks        <- Graph::createProblem: AddParameterBlock(data,size,
 *                              localParameterization) + SetParameterBlockConstant
 *                              for holdConstant() variables
 *   bsgpu_set_cameras       <- the (K, T_cam_baselink) pair every
 *                              EuclideanReprojectionConstraint carries
 *                              (bs_constraints/.../euclidean_reprojection_constraint.h:80-84)
 *   bsgpu_add_factors       <- AddResidualBlock(c.costFunction(), c.lossFunction(), blocks)
 *   bsgpu_solve             <- ceres::Solve(options, &problem, &summary)
 *   bsgpu_get_blocks        <- variables updated in place through Variable::data()
 *   bsgpu_get_iteration     <- summary.iterations[i]
 *   bsgpu_evaluate          <- ceres::Problem::Evaluate (used by the reference's tests,
 *                              bs_constraints/tests/euclidean_reprojection_test.cpp:150-180)
 *   bsgpu_covariance        <- Graph::getCovariance (bs_publishers/src/odometry_3d_publisher.cpp:82)
 *
 * Plain C types only: pointers, sizes, POD structs.  Host buffers are
 * caller-owned and copied on set/add; device memory is owned by the context.
 * All arithmetic is IEEE double, like the reference.
 *
 * Threading: a context is single-caller (the reference holds
 * optimization_mutex_ around optimize(), fixed_lag_smoother.cpp:185).
 *
 * Error model: every call returns BSGPU_OK (0) or a negative code;
 * bsgpu_last_error() returns a human-readable message for the last failure on
 * the context.  "NO_CONVERGENCE" is not an error (fixed_lag_smoother.cpp:284-285);
 * an unusable solution is reported through summary.is_solution_usable == 0 so
 * the caller can take the reference's fatal path (fixed_lag_smoother.cpp:286-295).
 */
#ifndef BSGPU_H_
#define BSGPU_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSGPU_ABI_VERSION 1

/* ---- return codes --------------------------------------------------------- */
enum {
  BSGPU_OK = 0,
  BSGPU_ERR_INVALID = -1,     /* bad argument / inconsistent problem          */
  BSGPU_ERR_DEVICE = -2,      /* HIP runtime failure or no device             */
  BSGPU_ERR_UNSUPPORTED = -3, /* problem shape the GPU path does not cover    */
  BSGPU_ERR_NUMERIC = -4      /* non-finite values / fatal linear-solver error */
};

/* ---- manifold kinds (fuse_core::Variable::localParameterization()) -------- */
enum {
  BSGPU_MANIFOLD_EUCLIDEAN = 0, /* nullptr parameterisation                    */
  BSGPU_MANIFOLD_QUAT_RIGHT = 1 /* fuse_variables::Orientation3DLocalParameterization:
                                   x (+) d = x (x) AngleAxisToQuaternion(d),
                                   (w,x,y,z) storage; restated in-tree by
                                   bs_constraints/src/jacobians.cpp:24-35,144-158 */
};

/* ---- loss kinds (fuse_core::Loss -> ceres::LossFunction) ------------------ */
enum {
  BSGPU_LOSS_TRIVIAL = 0,
  BSGPU_LOSS_CAUCHY = 1, /* ceres::CauchyLoss(a): rho(s) = a^2 log(1 + s/a^2)  */
  BSGPU_LOSS_HUBER = 2   /* ceres::HuberLoss(a)                                */
};

/* ---- factor types ----------------------------------------------------------
 * For every type: `block_idx` is n x BSGPU_NIDX(type) int32 (row-major) of
 * indices into the block table, in the constraint's variables() order;
 * `consts` is n x BSGPU_NCONST(type) doubles (row-major).
 */
enum {
  /* bs_constraints::EuclideanReprojectionConstraint
   *   (visual/euclidean_reprojection_function.h:28-179, SizedCostFunction<2,4,3,3>)
   *   idx   : q_WORLD_BASELINK, t_WORLD_BASELINK, P_WORLD, camera-table index
   *   consts: u, v (pixel), w (sqrt information = w * I2)                      */
  BSGPU_F_REPROJ = 0,
  /* bs_constraints::EuclideanReprojectionConstraintOnlineCalib
   *   (visual/euclidean_reprojection_functor_online_calib.h:16-83, AutoDiff<2,4,3,3,4,3>)
   *   idx   : q_WB, t_WB, P, q_BASELINK_CAM, p_BASELINK_CAM, camera-table index (K only)
   *   consts: u, v, w
   *   The two extrinsic blocks must be constant (bs_variables::Orientation3D /
   *   Position3D::holdConstant() == true, bs_variables/src/orientation_3d.cpp:39-41). */
  BSGPU_F_REPROJ_ONLINE_CALIB = 1,
  /* bs_constraints::RelativeImuState3DStampedConstraint
   *   (inertial/normal_delta_imu_state_3d_cost_functor.h:18-141, AutoDiff<15, 4,3,3,3,3, 4,3,3,3,3>)
   *   idx   : (q,p,v,bg,ba)_i, (q,p,v,bg,ba)_j
   *   consts: dt, dq[4 wxyz], dp[3], dv[3], dq_dbg[9], dp_dbg[9], dp_dba[9],
   *           dv_dbg[9], dv_dba[9] (3x3 row-major), bg_lin[3], ba_lin[3],
   *           A[225] (15x15 row-major = info_weight * sqrt_inv_cov)             */
  BSGPU_F_IMU_DELTA = 2,
  /* bs_constraints::AbsoluteImuState3DStampedConstraint
   *   (inertial/normal_prior_imu_state_3d_cost_functor.h:28-90, AutoDiff<15,4,3,3,3,3>)
   *   idx   : q,p,v,bg,ba        consts: b[16] (q wxyz,p,v,bg,ba), A[225]      */
  BSGPU_F_IMU_PRIOR = 3,
  /* bs_constraints::RelativePose3DStampedWithExtrinsicsConstraint
   *   (relative_pose/delta_pose_3d_with_extrinsics_cost_functor.h:19-109, AutoDiff<6,3,4,3,4,3,4>)
   *   idx   : p1,q1,p2,q2,p_ext,q_ext   consts: d[7] (x,y,z,qw,qx,qy,qz), A[36] */
  BSGPU_F_RELPOSE_EXT = 4,
  /* fuse_constraints::RelativePose3DStampedConstraint (NormalDeltaPose3DCostFunctor)
   *   idx   : p1,q1,p2,q2        consts: d[7], A[36]                           */
  BSGPU_F_RELPOSE = 5,
  /* fuse_constraints::AbsolutePose3DStampedConstraint and
   * bs_constraints::AbsolutePose3DConstraint (global/absolute_pose_3d_constraint.cpp:12-52)
   *   idx   : p,q                consts: b[7] (x,y,z,qw,qx,qy,qz), A[36]       */
  BSGPU_F_ABSPOSE = 6,
  /* fuse_constraints::AbsoluteConstraint<V> for 3-vectors (global/absolute_constraint.h:10-25)
   *   idx   : x                  consts: b[3], A[9];  r = A (x - b)             */
  BSGPU_F_ABS_VEC3 = 7,
  /* fuse_constraints::RelativeConstraint<V> (relative_pose/relative_constraints.h:12-19)
   *   idx   : x1,x2              consts: d[3], A[9];  r = A ((x2 - x1) - d)     */
  BSGPU_F_REL_VEC3 = 8,
  /* bs_constraints::GravityAlignmentStampedConstraint
   *   (global/gravity_alignment_cost_functor.h:32-82, AutoDiff<2,4>)
   *   idx   : q                  consts: g_b[3], A[4] (2x2 row-major)          */
  BSGPU_F_GRAVITY = 9,
  /* bs_constraints::InverseDepthReprojectionConstraint
   *   (visual/inversedepth_reprojection_functor.h:15-136, AutoDiff<2,4,3,4,3,1>;
   *    src/visual/inversedepth_reprojection_constraint.cpp:14-49)
   *   idx   : q_WORLD_BASELINKa, p_WORLD_BASELINKa (anchor), q_WORLD_BASELINKm,
   *           p_WORLD_BASELINKm (measurement), rho (bs_variables::InverseDepthLandmark,
   *           size 1), camera-table index
   *   consts: u, v (pixel), w (sqrt information = w * I2), m[3] (bearing of the
   *           landmark in the anchor camera, InverseDepthLandmark::bearing())
   *   An inverse-depth scalar that only such factors use is eliminated on the
   *   landmark side like a Euclidean landmark (csrc/k_idp.hip; its covariance
   *   cannot be queried then; BSGPU_IDP_ELIM=0 at finalize keeps it in the
   *   reduced system).  use_idp is off in every shipped configuration.         */
  BSGPU_F_IDP_REPROJ = 10,
  /* bs_constraints::InverseDepthReprojectionConstraintUnary
   *   (visual/inversedepth_reprojection_functor_unary.h:14-85, AutoDiff<2,4,3,1>)
   *   idx   : q_WORLD_BASELINKa, p_WORLD_BASELINKa, rho, camera-table index
   *   consts: u, v, w, m[3]
   *   The observation in the anchor frame: the residual is constant in every
   *   block (zero Jacobian), it only contributes to the cost.                  */
  BSGPU_F_IDP_REPROJ_UNARY = 11,
  BSGPU_F_NUM_TYPES = 12
};

/* number of int32 per factor in block_idx / doubles per factor in consts /
 * residual rows, for a type; -1 for an unknown type */
int bsgpu_nidx(int type);
int bsgpu_nconst(int type);
int bsgpu_nres(int type);

/* ---- linear solver choice (ceres::Solver::Options::linear_solver_type) ----- */
enum {
  BSGPU_LINEAR_AUTO = 0,         /* exact path whenever the reduced system fits */
  BSGPU_LINEAR_SCHUR_CHOLESKY = 1, /* landmark Schur complement + dense FP64
                                      Cholesky of the reduced system: the exact
                                      (SPARSE_NORMAL_CHOLESKY-equivalent) step   */
  BSGPU_LINEAR_PCG = 2,          /* block-Jacobi PCG on the block-sparse normal
                                      equations (inexact; pose-graph sized problems) */
  BSGPU_LINEAR_SCHUR_PCG = 3     /* landmark Schur complement + block-Jacobi PCG on the
                                      reduced camera system (Ceres ITERATIVE_SCHUR with
                                      SCHUR_JACOBI, the preconditioner named by
                                      beam_slam_launch/config/optimization/ceres_config.json:12;
                                      inexact: pcg_tolerance / pcg_max_iterations)        */
};

/* ---- termination (ceres::TerminationType) ---------------------------------- */
enum {
  BSGPU_CONVERGENCE = 0,
  BSGPU_NO_CONVERGENCE = 1,
  BSGPU_FAILURE = 2
};

/* mirrors the ceres::Solver::Options fields the reference sets
 * (beam_slam_launch/config/vio.yaml:7-17) plus the Ceres defaults it relies on */
typedef struct bsgpu_options {
  int32_t max_num_iterations;           /* vio.yaml:13 -> 10; Ceres default 50  */
  int32_t linear_solver_type;           /* BSGPU_LINEAR_*                       */
  int32_t jacobi_scaling;               /* Ceres default 1                      */
  int32_t max_num_consecutive_invalid_steps; /* Ceres default 5                 */
  double max_solver_time_in_seconds;    /* vio.yaml:14 -> 0.05; <=0 = unlimited */
  double function_tolerance;            /* vio.yaml:17 -> 1.5e-7; default 1e-6  */
  double gradient_tolerance;            /* vio.yaml:15 -> 1.5e-7; default 1e-10 */
  double parameter_tolerance;           /* vio.yaml:16 -> 1.5e-7; default 1e-8  */
  double initial_trust_region_radius;   /* 1e4                                  */
  double max_trust_region_radius;       /* 1e16                                 */
  double min_trust_region_radius;       /* 1e-32                                */
  double min_relative_decrease;         /* 1e-3                                 */
  double min_lm_diagonal;               /* 1e-6                                 */
  double max_lm_diagonal;               /* 1e32                                 */
  int32_t pcg_max_iterations;           /* BSGPU_LINEAR_PCG only                */
  int32_t reserved0;
  double pcg_tolerance;                 /* relative residual |r| / |b| at which an inner solve stops.  Default 1e-10: the reference's
                                           step on this path is the exact SPARSE_NORMAL_CHOLESKY one, and a default-option caller (the
                                           global mapper) gets a step that is equivalent to it (69 inner iterations per LM step on the
                                           5 000-pose graph of BASELINE config 4; 84 at 1e-12).  A caller that wants the inexact step
                                           sets it: at 1e-6 that graph's per-iteration costs stay within 1e-7 and the final cost within
                                           1e-10 of the exact trajectory at 39 inner iterations per LM step (scripts/c4_tolerance.py:
                                           ONE synthetic graph -- measured evidence, not a guarantee; bench.py reports both). */
} bsgpu_options;

/* fills `o` with Ceres' defaults */
void bsgpu_options_default(bsgpu_options* o);
/* fills `o` with the solver options the reference ships for VIO
 * (beam_slam_launch/config/vio.yaml:7-17) */
void bsgpu_options_vio(bsgpu_options* o);

/* mirrors the ceres::Solver::Summary fields the reference reads back
 * (fixed_lag_smoother.cpp:286,705-716) */
typedef struct bsgpu_summary {
  int32_t termination_type;      /* BSGPU_CONVERGENCE / NO_CONVERGENCE / FAILURE */
  int32_t is_solution_usable;    /* Summary::IsSolutionUsable()                  */
  int32_t num_iterations;        /* iterations.size() - 1 (iteration 0 = initial evaluation) */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int32_t num_parameters_tangent; /* columns of the reduced problem              */
  int32_t num_residuals;
  int32_t linear_solver_used;    /* BSGPU_LINEAR_*                               */
  int32_t num_linear_solves;     /* trust-region steps computed (incl. the one a
                                    parameter/function-tolerance exit does not
                                    record in `iterations`): the unit of the
                                    "LM iterations/s" metric                     */
  int32_t num_inner_iterations;  /* PCG iterations summed over the LM steps (0 on the exact path) */
  double initial_cost;
  double final_cost;
  double fixed_cost;             /* cost of residual blocks with only constant blocks */
  double total_time_in_seconds;  /* wall clock of bsgpu_solve                    */
  double device_time_in_seconds; /* HIP-event time of the LM loop on the stream  */
  double time_eval_seconds;      /* host-timed phase splits (sum <= total)       */
  double time_assemble_seconds;
  double time_linear_solve_seconds;
  char message[160];
} bsgpu_summary;

/* one entry of ceres::Solver::Summary::iterations */
typedef struct bsgpu_iteration {
  int32_t iteration;
  int32_t step_is_valid;
  int32_t step_is_successful;
  int32_t reserved0;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double gradient_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
  double model_cost_change;
} bsgpu_iteration;

/* one camera-table entry: K (skew-free) and T_cam_baselink */
typedef struct bsgpu_camera {
  double fx, fy, cx, cy;
  double R_cam_baselink[9]; /* row-major */
  double t_cam_baselink[3];
} bsgpu_camera;

typedef struct bsgpu_ctx bsgpu_ctx;

/* ---- life cycle ------------------------------------------------------------ */
/* Creates a context on HIP device `device`.  Returns NULL when no HIP device is
 * usable (there is no CPU fallback); bsgpu_create_error() then says why.       */
bsgpu_ctx* bsgpu_create(int device);
const char* bsgpu_create_error(void);
void bsgpu_destroy(bsgpu_ctx* ctx);
const char* bsgpu_last_error(const bsgpu_ctx* ctx);
int bsgpu_abi_version(void);

/* ---- problem definition ---------------------------------------------------- */
/* Drops blocks, cameras and factors (Graph::clear()). */
int bsgpu_clear(bsgpu_ctx* ctx);

/* Parameter-block table.  `values` is the concatenation of all blocks' ambient
 * coordinates; block b occupies values[offset[b] .. offset[b]+size[b]).
 * manifold[b] in BSGPU_MANIFOLD_*, is_const[b] != 0 <=> SetParameterBlockConstant.
 * Block order defines the deterministic variable index:
 * tangent columns are numbered in block order, pose-side blocks first, then the
 * landmark blocks eliminated by the Schur complement.                           */
int bsgpu_set_blocks(bsgpu_ctx* ctx, int32_t n_blocks, const double* values,
                     const int32_t* offset, const uint8_t* size,
                     const uint8_t* manifold, const uint8_t* is_const);

/* Overwrites the current block values (same layout as bsgpu_set_blocks). */
int bsgpu_set_values(bsgpu_ctx* ctx, const double* values, int64_t n_values);

int bsgpu_set_cameras(bsgpu_ctx* ctx, int32_t n_cameras, const bsgpu_camera* cams);

/* Appends n factors of one type.  loss_kind / loss_a may be NULL (trivial loss).
 * Factor order (type-major, then insertion order) defines the residual index.   */
int bsgpu_add_factors(bsgpu_ctx* ctx, int32_t type, int32_t n,
                      const int32_t* block_idx, const double* consts,
                      const int32_t* loss_kind, const double* loss_a);
/* bsgpu_add_factors for a caller that keeps its factor tables across solves: the block-index columns of `slot_idx` hold
 * caller-side variable slots (stable for the lifetime of a variable) and are translated through
 * slot_to_block[n_slots] (slot -> index in the current bsgpu_set_blocks table) while the rows are copied in; camera
 * columns are copied as they are.  A slot outside [0, n_slots) or mapped to a negative block is an error.  This is what
 * lets bs_optimizers::GpuGraph (beam_slam_amd/host/gpu_graph.h) hand over its persistent packed tables unchanged
 * every cycle although the block order shifts when the window slides.                    */
int bsgpu_add_factors_indirect(bsgpu_ctx* ctx, int32_t type, int32_t n, const int32_t* slot_idx, int32_t n_slots,
                               const int32_t* slot_to_block, const double* consts, const int32_t* loss_kind,
                               const double* loss_a);
/* bsgpu_add_factors_indirect for a type's WHOLE table (one call per type per description, before any other factors of the
 * type) from a caller that also knows what changed: changed_rows[n_changed] lists every row r < n whose contents differ
 * from row r of the table passed to the previous bsgpu_sync_factors_indirect call for this type on this context (rows
 * beyond the previous table's end included; rows that left at the end need no mention).  n_changed < 0: no promise, the
 * table is read whole (the first call).  For BSGPU_F_REPROJ the back-end keeps slot-named copies of the table on the host
 * and on the device ACROSS bsgpu_clear() and patches them — a window that slides by one keyframe sends ~3 500 of 400 000
 * rows, validation and landmark detection run on per-slot use counters, and bsgpu_finalize flattens from the resident
 * device table (the per-cycle rebuild as a delta).  Other types are copied in as by
 * bsgpu_add_factors_indirect.  A wrong change list is the caller's error: BSGPU_SYNC_CHECK=1 in the environment makes the
 * call compare its copy with the table passed and fail on any difference; BSGPU_SYNC_FULL=1 ignores the lists.            */
int bsgpu_sync_factors_indirect(bsgpu_ctx* ctx, int32_t type, int32_t n, const int32_t* slot_idx, int32_t n_slots,
                                const int32_t* slot_to_block, const double* consts, const int32_t* loss_kind,
                                const double* loss_a, int32_t n_changed, const int32_t* changed_rows);

/* Dense linear prior: [EXT] fuse_constraints::MarginalConstraint, what fuse_constraints::marginalizeVariables
 * adds to the graph (bs_optimizers/src/fixed_lag_smoother.cpp:270-271, `pseudo_marginalization: false`):
 *     r = b + sum_i A_i (x_i [-] xbar_i),    [-] = LocalParameterization::Minus(xbar_i, x_i)
 *   blocks : n_blocks indices into the block table (set_blocks first)
 *   A      : n_rows x (sum of the blocks' tangent sizes), row-major, columns in `blocks` order
 *   b      : n_rows
 *   xbar   : the blocks' linearisation points, concatenated (ambient sizes)
 * Jacobian as fuse's MarginalCostFunction: A_i MinusJacobian(x_i), brought to the tangent space with
 * PlusJacobian(x_i) (= A_i for a unit quaternion).  No loss function.  Its residual rows come after those of all fixed-size factor types, in
 * insertion order.  Blocks it touches are never Schur-eliminated.                                            */
int bsgpu_add_marginal(bsgpu_ctx* ctx, int32_t n_blocks, const int32_t* blocks, int32_t n_rows,
                       const double* A, const double* b, const double* xbar);
/* Replaces A, b and xbar of the index-th prior added with bsgpu_add_marginal (same blocks, same n_rows) IN PLACE: nothing is
 * re-flattened, the device tables of a finalized problem stay.  For priors whose contents change from one solve to the next
 * while the graph does not: [EXT] fuse_core::Graph::removeConstraint + addConstraint of a MarginalConstraint on the same
 * variables.  n_rows x n_cols = the shape of A (b: n_rows), n_xbar = the ambient size of xbar: they must be those of the prior
 * as it was added (BSGPU_ERR_INVALID otherwise — a payload of another shape would be read out of bounds).                     */
int bsgpu_update_marginal(bsgpu_ctx* ctx, int32_t index, int32_t n_rows, int32_t n_cols, int32_t n_xbar, const double* A, const double* b,
                          const double* xbar);

/* ---- solve ----------------------------------------------------------------- */
/* Uploads / builds the device-side structure (sorted factor tables, the tile plan of the
 * reduced system): what [EXT] fuse_core::Graph::optimize does in HashGraph::createProblem
 * before ceres::Solve (bs_optimizers/src/fixed_lag_smoother.cpp:281).  Called implicitly by
 * bsgpu_solve when the problem changed; exposed so a caller can keep it out of a timed region. */
int bsgpu_finalize(bsgpu_ctx* ctx);

/* Levenberg-Marquardt (Ceres TrustRegionMinimizer semantics) on the device.
 * On return the best accepted point is the context's current value set.         */
int bsgpu_solve(bsgpu_ctx* ctx, const bsgpu_options* options, bsgpu_summary* summary);
/* Several windows at once — what the reference does with one thread per optimiser (local smoother, global mapper and the
 * submap refinements side by side: bs_models/src/global_mapping/submap_refinement.cpp:35-115): bsgpu_solve of n DISTINCT
 * contexts.  Windows with eliminated Euclidean landmarks on one device (visual / visual-inertial windows, the reference's
 * local smoother and submap refinements) advance TOGETHER: every kernel of the LM step is launched once for all of them
 * (blockIdx.y = window, per-window argument tables; csrc/bsgpu_batch.cpp), the trust-region decisions are taken per window
 * on the host, converged windows drop out — each window's iterations are those of its lone bsgpu_solve.  Any other window
 * (pose graphs, PCG, inverse-depth landmarks, dense priors, other devices) is driven by a host thread of the library on its
 * context's own stream, in the same call — so is a window whose Jacobi-scaling flag or LM-diagonal bounds differ from the
 * first batched window's.  `options` holds one entry (shared) when options_stride == 0, else n entries; summaries: n
 * entries (device_time_in_seconds of a batched window = the batch's).  Every solve runs to its end; returns BSGPU_OK or the code
 * of the first context (lowest index) that failed — its message is that context's bsgpu_last_error.                      */
int bsgpu_solve_batch(bsgpu_ctx* const* ctxs, int32_t n, const bsgpu_options* options, int32_t options_stride,
                      bsgpu_summary* summaries);
/* Process-wide counters of bsgpu_solve_batch (tests, measurements): windows solved by the batched launches so far, and the
 * rounds (one set of launches each) they took.                                                                            */
int bsgpu_batch_stats(int64_t* windows_batched, int64_t* rounds);

/* Copies the current values back (device -> host), layout of bsgpu_set_blocks.  */
int bsgpu_get_blocks(bsgpu_ctx* ctx, double* values, int64_t n_values);

/* Restores the device-resident values to what bsgpu_set_blocks/set_values last
 * uploaded (device-to-device) — lets a benchmark re-solve the same window
 * without touching PCIe.                                                        */
int bsgpu_reset_values(bsgpu_ctx* ctx);

int bsgpu_num_iterations_recorded(const bsgpu_ctx* ctx);
int bsgpu_get_iteration(const bsgpu_ctx* ctx, int32_t i, bsgpu_iteration* out);

/* ---- evaluation (ceres::Problem::Evaluate) --------------------------------- */
/* Evaluates at the current values with the robust-loss corrector applied.
 * Any output may be NULL.
 *   cost      : 1/2 sum rho(|r|^2)                          (1 double)
 *   residuals : num_residuals doubles, factor order = type-major insertion order
 *   gradient  : num_parameters_tangent doubles (J^T r)
 *   jacobian  : dense row-major num_residuals x num_parameters_tangent (only for
 *               small problems: refuses above 64M entries)                      */
int bsgpu_evaluate(bsgpu_ctx* ctx, double* cost, double* residuals,
                   double* gradient, double* jacobian);
int bsgpu_num_residuals(const bsgpu_ctx* ctx);
int bsgpu_num_parameters_tangent(const bsgpu_ctx* ctx);
/* tangent offset of block b in the reduced problem, -1 for constant blocks */
int bsgpu_tangent_offset(const bsgpu_ctx* ctx, int32_t block);

/* ---- true marginalisation ---------------------------------------------------
 * [EXT] fuse_constraints::marginalizeVariables(source, vars_to_marginalize, graph)
 * (bs_optimizers/src/fixed_lag_smoother.cpp:270-271, `pseudo_marginalization: false`) on the device:
 * every factor that touches one of `marg_blocks` is linearised at the current values, the marginalised blocks are
 * eliminated (Schur complement) and the result is the dense linear prior on the other non-constant blocks those
 * factors touch, in the form bsgpu_add_marginal() takes (A upper-trapezoidal, A^T A = marginal information,
 * A^T b = marginal gradient, xbar = current values of the kept blocks).  Directions of the kept blocks the
 * eliminated factors carry no information about give no row (fuse's QR leaves a zero row there).
 * The caller then removes those factors and the marginalised blocks and adds the prior — the transaction
 * marginalizeVariables returns.  The context itself is not modified.
 *   n_kept / n_rows / n_cols : out — number of kept blocks, rows and columns (sum of tangent sizes) of A     */
int bsgpu_marginalize(bsgpu_ctx* ctx, int32_t n_marg, const int32_t* marg_blocks,
                      int32_t* n_kept, int32_t* n_rows, int32_t* n_cols);
/* Result of the last bsgpu_marginalize: kept_blocks[n_kept] (ascending), A[n_rows*n_cols] row-major (columns in
 * kept_blocks order), b[n_rows], xbar[sum of the kept blocks' sizes].                                       */
int bsgpu_get_marginal(const bsgpu_ctx* ctx, int32_t* kept_blocks, double* A, double* b, double* xbar);

/* ---- covariance (Graph::getCovariance) ------------------------------------- */
/* Marginal covariance block (tangent space) between two pose-side blocks at the
 * current values: out is ts(block_a) x ts(block_b) row-major.                   */
int bsgpu_covariance(bsgpu_ctx* ctx, int32_t block_a, int32_t block_b, double* out);
/* The JOINT marginal covariance of several pose-side blocks (distinct, not constant; their tangent dimensions add up to D <= 64):
 * out is D x D row-major, the blocks' tangent coordinates in the order given.  One undamped assembly + one factorisation, as
 * bsgpu_covariance.  What a submap's summary of itself on its boundary key frames is made of (the unit of independence of
 * bs_models/src/lib/global_mapping/submap_refinement.cpp:35-115; shared-pose consensus, beam_slam_amd/sharding.py).            */
int bsgpu_covariance_joint(bsgpu_ctx* ctx, int32_t n_blocks, const int32_t* blocks, double* out);

/* ---- factor producers either side of the solve ---------
 * Pixel error |z - projection| of every reprojection factor (types REPROJ then REPROJ_ONLINE_CALIB, insertion
 * order) at the current values, un-weighted and without loss — the screening quantity of
 * bs_models/src/visual_odometry.cpp:1247-1272 (ComputeAverageReprojection).  -1 for a point not in front of the
 * camera.  err: bsgpu_nfactors-sized, i.e. n(REPROJ) + n(REPROJ_ONLINE_CALIB) doubles.                          */
int bsgpu_reprojection_errors(bsgpu_ctx* ctx, double* err);

/* bs_common::PreIntegrator::Integrate (bs_common/src/bs_common/preintegrator.cpp:26-143) for a batch of keyframe
 * intervals on the device: interval i integrates samples [sample_start[i], sample_start[i+1]) (time-ordered,
 * t / gyro w[3] / accel a[3] per sample) up to t_end[i] with the bias estimates bg[3i..], ba[3i..], and the
 * continuous-time noise covariances cov_w, cov_a, cov_bg, cov_ba (3x3 row-major each).
 * consts_out: n_intervals x 287 doubles — the constant payload of BSGPU_F_IMU_DELTA (dt, dq, dp, dv, bias
 * Jacobians, bias linearisation point, A = info_weight * sqrt_inv_cov), ready for bsgpu_add_factors.           */
int bsgpu_preintegrate(int device, int32_t n_intervals, const int32_t* sample_start, const double* t, const double* w,
                       const double* a, const double* t_end, const double* bg, const double* ba, const double* cov_w,
                       const double* cov_a, const double* cov_bg, const double* cov_ba, double info_weight,
                       double* consts_out);

/* Landmark triangulation for a batch of feature tracks at the context's CURRENT values (after a solve: the values the
 * solve left on the device) — VisualOdometry::TriangulateLandmark (bs_models/src/visual_odometry.cpp:532-610) and
 * SLAMInitialization::TriangulateLandmark (bs_models/src/slam_initialization.cpp:699-701), i.e. the [EXT]
 * beam_cv::Triangulation::TriangulatePoint(cam, T_cam_world, pixels, max_dist, max_reprojection) call they make.
 * Track i holds views [track_start[i], track_start[i+1]); view o is seen from the keyframe whose orientation /
 * position blocks are q_block[o] / p_block[o] with the measured pixel pixels[2o..2o+1]; `camera` indexes the
 * bsgpu_set_cameras table (K and T_cam_baselink).  truncate_pixels != 0 reproduces the reference's
 * `m.value.cast<int>()` (visual_odometry.cpp:547).  max_dist / max_reproj <= 0 disable that check (the
 * `track_lost_` call at visual_odometry.cpp:600 passes neither; vo_params.json:2-3 ships 30 m / 20 px).
 * points: n_tracks x 3 (world frame); status: n_tracks, 0 = triangulated, 1 = fewer than 2 views (:572),
 * 2 = behind a camera, 3 = farther than max_dist, 4 = re-projection above max_reproj, 5 = point at infinity.   */
int bsgpu_triangulate(bsgpu_ctx* ctx, int32_t n_tracks, const int32_t* track_start, const int32_t* q_block,
                      const int32_t* p_block, const double* pixels, int32_t camera, int32_t truncate_pixels,
                      double max_dist, double max_reproj, double* points, int32_t* status);

/* ---- measurement helpers (used by bench.py only) --------------------------- */
/* Launches the Jacobian-evaluation kernel of the reprojection factors `reps`
 * times on the context's stream between two HIP events and returns the average
 * milliseconds per launch (<0 on error).                                        */
double bsgpu_time_reproj_jacobian_ms(bsgpu_ctx* ctx, int32_t reps);
/* The same for EVERY factor type of the problem (reprojection, IMU, relative / absolute pose, priors): mean milliseconds of one
 * evaluation of residuals + Jacobians at the current values, and the algorithmic bytes of that evaluation (SURVEY.md 8(d):
 * 196-200 B per reprojection factor, ~990 B per relative-pose factor, ~6.1 KB per IMU factor).  Measurement hooks: what
 * CostFunction::Evaluate costs per LM iteration ([EXT] ceres::Problem::Evaluate inside fixed_lag_smoother.cpp:281).        */
double bsgpu_time_eval_ms(bsgpu_ctx* ctx, int32_t reps);
int64_t bsgpu_eval_bytes(const bsgpu_ctx* ctx);
/* Block rows and non-zero 3x3 blocks of the block-sparse J^T J of the PCG path (both 0 on the dense Schur path).           */
int bsgpu_bsr_info(bsgpu_ctx* ctx, int32_t* block_rows, int32_t* nnz_blocks);
/* Algorithmic bytes one launch of that kernel moves. */
int64_t bsgpu_reproj_jacobian_bytes(const bsgpu_ctx* ctx);
/* Measurement: the phases of a full LM step timed IN SITU — `reps` steps exactly as bsgpu_solve enqueues them for an accepted step
 * (dense Schur path), a HIP event at every phase boundary on the solver's stream.  ms_out[BSGPU_PHASE_NUM]: mean milliseconds
 * per phase; work_out[BSGPU_PHASE_NUM] (may be NULL): the algorithmic work of the phase's dominant kernel — bytes, or FP64 flops
 * for BSGPU_PHASE_FACTOR — 0 where none is defined.  The values return to those of the last finalize. */
enum {
  BSGPU_PHASE_EVAL_REPROJ = 0,    /* reproj_eval_kernel<true>: residuals + Jacobians of the reprojection factors */
  BSGPU_PHASE_EVAL_OTHER = 1,     /* IMU / relative-pose / prior factors */
  BSGPU_PHASE_LANDMARK = 2,       /* clear + landmark_kernel: H_ll, its Cholesky, C and rho per factor */
  BSGPU_PHASE_PAIRS = 3,          /* pairs_kernel: camera-pair blocks of the reduced system */
  BSGPU_PHASE_ASSEMBLE_OTHER = 4, /* pose-only factors, LM diagonal, gradient norms */
  BSGPU_PHASE_FACTOR = 5,         /* Cholesky of the reduced camera system (FP64 MFMA) */
  BSGPU_PHASE_BACKSOLVE = 6,      /* L^T y = y' */
  BSGPU_PHASE_BACKSUB = 7,        /* landmark back-substitution + model-cost terms */
  BSGPU_PHASE_CANDIDATE = 8,      /* x [+] delta, cost at the candidate, end-of-step reduction */
  BSGPU_PHASE_NUM = 9
};
int bsgpu_profile_step(bsgpu_ctx* ctx, const bsgpu_options* options, int32_t reps, double* ms_out, double* work_out);

/* Stand-alone dense SPD solve A x = b (row-major n x n, host pointers) through the kernels the
 * reduced camera system uses after Schur elimination — test and measurement hook for the FP64
 * MFMA Cholesky.  The tile structure is taken from the non-zeros of A; max_chains caps the number
 * of independent sub-chains of the nested-dissection ordering (<= 1: natural order).
 * ms_out: HIP-event time.                                                                       */
int bsgpu_dense_solve(int device, int32_t n, const double* A, const double* b, double* x,
                      int32_t max_chains, double* ms_out);
/* Diagnostics of the tiled-Cholesky plan of the finalized problem. */
int bsgpu_plan_info(const bsgpu_ctx* ctx, int32_t* n_chains, int32_t* n_steps, int32_t* n_tiles);
/* What the elimination order of the reduced camera system is planned for (the [EXT] choice Ceres makes once for everyone in
 * ceres::Solver::Options::linear_solver_ordering_type; the reference's fixed-lag smoother takes the default,
 * bs_optimizers/src/fixed_lag_smoother.cpp:281 through fuse_core::Graph::optimize):
 *   BSGPU_PLAN_LATENCY (default)  one window solved by itself: a small system (<= 2 000 reduced dimensions) is planned under several
 *                                 settings of the dissection's cost model and keeps the one whose task list replays shortest;
 *   BSGPU_PLAN_THROUGHPUT         the window is one of many advanced side by side (bsgpu_solve_batch): the setting with the fewest
 *                                 supernodes — a batch is bound by the number of its factorisation workgroups, not by one window's path.
 * Takes effect at the next bsgpu_finalize() (a finalized context is planned again).                                                       */
enum { BSGPU_PLAN_LATENCY = 0, BSGPU_PLAN_THROUGHPUT = 1 };
int bsgpu_set_plan_preference(bsgpu_ctx* ctx, int32_t preference);

#ifdef __cplusplus
}
#endif
#endif /* BSGPU_H_ */
