/* dfx.h -- C ABI of libdfx, the MI355X engine for DifFlexMM's hot path.
 *
 * The library replaces, for one lattice, what the reference builds in Python/JAX:
 *
 *   dfx_create      <->  setup_dynamic_solver(...)            difflexmm/dynamics.py:60-136
 *                        (static data: connectivity, bond model, contact on/off, which DOFs are
 *                        constrained / loaded and by which time function)
 *   dfx_set_params  <->  the ControlParams pytree passed to solve_dynamics
 *                                                            difflexmm/utils.py:48-163
 *   dfx_forward     <->  solve_dynamics(state0, timepoints, control_params)
 *                                                            difflexmm/dynamics.py:138-184
 *                        (odeint call at dynamics.py:166 + reconstruction :169-182)
 *   dfx_adjoint     <->  the VJP jax.grad takes through solve_dynamics
 *                        (problems/quads_focusing.py:565; jax.experimental.ode._odeint_rev)
 *   dfx_rhs         <->  rhs(state, t, control_params, inertia)  difflexmm/dynamics.py:33-55
 *   dfx_rhs_vjp     <->  jax.vjp(rhs, ...)                     (test hook for the adjoint kernel)
 *
 * Conventions: every function returns 0 on success, non-zero on failure (message via
 * dfx_last_error).  All arrays are C-contiguous float64 / int32 HOST buffers owned by the
 * caller; the library copies what it needs and keeps no caller pointer after a call returns.
 * Device memory lives inside the handle.  A handle belongs to one (process, device) and is not
 * thread safe.  `batch` independent members (designs / inputs of one lattice) are integrated
 * side by side; every per-member array has a leading batch axis.
 *
 * DOF layout (geometry.py:174-175): dof = 3*block + {0:x, 1:y, 2:theta};  node = n_npb*block + local.
 */
#ifndef DFX_H
#define DFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFX_MAX_FNS 2         /* time functions per problem                                   */
#define DFX_FN_PARAMS 5       /* parameters per time function                                 */

enum { DFX_BOND_LINEARIZED = 0, DFX_BOND_NONLINEAR = 1,         /* energy.py:99 / energy.py:158 */
       DFX_BOND_SIMPLE_SPRING = 2,                              /* energy.py:30-48: k_stretch (|dU + l0| - |l0|)^2 / 2; k_bond[1..2] ignored */
       DFX_BOND_STRETCH_TORSION = 3 };                          /* energy.py:51-67: zero-length spring, k_stretch |dU|^2/2 + k_rot dtheta^2/2;
                                                                   k_bond[1] and reference_vector ignored (pass any non-zero vector) */
enum { DFX_CONTACT_NONE = 0, DFX_CONTACT_ANGLE = 1,            /* energy.py:364 (angle_based)  */
       DFX_CONTACT_DISTANCE = 2 };                              /* energy.py:222-330 (angle_based=False): void-edge distances;
                                                                   `contact` = (min, cutoff, k) are lengths, needs block_centroids */
enum { DFX_TABLEAU_DOPRI5 = 0, DFX_TABLEAU_RK4 = 1 };
/* time-function library; parameters p[] in this order */
enum {
  DFX_FN_ZERO = 0,
  DFX_FN_PULSE = 1,        /* (A, f, t_d)   problems/quads_focusing.py:211-222                   */
  DFX_FN_HARMONIC = 2,     /* (A, f, t_d)   problems/quads_spin.py:210-222                       */
  DFX_FN_RAMP = 3,         /* (A, r)        tests/test_difflexmm.py:85-86                        */
  DFX_FN_SECH2TANH = 4,    /* (A, s)        scripts/pulse_RS.py:49-50                            */
  DFX_FN_CONSTANT = 5,     /* (A)                                                                 */
  DFX_FN_RAMP_CAP = 6,     /* (L, r, cap): L min(t r, cap)  static compression, problems/quads_kinetic_energy_static_tuning.py:176-182;
                              the delayed pulse of :184-186 is DFX_FN_PULSE with t_d = cap / r + input_delay (host-side chain rule) */
  DFX_FN_TABLE = 7         /* (A, t_d): A * interp(t - t_d; table), piecewise linear, end values held (jnp.interp semantics):
                              a recorded input signal; the table itself is static data of dfx_problem             */
};

/* A block that has constrained and/or force-loaded DOFs.  u[dof] = sum_m con_coef[d][m] * g_m(t)
 * for constrained DOFs (kinematics.py:68-73); F_load[dof] = sum_m load_coef[d][m] * g_m(t)
 * (loading.py:36-45). */
typedef struct dfx_special {
  int32_t block;
  int32_t con_mask;                       /* bit d set: DOF d of the block is constrained       */
  double con_coef[3][DFX_MAX_FNS];
  double load_coef[3][DFX_MAX_FNS];
} dfx_special;

typedef struct dfx_problem {
  int32_t n_blocks;
  int32_t n_npb;                          /* nodes per block: 3 (kagome) or 4 (quads)           */
  int32_t n_bonds;
  const int32_t* bonds;                   /* (n_bonds, 2) node ids, geometry.bond_connectivity() */
  int32_t bond_model;                     /* DFX_BOND_*                                          */
  int32_t contact;                        /* DFX_CONTACT_*                                       */
  int32_t n_special;
  const dfx_special* special;
  int32_t n_fns;                          /* <= DFX_MAX_FNS                                      */
  int32_t fn_type[DFX_MAX_FNS];           /* DFX_FN_*                                            */
  int32_t batch;                          /* ensemble members integrated together                */
  int32_t tableau;                        /* DFX_TABLEAU_*                                       */
  int32_t device;                         /* HIP device ordinal (ignored by the CPU port)        */
  int32_t fn_table_n[DFX_MAX_FNS];        /* DFX_FN_TABLE: number of breakpoints (>= 2), else 0   */
  const double* fn_table[DFX_MAX_FNS];    /* DFX_FN_TABLE: fn_table_n increasing times, then fn_table_n values */
  int32_t streams;                        /* member groups advanced on their own HIP streams; 0 = choose (2 when the launches fill the
                                             chip, else 1).  Several engines driven concurrently (multi-input problems) run best with 1 */
} dfx_problem;

/* ControlParams, flattened (utils.py:48-163).  Leading axis of every array = batch. */
typedef struct dfx_params {
  const double* centroid_node_vectors;    /* (batch, n_blocks, n_npb, 2)                         */
  const double* reference_vector;         /* (batch, n_bonds, 2)                                 */
  const double* k_bond;                   /* (batch, n_bonds, 3) = k_stretch, k_shear, k_rot     */
  const double* inertia;                  /* (batch, n_blocks, 3)  [m, m, J]                     */
  const double* damping;                  /* (batch, n_blocks, 3)  per-DOF viscous coefficient   */
  const double* void_angle0;              /* (batch, n_bonds, 2) undeformed void angles or NULL  */
  const double* contact;                  /* (batch, 3) min_angle, cutoff_angle, k_contact / NULL */
  const double* fn_params;                /* (batch, n_fns, DFX_FN_PARAMS)                       */
  const double* block_centroids;          /* (batch, n_blocks, 2): DFX_CONTACT_DISTANCE only, else NULL */
} dfx_params;

/* Gradient of  L = sum(fields_bar * fields)  with respect to everything in dfx_params and the
 * initial state.  Any pointer may be NULL (that gradient is then not accumulated). */
typedef struct dfx_grads {
  double* centroid_node_vectors;          /* (batch, n_blocks, n_npb, 2)                         */
  double* reference_vector;               /* (batch, n_bonds, 2)                                 */
  double* k_bond;                         /* (batch, n_bonds, 3)                                 */
  double* inertia;                        /* (batch, n_blocks, 3)                                */
  double* damping;                        /* (batch, n_blocks, 3)                                */
  double* void_angle0;                    /* (batch, n_bonds, 2)                                 */
  double* contact;                        /* (batch, 3)                                          */
  double* fn_params;                      /* (batch, n_fns, DFX_FN_PARAMS)                       */
  double* state0;                         /* (batch, 2, n_blocks, 3)                             */
  double* block_centroids;                /* (batch, n_blocks, 2): non-zero with DFX_CONTACT_DISTANCE only */
} dfx_grads;

typedef struct dfx_stats {
  int64_t steps;                          /* RK steps taken per member                           */
  int64_t rhs_evals;                      /* RHS evaluations per member (forward)                */
  int64_t launches;                       /* kernel launches issued                              */
  double kernel_ms;                       /* device time of the integration loop (HIP events)    */
  double stage_kernel_us;                 /* mean duration of one stage-kernel launch incl. gap  */
  int64_t streams;                        /* member groups integrated concurrently (one HIP stream each); reverse sweep at the segments level
                                             with the re-run of the next piece beside the reverse stages of this one: 2 */
  int64_t stage_checkpoint;               /* 1: the forward pass also kept the stage accelerations of every step, so the
                                             reverse sweep runs without recompute launches (chosen when it fits in HBM) */
  int64_t checkpoint_records;             /* 1: the forward pass kept EVERY stage record of every step (56 s B per unit and step): the
                                             reverse launches read them directly, nothing is rebuilt or recomputed (richest level,
                                             taken when it fits; then stage_checkpoint = 0);
                                             2: "segments": nothing but the outputs was kept, the reverse sweep re-runs one output
                                             interval at a time with the records of that interval only */
  int64_t tile_kernels;                   /* which builds of the stage kernels this call launched: 0 the generic slot kernels; 2 their
                                             per-stage builds (stage index and common parameter shape compiled in: launches that fill the
                                             chip; DFX_STAGE_BUILDS=0 switches them off); 1 the tile kernels (every
                                             ligament evaluated once on lattice tiles, DFX_TILE=1: opt-in); 3 the persistent stage
                                             loop (one launch per segment of <= 256 steps, stage records handed between neighbouring waves: solves
                                             whose launches do not fill the chip; DFX_PERSIST=0 switches it off) */
} dfx_stats;

typedef struct dfx_handle dfx_handle;

int dfx_create(const dfx_problem* problem, dfx_handle** out);
int dfx_destroy(dfx_handle* h);
const char* dfx_last_error(const dfx_handle* h);  /* h may be NULL: error of the last failed create */

int dfx_set_params(dfx_handle* h, const dfx_params* params);

/* Optional: pre-allocate device buffers for solves of up to `max_steps` RK steps and `max_timepoints`
 * outputs (trajectory checkpoint only when keep_trajectory != 0), so that later dfx_forward / dfx_adjoint
 * calls neither allocate nor re-instantiate their hipGraphs. */
int dfx_reserve(dfx_handle* h, int64_t max_steps, int32_t max_timepoints, int32_t keep_trajectory);

/* Let `h` keep its trajectory checkpoint in the buffers of `with` (same device) instead of its own: for handles whose solves never
 * overlap in time -- the engines of a multi-input objective (problems/quads_focusing_multi_input.py:66-86), each running forward +
 * reverse before the next starts.  One allocation instead of one per input, and room for a richer checkpoint level.  A reverse sweep
 * on a handle whose checkpoint has meanwhile been overwritten by another handle's forward pass fails with an error. */
int dfx_share_checkpoint(dfx_handle* h, dfx_handle* with);

/* Integrate from timepoints[0] with `steps_per_interval` equal RK steps between consecutive
 * timepoints.  state0: (batch, 2, n_blocks, 3), or NULL: every member starts at rest (the reference's problems all do,
 * problems/quads_focusing.py:300); fields: (batch, T, 2, n_blocks, 3) or NULL (they stay on the device), row 0 is the
 * reconstructed initial state.  keep_trajectory != 0 checkpoints every step state in HBM so that
 * dfx_adjoint can run afterwards. */
int dfx_forward(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                int32_t steps_per_interval, int32_t keep_trajectory, double* fields, dfx_stats* stats);

/* Same on a caller-chosen grid: steps_per_interval[k] steps between timepoints[k] and timepoints[k+1]
 * (n_timepoints - 1 entries); step_times == NULL: equal steps inside every interval; otherwise step_times holds the
 * sum(steps_per_interval) + 1 step boundaries, strictly increasing, with the boundary that starts interval k equal to
 * timepoints[k].  This is how the grid chosen by the adaptive controller (dfx_adaptive_step_counts /
 * dfx_adaptive_step_times) is made differentiable: dfx_adjoint is the exact reverse of this solve. */
int dfx_forward_grid(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                     const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                     double* fields, dfx_stats* stats);

/* The same with one time grid PER MEMBER: timepoints (batch, n_timepoints), step_times (batch, sum(steps_per_interval) + 1), required.
 * The step counts are shared (all members advance in the same launches), the times are not: forward inputs whose static phases differ
 * in length (problems/quads_kinetic_energy_static_tuning.py:246-259, mapped over devices with pmap at :473-478) are ensemble members
 * of one call.  dfx_adjoint reverses it like any other fixed-grid solve. */
int dfx_forward_grid_members(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                             const int32_t* steps_per_interval, const double* step_times, int32_t keep_trajectory,
                             double* fields, dfx_stats* stats);

/* The reference's own integrator semantics (jax.experimental.ode.odeint 0.4.8 called at dynamics.py:166): adaptive
 * Dormand-Prince 5(4), RMS error norm over the free (q, v) components with tolerance atol + rtol*max(|y0|,|y1|),
 * (state0 == NULL: every member starts at rest, as in dfx_forward / dfx_forward_grid)
 * step factor min(10, max(0.9 ratio^-1/5, 1 | 0.2)) applied on accept and reject, Hairer initial step, quartic dense
 * output at `timepoints` (steps are not clipped to output times).  Every member controls its own step.  Forward only:
 * the reverse sweep needs the fixed grid of dfx_forward.  stats->steps = accepted steps (max over members),
 * stats->rhs_evals = RHS evaluations (max over members). */
int dfx_forward_adaptive(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                         double rtol, double atol, int64_t max_attempts, double* fields, dfx_stats* stats);

/* The same solve, keeping what the reverse sweep needs (keep_trajectory != 0): the stage records of every ACCEPTED step (a rejected
 * attempt's records are overwritten in place), every member's own step boundaries, and for every output the step it was interpolated in.
 * dfx_adjoint / dfx_adjoint_kinetic / dfx_kinetic_value_and_grad[_device] then run the exact discrete adjoint of THIS solve with its step
 * sizes frozen -- the outputs' cotangents enter through the quartic dense output (an output at relative position r of step n adds g to
 * lambda_n and h_n B_j(r) g to the cotangent of stage slope j; the FSAL slope's share joins the first slope of the next step) -- so that
 * value_and_grad of the reference's default call, jit(value_and_grad(objective)) over odeint at dynamics.py:166, is ONE forward pass and
 * ONE reverse sweep here, and the gradient is the derivative of exactly the fields that were returned (neither the accept / reject
 * decisions nor the step sizes are differentiated; nor does the reference's continuous adjoint, jax.experimental.ode._odeint_rev).
 * fields may be NULL (they stay on the device).  keep_trajectory == 0: dfx_forward_adaptive. */
int dfx_forward_adaptive_keep(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                              double rtol, double atol, int64_t max_attempts, int32_t keep_trajectory, double* fields, dfx_stats* stats);

/* Failure isolation.  In the reference a member of an ensemble that diverges -- the list of forward problems of
 * problems/quads_focusing_multi_input.py:66-77, the pmap of quads_kinetic_energy_static_tuning.py:473-478 -- yields NaN for itself only.
 * dfx_member_status: status of every member after the last forward pass, (batch,): 0 ok, 1 non-finite state / error estimate, 2 step size
 * underflow, 3 step budget exceeded.  dfx_set_failure_policy(h, 1): such a member no longer fails the call (return 3 / 4) -- the call
 * returns 0, the member's outputs are NaN from the point of failure on (adaptive: from row 1 on), its status says why, the other members are
 * untouched, and a reverse sweep leaves NaN (fixed grid) or zeros (adaptive) in its gradients.  Default: 0, the call fails. */
int dfx_member_status(dfx_handle* h, int32_t* status);
int dfx_set_failure_policy(dfx_handle* h, int32_t isolate);

/* Accepted steps of the last dfx_forward_adaptive per member and output interval: counts (batch, n_timepoints - 1);
 * a step is counted in the interval that contains its start. */
int dfx_adaptive_step_counts(dfx_handle* h, int32_t* counts);

/* End times of the steps member `member` accepted in the last dfx_forward_adaptive: writes min(*n, capacity) values
 * into times and the number of accepted steps into *n (at most 2^20 steps per member are recorded). */
int dfx_adaptive_step_times(dfx_handle* h, int32_t member, double* times, int64_t capacity, int64_t* n);

/* Reverse sweep over the checkpointed trajectory of the last dfx_forward(keep_trajectory=1).
 * fields_bar: (batch, T, 2, n_blocks, 3) cotangent of `fields`. */
int dfx_adjoint(dfx_handle* h, const double* fields_bar, dfx_grads* grads, dfx_stats* stats);

/* Device-resident variants for benchmarking: the forward keeps the (T, ...) fields on the device and
 * the cotangent is the target-kinetic-energy objective  sum_t sum_{b in target} m_bd v_bd^2 / 2
 * (energy.py:494-499, problems/quads_focusing.py:447-467), evaluated on the device. */
int dfx_objective_kinetic(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, double* objective);
int dfx_adjoint_kinetic(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, dfx_grads* grads,
                        dfx_stats* stats);

/* The same objective and gradient in one call and without host-side copies: what jit(value_and_grad(objective)) returns at
 * problems/quads_focusing.py:565.  `objective` (batch,) may be NULL.  Non-NULL entries of `want` select the gradients; `views`
 * receives pointers to them in LIBRARY-OWNED (pinned) memory, laid out as in dfx_grads, valid until the next call on the handle.
 * The accumulators are re-laid-out on the device, so the host does no scatter work. */
int dfx_kinetic_value_and_grad(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, double* objective,
                               const dfx_grads* want, dfx_grads* views, dfx_stats* stats);

/* The same call with the gradients left where the reverse sweep accumulated them: `device_views` receives DEVICE pointers (HBM of the
 * handle's GPU, dfx_grads layout, valid until the next call on the handle) -- what jit(value_and_grad(objective)) hands back in the
 * reference: device arrays, nothing crosses PCIe but the (batch,) objective.  Available for centroid_node_vectors, void_angle0,
 * inertia, damping, state0, block_centroids on lattices without extra ligaments; asking for an entry the library assembles on the host
 * (reference_vector, k_bond, contact, fn_params) returns 1.  dfx_download copies n doubles behind such a pointer to host memory
 * (through the pinned staging area).  A device-side consumer (the design map, an optimiser update, dfx_reduce_grads) reads them in place. */
int dfx_kinetic_value_and_grad_device(dfx_handle* h, const int32_t* target_blocks, int32_t n_target, double* objective,
                                      const dfx_grads* want, dfx_grads* device_views, dfx_stats* stats);
int dfx_download(dfx_handle* h, double* dst, const double* device_src, int64_t n);

/* Forward solve + objective + reverse sweep in ONE call -- jit(value_and_grad(objective))(design) at problems/quads_focusing.py:565 is one
 * XLA program too.  Equal to dfx_forward_grid(keep_trajectory = 1, fields = NULL) followed by dfx_kinetic_value_and_grad[_device]
 * (device_views != 0: `views` receives device pointers) bit for bit; the host does not wait for the forward pass before it enqueues the
 * sweep (one synchronisation and one round trip through the caller less: ~0.15 ms of a 5 ms job).  A non-finite forward state is
 * reported at the end (return 3, as dfx_forward; the sweep has then run on that state and its outputs are meaningless).
 * steps_per_interval: (n_timepoints - 1,); state0 NULL = at rest. */
int dfx_forward_kinetic_value_and_grad(dfx_handle* h, const double* state0, const double* timepoints, int32_t n_timepoints,
                                       const int32_t* steps_per_interval, const int32_t* target_blocks, int32_t n_target,
                                       double* objective, const dfx_grads* want, dfx_grads* views, int32_t device_views,
                                       dfx_stats* forward_stats, dfx_stats* adjoint_stats);

/* Post-processing of the last forward solve on its device-resident history (problems/quads_focusing.py:319-372 with
 * energy.py:522-534): strain energies of every ligament 1/2 k (strain |l0|)^2 for the axial, shear and bending strain of
 * the NONLINEAR kinematics, (batch, T, n_bonds) each, and the kinetic energy of every block sum_d m_d v_d^2 / 2,
 * (batch, T, n_blocks).  Any pointer may be NULL. */
int dfx_response_data(dfx_handle* h, double* strain_energy_stretch, double* strain_energy_shear, double* strain_energy_bending,
                      double* kinetic_energy);

/* Test hooks: one RHS evaluation and its vector-Jacobian product on full-DOF arrays.
 * y, dy, lam, y_bar: (batch, 2, n_blocks, 3).  Constrained DOFs of y are ignored (they follow the
 * driving functions); their dy / y_bar entries are returned as 0. */
int dfx_rhs(dfx_handle* h, const double* y, double t, double* dy);
int dfx_rhs_vjp(dfx_handle* h, const double* y, double t, const double* lam, double* y_bar, dfx_grads* grads);

/* Potential energy of a full-DOF configuration, (batch, n_blocks, 3) -> (batch,)  (test hook) */
int dfx_energy(dfx_handle* h, const double* u, double* energy);

/* Test hook: polls after which a wave of the persistent stage loop gives up waiting for a neighbour's record (0: the default, seconds).
 * A tiny value makes a launch give up at once, which exercises what happens when a workgroup of such a launch is not resident (another
 * process on the device): the handle latches onto one launch per stage and the solve is run again that way, in the same process. */
int dfx_test_set_spin_limit(dfx_handle* h, int32_t polls);

int dfx_device_count(void);
const char* dfx_version(void);

/* Layout of the five public structs AS THIS LIBRARY WAS COMPILED: for dfx_special, dfx_problem, dfx_params, dfx_grads, dfx_stats in
 * that order, sizeof followed by offsetof of every field in declaration order -- 5 + 4 + 16 + 9 + 10 + 9 = 53 int32 values.  A binding
 * that mirrors the structs by hand (difflexmm_amd/_binding.py flattens the ControlParams tree of utils.py:48-163 into them) compares
 * them with its own before the first call (tests/test_abi.py).  Writes min(n, 53) values, returns 53. */
int dfx_abi_layout(int32_t* out, int32_t n);

/* ---- design -> geometry on the host, fused (the per-evaluation prologue of the design loop) -------------------------------------------
 * The lattice maps of the reference (geometry.py:607-952: QuadGeometry / KagomeGeometry -- every node of every block = a static base vector
 * + one row of the design), the polygon pass (geometry.py:71-127), compute_inertia (geometry.py:144-160) and the undeformed void angles
 * (energy.py:204-219 + geometry.py:181-253 at rest) in one loop over the blocks; dfx_design_vjp is the cotangent of all of it.  Plain host
 * code (no handle, no device): `design` (batch, n_design, 2) holds the design arrays of a lattice flattened and concatenated, `gather[b*n_npb+k]`
 * the row node k of block b takes.  Outputs / cotangents may be NULL where noted.  Equal to difflexmm_amd/geometry.py (NumPy) to 1e-13. */
typedef struct dfx_design_map {
  int32_t n_blocks, n_npb, n_bonds, n_design;
  const double* base;                     /* (n_blocks, n_npb, 2) node vectors of the zero design                  */
  const int32_t* gather;                  /* (n_blocks, n_npb)    row of `design` added to each node                */
  const double* ref_points;               /* (n_blocks, 2)        lattice points the centroids are measured from    */
  const int32_t* bonds;                   /* (n_bonds, 2) node ids, or NULL (no void angles)                        */
} dfx_design_map;
int dfx_design_forward(const dfx_design_map* map, const double* design, int32_t batch, double density,
                       double* block_centroids /* (batch, n_blocks, 2) or NULL */, double* centroid_node_vectors /* (batch, n_blocks, n_npb, 2) */,
                       double* inertia /* (batch, n_blocks, 3) or NULL */, double* void_angle0 /* (batch, n_bonds, 2) or NULL */);
int dfx_design_vjp(const dfx_design_map* map, const double* design, int32_t batch, double density,
                   const double* centroid_node_vectors_bar, const double* block_centroids_bar /* or NULL */, const double* inertia_bar /* or NULL */,
                   const double* void_angle0_bar /* or NULL */, double* design_bar /* (batch, n_design, 2) */);

/* ---- multi-GPU: one process per GPU, independent members per rank, ONE collective per evaluation -------------
 * Replaces the reference's pmap over forward inputs + host sum (problems/quads_kinetic_energy_static_tuning.py:454-478) and the
 * sequential list of forward problems of problems/quads_focusing_multi_input.py:66-86.  RCCL over xGMI inside the library;
 * all buffers are HOST arrays (payloads are 8 B per member / a few KB of shared-design gradient: latency-bound).
 * Rank 0 creates the unique id; the launcher hands it to the other ranks (file / environment). */
#define DFX_COMM_UID_BYTES 128
enum { DFX_REDUCE_SUM = 0, DFX_REDUCE_MAX = 1, DFX_REDUCE_MIN = 2 };
typedef struct dfx_comm dfx_comm;
int dfx_comm_unique_id(char* uid128);
int dfx_comm_init(int32_t rank, int32_t nranks, const char* uid128, int32_t device, dfx_comm** out);
int dfx_comm_destroy(dfx_comm* c);
/* RCCL version the process runs / the engine was compiled against (major*10000 + minor*100 + patch) */
int dfx_comm_rccl_version(int32_t* runtime, int32_t* compiled);
int dfx_comm_rank(const dfx_comm* c);
int dfx_comm_size(const dfx_comm* c);
/* all[r * n_local + i] = local[i] of rank r, on every rank (one ncclAllGather) */
int dfx_gather_objectives(dfx_comm* c, const double* local, int32_t n_local, double* all);
/* sum over ranks, in place (one ncclAllReduce): gradients w.r.t. a design every rank shares */
int dfx_reduce_grads(dfx_comm* c, double* inout, int64_t n);
int dfx_comm_allreduce(dfx_comm* c, double* inout, int64_t n, int32_t op /* DFX_REDUCE_* */);
int dfx_comm_barrier(dfx_comm* c);
const char* dfx_comm_last_error(void);

/* device helpers (free / total HBM of a device, wait for everything queued on it) */
int dfx_mem_info(int32_t device, int64_t* free_bytes, int64_t* total_bytes);
int dfx_device_synchronize(int32_t device);

#ifdef __cplusplus
}
#endif

#ifdef DFX_ABI_LAYOUT_IMPL   /* the body of dfx_abi_layout, compiled into each library that implements this header */
#include <stddef.h>
static inline int dfxabi_fill(int32_t* out, int32_t n) {
  const int32_t v[] = {
      (int32_t)sizeof(dfx_special), (int32_t)offsetof(dfx_special, block), (int32_t)offsetof(dfx_special, con_mask),
      (int32_t)offsetof(dfx_special, con_coef), (int32_t)offsetof(dfx_special, load_coef),
      (int32_t)sizeof(dfx_problem), (int32_t)offsetof(dfx_problem, n_blocks), (int32_t)offsetof(dfx_problem, n_npb),
      (int32_t)offsetof(dfx_problem, n_bonds), (int32_t)offsetof(dfx_problem, bonds), (int32_t)offsetof(dfx_problem, bond_model),
      (int32_t)offsetof(dfx_problem, contact), (int32_t)offsetof(dfx_problem, n_special), (int32_t)offsetof(dfx_problem, special),
      (int32_t)offsetof(dfx_problem, n_fns), (int32_t)offsetof(dfx_problem, fn_type), (int32_t)offsetof(dfx_problem, batch),
      (int32_t)offsetof(dfx_problem, tableau), (int32_t)offsetof(dfx_problem, device), (int32_t)offsetof(dfx_problem, fn_table_n),
      (int32_t)offsetof(dfx_problem, fn_table), (int32_t)offsetof(dfx_problem, streams),
      (int32_t)sizeof(dfx_params), (int32_t)offsetof(dfx_params, centroid_node_vectors), (int32_t)offsetof(dfx_params, reference_vector),
      (int32_t)offsetof(dfx_params, k_bond), (int32_t)offsetof(dfx_params, inertia), (int32_t)offsetof(dfx_params, damping),
      (int32_t)offsetof(dfx_params, void_angle0), (int32_t)offsetof(dfx_params, contact), (int32_t)offsetof(dfx_params, fn_params),
      (int32_t)offsetof(dfx_params, block_centroids),
      (int32_t)sizeof(dfx_grads), (int32_t)offsetof(dfx_grads, centroid_node_vectors), (int32_t)offsetof(dfx_grads, reference_vector),
      (int32_t)offsetof(dfx_grads, k_bond), (int32_t)offsetof(dfx_grads, inertia), (int32_t)offsetof(dfx_grads, damping),
      (int32_t)offsetof(dfx_grads, void_angle0), (int32_t)offsetof(dfx_grads, contact), (int32_t)offsetof(dfx_grads, fn_params),
      (int32_t)offsetof(dfx_grads, state0), (int32_t)offsetof(dfx_grads, block_centroids),
      (int32_t)sizeof(dfx_stats), (int32_t)offsetof(dfx_stats, steps), (int32_t)offsetof(dfx_stats, rhs_evals),
      (int32_t)offsetof(dfx_stats, launches), (int32_t)offsetof(dfx_stats, kernel_ms), (int32_t)offsetof(dfx_stats, stage_kernel_us),
      (int32_t)offsetof(dfx_stats, streams), (int32_t)offsetof(dfx_stats, stage_checkpoint), (int32_t)offsetof(dfx_stats, checkpoint_records),
      (int32_t)offsetof(dfx_stats, tile_kernels)};
  const int32_t total = (int32_t)(sizeof(v) / sizeof(v[0]));
  for (int32_t i = 0; i < total && i < n; ++i) out[i] = v[i];
  return total;
}
#endif
#endif /* DFX_H */
