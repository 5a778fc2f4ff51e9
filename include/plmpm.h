/* plmpm.h -- C ABI of the MI355X-native differentiable MPM engine.
 *
 * Drop-in boundary.  The reference has no FFI: its
 * "operator API" is the Python class surface of
 *   plb.engine.mpm_simulator.MPMSimulator   (/root/reference/plb/engine/mpm_simulator.py)
 *   plb.engine.primitive.Primitives         (/root/reference/plb/engine/primitive/primitives.py:262-320)
 *   plb.engine.losses.Loss                  (/root/reference/plb/engine/losses/loss.py)
 * Each entry point below names the reference method(s) it stands in for
 * (file:line).  A host-language binding only needs this header: plain
 * pointers and sizes, int status returns (0 = ok, <0 = error; text via
 * plmpm_last_error), no exceptions across the boundary, no torch types.
 *
 * Memory: the caller owns all device memory.  plmpm_workspace_bytes reports
 * what a simulator needs, the caller allocates it (the Python facade uses
 * torch ROCm tensors) and hands the base pointers to plmpm_bind_workspace.
 * All kernels are enqueued on the HIP stream given to plmpm_set_stream
 * (default: the null stream); calls that return host data synchronise that
 * stream.  A handle is not re-entrant.
 */
#ifndef PLMPM_H
#define PLMPM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLMPM_MAX_PRIMITIVES 8
#define PLMPM_MAX_ACTION_DIM 7

enum plmpm_dtype { PLMPM_F32 = 0, PLMPM_F64 = 1 };
enum plmpm_shape {
    PLMPM_SPHERE = 0, PLMPM_CAPSULE = 1, PLMPM_CYLINDER = 2, PLMPM_TORUS = 3, PLMPM_BOX = 4,
    PLMPM_CHOPSTICKS = 5      /* primitives.py:83-154: two Capsules `gap` apart hanging below the pose */
};

/* forward_kinematics flavour: base class (primive_base.py:117-121, world-frame rotation),
 * RollingPin (primitives.py:66-80: roll about own axis, turn about world y, move in y; shape = Capsule) or
 * Chopsticks (primitives.py:94-98: body-frame rotation + gap[f+1] = max(gap[f] - gap_vel[f], minimal_gap);
 * 7-dim action, shape = PLMPM_CHOPSTICKS) */
enum plmpm_kinematics { PLMPM_KIN_DEFAULT = 0, PLMPM_KIN_ROLLINGPIN = 1, PLMPM_KIN_CHOPSTICKS = 2 };

/* Simulator constants; mirrors MPMSimulator.__init__ (mpm_simulator.py:6-51). */
typedef struct plmpm_config {
    int32_t dtype;            /* plmpm_dtype: arithmetic type of the hot path            */
    int32_t n_grid;           /* :19   int(128 * quality * 0.5)                          */
    int32_t n_particles;      /* :18                                                     */
    int32_t max_frames;       /* :33   cfg.max_steps (+1 frames are stored)              */
    int32_t substeps;         /* :34   int(2e-3 // dt)                                   */
    int32_t n_primitives;     /* :13                                                     */
    double dt;                /* :22                                                     */
    double p_vol;             /* :23   (dx/2)^2                                          */
    double p_mass;            /* :24                                                     */
    double gravity[3];        /* :50   cfg.gravity                                       */
    double ground_friction;   /* :11                                                     */
    double svd_grad_clamp;    /* :143-151 clamp of backward_svd; 1e-6 = reference, 0 = exact derivative */
    /* z-slab owned by this rank for multi-GPU runs: nodes z in [slab_z0, slab_z1); 0,n_grid = whole grid */
    int32_t slab_z0, slab_z1;
    /* 1: keep grid_m / grid_v_in and grid_v_out of every frame resident (max_frames x 32 n^3 bytes at fp32) so
     * that substep_grad skips the forward recompute of mpm_simulator.py:262-268 (same results, less work);
     * 0: recompute like the reference (use this for very large grids / copy-mode-only use). */
    int32_t store_grid;
    /* node layers beyond [slab_z0, slab_z1) that this rank's particles may still reach (fixed ownership, no
     * migration yet): a particle whose stencil leaves [slab_z0 - slab_halo, slab_z1 + slab_halo) raises
     * PLMPM_ERR_HALO in plmpm_check_error.  0 on a single GPU. */
    int32_t slab_halo;
    /* R > 0: every R-th env step, plmpm_step first re-sorts the particles of the step's first frame along the Hilbert
     * curve of their cells (device radix sort + one gather pass, ~0.2 ms) -- the north star's "particles sorted by
     * cell", kept true as the material flows; the storage order then changes every R env steps, state / gradient
     * I/O stays in caller order and the reverse sweep converts the adjoint frame at those boundaries.  Single-GPU
     * engines only (ignored for slabs).  0: the order chosen at reset is kept for the whole episode. */
    int32_t resort_steps;
    /* Grid window: only the box of nodes [grid_lo[d], grid_hi[d]) per axis -- rounded outwards to whole 4^3 blocks -- is
     * allocated, stored per frame and swept by the grid kernels (all zero = the whole n^3 grid, the reference's
     * layout).  Results are those of the full grid as long as every particle's stencil stays inside the window; one
     * that leaves it raises PLMPM_ERR_HALO in plmpm_check_error (its accesses are clamped into the window, so the
     * run stays memory-safe).  This is what lets a 512^3 rank keep per-frame grids: the per-frame store costs
     * 32 B x window nodes instead of 32 B x n^3. */
    int32_t grid_lo[3], grid_hi[3];
    /* rows every particle frame has room for (>= n_particles; 0 = n_particles).  Slab engines gain and lose
     * particles by migration (plmpm_migrate_*), so their frames are sized with head-room. */
    int32_t particle_capacity;
    /* 1: bit-reproducible runs.  Every sum that several waves contribute to (grid_m / grid_v_in, grid_v_out.grad, the
     * loss grid and scalars, pose adjoints) is accumulated in two 64-bit integer limbs instead of floating-point
     * atomics, so the result does not depend on the arrival order; the same rollout then gives the same bits every
     * time, on every launch schedule and across re-sorts.  Slower (measured in DESIGN.md); contributions must stay below 2^38 in
     * magnitude.  Slab engines: the rows that migrate leave in slot order, so a fixed rank layout reproduces too (the
     * host's reductions across ranks must be order-fixed as well: gloo and RCCL rings are).  0: fp atomics. */
    int32_t deterministic;
    /* Two pieces of Taichi autodiff semantics that the reference's gradients depend on and that cannot be verified without
     * a Taichi 0.7.14 installation are switches, so that a Taichi-generated golden vector that disagrees
     * costs a flag, not a kernel edit (oracle/plb_oracle.py has the same two flags):
     *   contact_min_adjoint  0: ti.atomic_min in the hard contact loss (loss.py:123-128) differentiated like an add --
     *                           min_dist's adjoint goes to EVERY particle (the assumed Taichi 0.7.x behaviour, default);
     *                        1: it goes to the particle(s) attaining the minimum (the mathematical derivative);
     *   minmax_tie           0: the adjoint of max(a, b) / min(a, b) goes to the SECOND operand on an exact tie (assumed
     *                           Taichi behaviour, default); 1: to the first.  Applies to every max / min on the
     *                           differentiated path: position clamps (mpm_simulator.py:242, primive_base.py:119), the
     *                           0.05 floor of the return mapping (:126), friction and normal-velocity clamps of collide
     *                           (primive_base.py:100-103), the influence clamp, shape SDFs, the contact loss's max(sdf, 0). */
    int32_t contact_min_adjoint;
    int32_t minmax_tie;
    /* persistent workgroups of the grid kernels (grid_op, grid_op.grad; rounded down to a power of two, at most 512 and at most
     * blocks / 4); 0 = 512.  Slab engines that are to run the fused exchange + grid kernels of the device-side halo exchange
     * (PLMPM_PEER_FUSED=1) must be created with 1 .. 256 here: those launches wait for the neighbours, so every grid workgroup of
     * every rank on a GPU must be resident at once -- at most 256 / ranks-per-GPU; an engine created with 0 keeps the exchange
     * kernels whatever the variable says (plmpm_peer_fused reports the form in use). */
    int32_t grid_workgroups;
} plmpm_config;

/* One rigid manipulator; mirrors Primitive.default_config + per-shape params
 * (primive_base.py:209-224, primitives.py:30-34,56-61,185-190,215-220,253-257). */
typedef struct plmpm_primitive {
    int32_t shape;                        /* plmpm_shape                                              */
    int32_t action_dim;                   /* cfg.action.dim (0 = static)                              */
    double params[3];                     /* Sphere: radius | Capsule: h,r | Cylinder: h,r | Torus: tx,ty | Box: size
                                           * | Chopsticks: h, r, minimal_gap                          */
    double friction;                      /* primive_base.py:162                                      */
    double action_scale[PLMPM_MAX_ACTION_DIM];
    double lower_bound[3], upper_bound[3];/* xyz_limit, primive_base.py:160                           */
    int32_t kinematics;                   /* plmpm_kinematics: which forward_kinematics the primitive uses */
    int32_t reserved;
} plmpm_primitive;

/* Sizes (bytes) of the four device workspaces of one simulator. */
typedef struct plmpm_workspace {
    size_t state_bytes;   /* particle frames: (max_frames+1) x 24 x Npad scalars  (x,v,C,F-I; SoA)  */
    size_t adjoint_bytes; /* two ping-pong adjoint frames + material arrays                        */
    size_t grid_bytes;    /* grid_m/grid_v_in, grid_v_out and their adjoints, block flags, loss grids */
    size_t misc_bytes;    /* primitive trajectories, action buffers, their adjoints, loss scalars, staging */
} plmpm_workspace;

typedef struct plmpm_sim* plmpm_handle;

const char* plmpm_last_error(void);
int plmpm_version(void);

/* ---- lifetime ------------------------------------------------------------------------------ */
/* MPMSimulator.__init__ + Primitives.__init__ (mpm_simulator.py:6-51, primitives.py:263-279) */
int plmpm_create(const plmpm_config* cfg, const plmpm_primitive* prims, plmpm_handle* out);
int plmpm_destroy(plmpm_handle h);
int plmpm_workspace_bytes(plmpm_handle h, plmpm_workspace* out);
int plmpm_bind_workspace(plmpm_handle h, void* state, void* adjoint, void* grid, void* misc);
int plmpm_set_stream(plmpm_handle h, void* hip_stream);

/* ---- state I/O (host float64 arrays in the reference's AoS layout) -------------------------- */
/* MPMSimulator.initialize (mpm_simulator.py:53-57): per-particle mu, lam, yield_stress */
int plmpm_set_materials(plmpm_handle h, const double* mu, const double* lam, const double* yield_stress);
/* setframe / set_state (mpm_simulator.py:292-300,325-328).  x,v:(N,3) F,C:(N,3,3).
 * resort != 0 recomputes the cell-sorted storage order from x (do this at episode reset). */
int plmpm_set_frame(plmpm_handle h, int frame, const double* x, const double* v, const double* F,
                    const double* C, int resort);
/* readframe / get_state / get_x / get_v (mpm_simulator.py:282-290,314-323,343-363); any pointer may be NULL */
int plmpm_get_frame(plmpm_handle h, int frame, double* x, double* v, double* F, double* C);
/* copyframe (mpm_simulator.py:302-312) incl. primitive poses */
int plmpm_copy_frame(plmpm_handle h, int source, int target);
/* Primitive.set_state / get_state (primive_base.py:129-151; Chopsticks primitives.py:135-146):
 * 8 doubles = position(3) + rotation(4) + gap(1); the gap slot is carried but unused by the other shapes */
int plmpm_set_primitive_state(plmpm_handle h, int prim, int frame, const double* state8);
int plmpm_get_primitive_state(plmpm_handle h, int prim, int frame, double* state8);
/* switch the per-env-step re-sort of cfg.resort_steps off / on at run time (segment-checkpointed runs keep one order) */
int plmpm_set_resort(plmpm_handle h, int on);
/* Primitives.set_softness (primitives.py:303-305) */
int plmpm_set_softness(plmpm_handle h, double softness);
/* Primitive.sdf (primive_base.py:57-60; Sphere primitives.py:22-24, ...): signed distance of n host points (n,3) to
 * primitive `prim` at its pose of `frame` */
int plmpm_primitive_sdf(plmpm_handle h, int prim, int frame, const double* points, int n, double* out);
/* Primitive.set_velocity (primive_base.py:184-192): refill v, w of env step `step`'s frames from action_buffer[step] */
int plmpm_set_velocity(plmpm_handle h, int prim, int step, int n_substeps);

/* ---- actions ----------------------------------------------------------------------------- */
/* Primitives.set_action (primitives.py:289-293) -> per primitive no_grad_set_action_kernel +
 * set_velocity (primive_base.py:166-198): clip to [-1,1], store in action_buffer[step], fill
 * v,w for frames [step*n_substeps, (step+1)*n_substeps), and run forward_kinematics
 * (primive_base.py:117-121) over those frames (the pose chain does not depend on particles). */
int plmpm_set_action(plmpm_handle h, int step, int n_substeps, const double* action);
/* Primitives.get_grad (primitives.py:295-301): out is (n_steps, sum action_dim) row major */
int plmpm_get_action_grad(plmpm_handle h, int n_steps, double* out);

/* ---- the hot path ------------------------------------------------------------------------- */
/* MPMSimulator.substep (mpm_simulator.py:245-257): frame f -> f+1 */
int plmpm_substep(plmpm_handle h, int frame);
/* fused loop of MPMSimulator.step (mpm_simulator.py:372-373) */
int plmpm_step(plmpm_handle h, int first_frame, int n_substeps);
/* what ti.Tape.__enter__ does for this path: zero every adjoint, frame `last_frame` becomes the
 * adjoint seed frame (solver.py:36) */
int plmpm_grad_begin(plmpm_handle h, int last_frame);
/* MPMSimulator.substep_grad (mpm_simulator.py:260-278): consumes adjoint of frame f+1, produces frame f */
int plmpm_substep_grad(plmpm_handle h, int frame);
/* reverse of plmpm_step: substep_grad for frames first+n-1 .. first, then forward_kinematics.grad and
 * set_velocity.grad for env step `step` (primive_base.py:117-121,184-192) */
int plmpm_step_grad(plmpm_handle h, int first_frame, int n_substeps, int step);
/* segment-checkpointed backward (plb/optimizer/long_term_gradient.ipynb, copy_and_clear): after the forward of the
 * earlier segment has been re-run, hand the adjoint of `from_frame` (first frame of the later segment) over to
 * `to_frame` (last frame of the earlier one); pose adjoints move along, all other primitive / action adjoints clear */
int plmpm_segment_carry(plmpm_handle h, int from_frame, int to_frame);
/* add a host-provided cotangent to the current adjoint of `frame` (x,v:(N,3) F,C:(N,3,3), any may be NULL);
 * used by tests and by callers that differentiate their own loss */
int plmpm_add_frame_grad(plmpm_handle h, int frame, const double* xa, const double* va, const double* Fa,
                         const double* Ca);
/* read back the adjoint currently held for `frame` (x.grad[f] etc. in the reference) */
int plmpm_get_frame_grad(plmpm_handle h, int frame, double* xa, double* va, double* Fa, double* Ca);
/* primitive pose adjoints position.grad[f], rotation.grad[f], gap.grad[f] (8 doubles) */
int plmpm_get_primitive_grad(plmpm_handle h, int prim, int frame, double* grad8);
/* position.grad[f] / rotation.grad[f] / gap.grad[f] += grad8: lets an outer program differentiate through an
 * observation of the manipulator poses (the Taichi MLP's input_primitives kernel, plb/engine/nn/mlp.py:75-84) */
int plmpm_add_primitive_grad(plmpm_handle h, int prim, int frame, const double* grad8);

/* ---- loss (Loss, loss.py) ------------------------------------------------------------------ */
/* Loss.load_target_density + update_target (loss.py:46-57,81-106): density is (n,n,n) float64, [i][j][k] */
int plmpm_loss_set_target(plmpm_handle h, const double* density);
/* Loss.set_weights (loss.py:68-72) */
int plmpm_loss_set_weights(plmpm_handle h, double sdf, double density, double contact, int soft_contact);
/* Loss.compute_loss_kernel (loss.py:186-208) at `frame`; out6 = {loss increment, sdf, density, contact,
 * iou (loss.py:239-254), reserved}.  Does not touch adjoints. */
int plmpm_loss_forward(plmpm_handle h, int frame, double* out6);
/* Loss.compute_loss_kernel_grad (loss.py:210-237) with d(total)/d(loss) = 1: adds into the adjoint of
 * `frame` (particles) and into the pose adjoints of the movable primitives */
int plmpm_loss_backward(plmpm_handle h, int frame);
/* compute_grid_m_kernel (mpm_simulator.py:382-392) -> host (n,n,n) float64 */
int plmpm_get_grid_mass(plmpm_handle h, int frame, double* out);
/* target_sdf as computed by plmpm_loss_set_target, host (n,n,n) float64 (nodes outside the grid window read 0) */
int plmpm_loss_get_target_sdf(plmpm_handle h, double* out);
/* Loss.min_dist / Loss.dist_norm per primitive after the last loss evaluation (loss.py:116-135); either may be NULL */
int plmpm_loss_contact_scalars(plmpm_handle h, double* min_dist, double* dist_norm);

/* ---- multi-GPU building blocks (z-slab decomposition; the host side exchanges halos with RCCL) ------------
 * The forward substep is p2g | halo sum-exchange of grid_m,grid_v_in | grid_op + g2p, the reverse
 * g2p.grad | halo sum-exchange of grid_v_out.grad | grid_op.grad + p2g.grad.  These split
 * plmpm_substep / plmpm_substep_grad at the exchange points.  Requires store_grid = 1.
 *
 * Halos are ZERO-COPY: slab faces sit on multiples of 4, the grid is stored in 4^3 blocks with the z block index
 * slowest, so the block planes around a face are one contiguous range per SoA component.  The host sends those
 * ranges straight out of the grid arrays (plmpm_halo_region) and receives the neighbour's copy into a buffer it
 * registers once (plmpm_halo_set_recv); grid_op / grid_op.grad add the received values on first touch.  No pack or
 * unpack kernels: a fwd+bwd substep is the same 5 launches as on one GPU. */
enum plmpm_halo_field { PLMPM_HALO_GRID_IN = 0 /* 4 comps */, PLMPM_HALO_GRID_OUT_ADJ = 1 /* 3 comps */,
                        PLMPM_HALO_LOSS_MASS = 2 /* 1 comp */ };
#define PLMPM_ERR_HALO 1
int plmpm_fk(plmpm_handle h, int first_frame, int n_substeps);              /* forward_kinematics chain only */
/* chain = 1: g2p(frame - 1), left pending by plmpm_grid_g2p(frame - 1, chain = 1), runs fused with p2g(frame) in one
 * kernel (as plmpm_step does on one GPU) */
int plmpm_p2g(plmpm_handle h, int frame, int chain);
/* grid_op(frame) -- adding the halo planes registered for PLMPM_HALO_GRID_IN -- then g2p(frame); chain = 1 leaves the
 * g2p to the next plmpm_p2g(frame + 1, chain = 1), which must be the next call.  The last substep of an env step
 * passes chain = 0. */
int plmpm_grid_g2p(plmpm_handle h, int frame, int chain);
/* Overlap of the exchange with grid work (optional).  Blocks outside the exchanged planes do not need the neighbours'
 * values: plmpm_grid_interior runs grid_op on them while the halos are still in flight; the plmpm_grid_g2p(frame) that
 * follows -- after the halos have arrived -- then only does the blocks of the exchanged planes and g2p.
 * plmpm_grad_gather_interior / plmpm_grad_gather are the same split of grid_op.grad. */
int plmpm_grid_interior(plmpm_handle h, int frame);
int plmpm_grad_gather_interior(plmpm_handle h, int frame);
int plmpm_grad_scatter(plmpm_handle h, int frame);                          /* g2p.grad */
int plmpm_grad_gather(plmpm_handle h, int frame);                           /* grid_op.grad (+ halo planes) + p2g.grad + clear */
int plmpm_chain_grad(plmpm_handle h, int first_frame, int n_substeps, int step);   /* fk.grad + set_velocity.grad */
/* origin node and extent in 4^3 blocks of the allocated grid window (cfg.grid_lo / grid_hi after rounding) */
int plmpm_grid_window(plmpm_handle h, int32_t* origin3, int32_t* blocks3);
/* component `comp` of block planes [bz_a, bz_b) (absolute block-plane indices, node z / 4) of a halo field: one
 * contiguous device range of `count` scalars of the engine's type.  Both sides of a face must use the same xy window. */
int plmpm_halo_region(plmpm_handle h, int field, int frame, int comp, int bz_a, int bz_b, void** dev_ptr, size_t* count);
/* where the neighbours' copies arrive: recv[i] holds [ncomp][count] scalars for block planes [bz_a[i], bz_b[i]);
 * n_faces <= 2; n_faces = 0 unregisters.  The pointers must stay valid while the engine runs phases. */
int plmpm_halo_set_recv(plmpm_handle h, int field, int n_faces, const int* bz_a, const int* bz_b, void* const* recv);
/* PLMPM_HALO_LOSS_MASS only: add the registered buffers into the loss mass grid (the substep fields are added by
 * grid_op / grid_op.grad themselves) */
int plmpm_halo_apply(plmpm_handle h, int field, int frame);

/* ---- device-side halo exchange (peer writes) and the native slab substep loops -------------------------------------
 * The host-driven exchange above costs a communication-library call per neighbour, component and substep.  This form
 * takes the host out of the substep loop: every rank allocates, per halo field and face, a RECEIVE AREA in fine-grained
 * device memory (plmpm_peer_area_bytes, plmpm_peer_alloc -> pointer + 64-byte IPC handle), sends the handle to the
 * neighbour on that face by whatever means the host has (once), maps the neighbour's area (plmpm_peer_open) and
 * registers both (plmpm_halo_peer_setup).  plmpm_halo_peer_exchange is then ONE kernel on the engine's stream: copy this
 * rank's exchange planes into the neighbours' areas (xGMI peer writes on a multi-GPU node), publish an arrival counter
 * behind them, wait (bounded: PLMPM_PEER_TIMEOUT seconds, default 20) for the neighbours' counters; the grid kernels
 * that follow add the received planes as before.  plmpm_slab_step / plmpm_slab_step_grad are the substep loops of one
 * env step for a slab rank -- mpm_simulator.py:245-278, 365-376 with one exchange per grid phase -- and only enqueue.
 * All ranks must run the same sequence of exchanges per field.  An arrival that timed out is reported by the next
 * exchange call and by plmpm_peer_status (0 = fine). */
int plmpm_peer_area_bytes(plmpm_handle h, int field, int bz_a, int bz_b, size_t* bytes);
int plmpm_peer_alloc(plmpm_handle h, size_t bytes, void** dev_ptr, void* ipc_handle64);   /* owned by the engine, zeroed */
int plmpm_peer_open(plmpm_handle h, const void* ipc_handle64, void** dev_ptr);            /* unmapped at plmpm_destroy */
/* local[i]: this rank's area for face i; remote[i]: the area the neighbour on face i allocated for ITS face towards this
 * rank, mapped here (a loop-back run may pass its own areas).  n_faces = 0 unregisters. */
int plmpm_halo_peer_setup(plmpm_handle h, int field, int n_faces, const int* bz_a, const int* bz_b, void* const* local, void* const* remote);
int plmpm_halo_peer_exchange(plmpm_handle h, int field, int frame);
int plmpm_peer_status(plmpm_handle h, int* status);
/* collective re-synchronisation in two phases, every rank calling each and meeting the others at a host barrier behind it:
 * phase 0 drains this rank's enqueued exchange kernels (nobody can publish an old sequence number any more), phase 1 sets
 * counters, sequence numbers and the status word back to zero (after a timeout, or an exception between exchanges) */
int plmpm_halo_peer_reset(plmpm_handle h, int phase);
/* 1: receive areas in uncached device memory (hipDeviceMallocUncached), 0: fine-grained (PLMPM_PEER_MEM=finegrained, or refused) */
int plmpm_peer_memory_kind(plmpm_handle h, int* uncached);
/* First contact with the neighbours, collective (every rank calls it with the same non-zero token and then meets the others
 * at a host barrier): the token is stored through into a spare word of both neighbours' receive-area headers of `field` and
 * theirs is polled for with system-scope loads -- the exchange's own hand-off on the real areas, before a halo depends on it.
 * ok2[i] = 1: face i's token arrived within timeout_s; wait_us2[i]: how long the poll took.  Exchange counters untouched. */
int plmpm_peer_ping(plmpm_handle h, int field, unsigned token, double timeout_s, int* ok2, double* wait_us2);
/* 1: plmpm_slab_step / plmpm_slab_step_grad fold each exchange into the grid kernel that consumes it (send the owned blocks of
 * the exchanged planes | interior blocks | wait | blocks of the exchanged planes: one launch, the interior hides the arrival);
 * 0: exchange kernel + grid kernel (PLMPM_PEER_FUSED unset or 0 -- the build's default, -DPLB_PEER_FUSED_DEFAULT=0 -- or an engine
 * that was not created for it: only engines with plmpm_config.grid_workgroups in 1 .. 256 take the fused form, because every
 * grid workgroup of a fused launch waits for the neighbours and must be resident) */
int plmpm_peer_fused(plmpm_handle h, int* fused);
int plmpm_slab_step(plmpm_handle h, int first_frame, int n_substeps);         /* fk + n x (p2g | exchange | grid_op + g2p) */
int plmpm_slab_step_grad(plmpm_handle h, int first_frame, int n_substeps);    /* n x (g2p.grad | exchange | grid_op.grad + p2g.grad), in reverse */

/* ---- particle migration between z-slabs --------------------------------------------------------------
 * A rank owns the particles whose stencil centre node lies in its slab.  At the first frame of an env step:
 *   plmpm_migrate_begin  classifies the rows of `frame`, packs the leavers (28 doubles per row: global id, x, v, C,
 *                        F - I, mu, lam, yield stress) and returns their counts [down, up] and device buffers;
 *   (the host exchanges counts and rows with the two neighbours)
 *   plmpm_migrate_finish merges the arrivals, re-sorts everything along the Hilbert curve of the cells and makes the
 *                        result the frame's content in a NEW storage epoch (so this is also the slab engines' cell
 *                        re-sort; with no leavers and no arrivals it is exactly that).
 * The reverse sweep, when it reaches that frame: plmpm_migrate_adjoint_begin packs the adjoint rows (24 doubles) of the
 * particles that had arrived, to be sent back (send2 = rows to send down / up, recv2 = rows to expect);
 * plmpm_migrate_adjoint_finish takes the adjoints of the particles that had left and leaves the adjoint frame in the
 * order of the previous epoch.  State / gradient I/O of frames in epochs > 0 is in storage order; plmpm_get_ids names
 * the rows. */
int plmpm_set_ids(plmpm_handle h, const int32_t* ids);                      /* global ids of the epoch-0 rows (caller order) */
int plmpm_get_ids(plmpm_handle h, int frame, int32_t* ids);
/* Slab engines, segment-checkpointed rollouts (plb/optimizer/long_term_gradient.ipynb cells 2-4 on a population that
 * changes with every migration): plmpm_set_population re-enters the engine with a NEW set of rows -- call it, then
 * plmpm_set_ids, plmpm_set_frame(0, ..., resort = 1) and plmpm_set_materials with the rows of a checkpoint (what
 * plmpm_get_ids / plmpm_get_frame / plmpm_get_materials returned for the checkpointed frame). */
int plmpm_set_population(plmpm_handle h, int n_rows);
/* rows of the resident adjoint of `frame` (those of the storage epoch the adjoint is in: after plmpm_migrate_adjoint_* at a
 * frame that migrated, the epoch BEFORE the migration -- not the frame's own row count): what plmpm_get_frame_grad writes */
int plmpm_adjoint_rows(plmpm_handle h, int frame, int32_t* rows);
/* mu, lam, yield stress of the rows of `frame`, in the row order plmpm_get_frame uses for it */
int plmpm_get_materials(plmpm_handle h, int frame, double* mu, double* lam, double* yield_stress);
int plmpm_frame_info(plmpm_handle h, int frame, int32_t* count, int32_t* epoch, int32_t* adjoint_epoch /* -1: not resident */);
int plmpm_migrate_begin(plmpm_handle h, int frame, int32_t* out2, void** rows_down, void** rows_up);
int plmpm_migrate_finish(plmpm_handle h, int frame, int n_in_down, const void* rows_down, int n_in_up, const void* rows_up,
                         int32_t* new_count);
int plmpm_migrate_adjoint_begin(plmpm_handle h, int frame, int32_t* send2, int32_t* recv2, void** rows_down, void** rows_up);
int plmpm_migrate_adjoint_finish(plmpm_handle h, int frame, const void* rows_down, const void* rows_up);
/* the primitive pose adjoints (double) for the cross-rank sum */
int plmpm_pose_grad_region(plmpm_handle h, int first_frame, int n_frames, void** pos_adj, size_t* pos_count,
                           void** rot_adj, size_t* rot_count, void** gap_adj, size_t* gap_count);
int plmpm_action_grad_region(plmpm_handle h, void** dev_ptr, size_t* count);
/* loss in phases: scatter (then exchange PLMPM_HALO_LOSS_MASS), local partial sums over owned nodes /
 * particles, set globally reduced contact scalars, local adjoint.
 * partials (32 doubles): [0] density [1] sdf [2] max grid_m [3] sum m*target [4] sum m; [8+q] min_dist (hard: min,
 * soft: weighted sum), [16+q] dist_norm (soft).  phase 0: hard-min or soft normaliser; phase 1: soft weighted sum */
int plmpm_loss_scatter(plmpm_handle h, int frame);
int plmpm_loss_partials(plmpm_handle h, int frame, int phase, double* out32);
int plmpm_loss_set_globals(plmpm_handle h, const double* in32);
int plmpm_loss_finish(plmpm_handle h, const double* global32, double* out6);   /* host arithmetic of loss.py:137-162,252-254 */
int plmpm_loss_backward_local(plmpm_handle h, int frame);
int plmpm_check_error(plmpm_handle h, int* flags);

#ifdef __cplusplus
}
#endif
#endif /* PLMPM_H */
