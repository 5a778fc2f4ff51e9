/* libplmpm.so -- measurement, diagnostics and test hooks.
 *
 * NOT part of the drop-in boundary: include/plmpm.h is what a PlasticineLab maintainer binds.  The entry
 * points below live in the same library and serve bench.py, profiles/tools/ and tests/ only: per-kernel HIP-event timing,
 * kernel / env-step replay on the engine's current state (A/B of library builds on identical inputs), the measured HBM
 * roof, workgroup stencil boxes, active-node counts, build flags, and two test hooks.  None of them has a counterpart in
 * the reference (plb/engine/mpm_simulator.py has no instrumentation beyond plb/utils timers).
 */
#ifndef PLMPM_TOOLS_H
#define PLMPM_TOOLS_H

#include "plmpm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* what this build of the library was compiled with: bit 1 the elastic fast path (-DPLB_FAST=1), bit 2 the XCD-aware chunk map
 * (-DPLB_XCD_MAP=1), bit 3 particle arrays through buffer descriptors (-DPLB_BUFIO=1); bit 0 (the experimental engine variants of rounds 3-4)
 * is never set any more */
int plmpm_build_flags(void);

/* number of grid nodes with mass > 0 and number of active 4^3 blocks after the last forward substep */
int plmpm_grid_stats(plmpm_handle h, int frame, int64_t* active_nodes, int64_t* active_blocks);
/* diagnostics: the stencil bounding box (origin node x,y,z, extent x,y,z) of every 256-particle workgroup of a frame
 * that has been scattered, as the kernels stage it in LDS; out = int32[n_workgroups][6] (out may be NULL to query
 * n_workgroups).  A box of more than 1024 (fp32) / 512 (fp64) nodes takes the slow global-memory path. */
int plmpm_tile_boxes(plmpm_handle h, int frame, int32_t* out, int max_workgroups, int* n_workgroups);
/* per-kernel timing with HIP events recorded on the launch stream, around every hot-path kernel.
 * enable(1) starts collecting; read() synchronises, returns summed milliseconds and launch counts for
 * the plmpm_profile_kernel_count() kernel classes and resets the collection. */
int plmpm_profile_enable(plmpm_handle h, int on);
int plmpm_profile_kernel_count(void);
const char* plmpm_profile_kernel_name(int id);
int plmpm_profile_read(plmpm_handle h, double* total_ms, int64_t* launches);
/* profiling aid: launch one hot-path kernel `reps` times on the engine's current state, mean duration in microseconds
 * (kind 0: fused g2p(frame-1)+p2g(frame), 1: g2p.grad(frame), 2: p2g.grad(frame), 3: p2g(frame)).  The rollout is not
 * usable afterwards (the replays accumulate into the grids).  profiles/tools/replay_ab.py */
int plmpm_replay(plmpm_handle h, int kind, int frame, int reps, double* mean_us);
/* the same for the substep loop of a whole env step, frames [first, first + n): dir 0 forward, 1 reverse; graph 0: launched
 * eagerly `reps` times, 1: `reps` replays of one captured hipGraph (what the launch boundaries cost, on identical work) */
int plmpm_replay_step(plmpm_handle h, int graph, int dir, int first, int n, int reps, double* mean_us);
/* Measured HBM roof of the device the buffers live on: a 16-byte-per-lane copy src -> dst and a read-only sweep of
 * `bytes` bytes, best of `reps` runs, in GB/s of bytes moved (bench.py reports it next to the 8 TB/s spec). */
int plmpm_measure_hbm(void* src, void* dst, size_t bytes, int reps, void* hip_stream, double* copy_gbs, double* read_gbs);
/* storage order: perm[i] = original particle index stored at sorted slot i */
int plmpm_get_order(plmpm_handle h, int32_t* perm);

/* ---- test hooks ---------------------------------------------------------------------------- */
/* test hook: scale what this rank sends through face 0 (1 = off) -- a spoiled halo must be noticed by the transport check */
int plmpm_debug_peer_spoil(plmpm_handle h, double factor);
/* tuning aid: {error word, workgroups of the fused forward kernel that fell back to global atomics, workgroups on
 * the LDS-tile path, sum of their tile sizes in nodes} since the last call */
int plmpm_debug_counters(plmpm_handle h, int* out4);
/* test hook: the list of 4^3 blocks whose pose adjoints k_grid_op_grad left to the next k_p2g_grad launch.  seed >= 0 first
 * overwrites the list with `seed` stale entries (block 0); *count = entries listed now.  A reverse substep must leave only its
 * own entries behind, on a rank without particles too */
int plmpm_debug_contact(plmpm_handle h, int seed, int* count);

#ifdef __cplusplus
}
#endif
#endif /* PLMPM_TOOLS_H */
