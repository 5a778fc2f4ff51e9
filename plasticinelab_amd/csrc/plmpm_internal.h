// Shared by the translation units of libplmpm.so: the engine object, the launch / error macros and the device view
// (Dev<T>) of an engine.  plmpm_capi.hip: object life cycle, state I/O, the hot path and its phase-split form, halos,
// profiling | plmpm_kinematics.hip: actions, the serial kinematics chain and its adjoint, primitive queries |
// plmpm_loss.hip: the loss and its adjoint | plmpm_migrate.hip: particle migration between z-slabs | plmpm_peer.hip: the
// device-side halo exchange (peer writes) and the slab substep loops that use it.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/plmpm.h"
#include "../../include/plmpm_tools.h"       // measurement / diagnostics / test hooks: same library, not the boundary
#include "plmpm_kernels.h"

// plmpm_sort.hip
extern "C" size_t plmpm_sort_temp_bytes(int n);
extern "C" size_t plmpm_scan_temp_bytes(size_t n);
extern "C" int plmpm_exclusive_scan(void* tmp, size_t bytes, const unsigned* in, unsigned* out, size_t n, void* stream);
extern "C" int plmpm_sort_pairs(void* tmp, size_t bytes, const unsigned* kin, unsigned* kout, const int* vin, int* vout, int n, int key_bits,
                                void* stream);

using namespace plb;

// last error of the calling thread (plmpm_last_error); defined in plmpm_capi.hip
int plmpm_fail(const char* fmt, ...);
#define fail(...) plmpm_fail(__VA_ARGS__)
#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define REQUIRE(cond, ...) do { if (!(cond)) return fail(__VA_ARGS__); } while (0)

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// loss scalar slots (doubles)
enum { LS_DENSITY = 0, LS_SDF = 1, LS_MAXGM = 2, LS_DOT = 3, LS_SUMGM = 4, LS_MIND = 8, LS_DNORM = 16, LS_COUNT = 32 };

struct plmpm_sim {
    bool mats_uniform = false, mats_filled = false;   // set_materials: all particles alike / device arrays written at least once
    double mats_u[3] = {0, 0, 0};                     // the one (mu, lam, yield stress) of a uniform body: travels in the kernel arguments
    bool det = false;                           // cfg.deterministic: integer-limb accumulation (plmpm_kernels.h: det_add)
    long long* det_grid = nullptr;              //   [8][G] limbs of the grid scatters
    long long* det_small = nullptr;             //   [LS_COUNT + kMaxPrim * 8][2] limbs of the loss scalars / loss pose adjoints
    plmpm_config cfg;
    plmpm_primitive prims[PLMPM_MAX_PRIMITIVES];
    int gwg = 1, gwg_log2 = 0, fs = 1, nflag = 1;   // grid workgroups (a power of two), flags per workgroup, flag slots = gwg * fs >= nblk
    int N, Npad, n, nblk, P, F, act_total;      // N: rows of storage epoch 0; Npad: padded row capacity of a frame
    int go[3], nbw[3];                          // grid window: origin node (multiple of 4) and extent in 4^3 blocks
    int act_ofs[PLMPM_MAX_PRIMITIVES + 1];
    size_t G, Gfull, tsz, frame_bytes;          // G: nodes of the window (what is allocated), Gfull = n^3
    std::vector<int> epochN;                    // rows per storage epoch (multi-GPU ranks gain / lose particles by migration)
    hipStream_t stream = nullptr;
    bool bound = false;
    plmpm_workspace ws;
    // device pointers
    char *state = nullptr, *adjw = nullptr, *gridw = nullptr, *miscw = nullptr;
    char* adj[2];
    char *mu, *lam, *ys;
    int* perm_d;
    char *grid_in, *grid_out, *grid_out_adj, *grid_in_adj;
    int* flags;
    char *loss_gm, *loss_td, *loss_ts;
    double *ppos, *prot, *ppos_a, *prot_a, *pv, *pw, *pv_a, *pw_a, *act, *act_a, *lscal, *staging;
    double *pgap, *pgap_a, *pgv, *pgv_a;           // Chopsticks gap trajectory, gap velocity and adjoints: [(F+1)][P]
    int* err_d = nullptr;
    // multi-GPU: pose adjoints produced by this rank's nodes/particles accumulate in *_l, get summed over
    // ranks by the host and are then merged into the global ppos_a/prot_a the kinematics chain reads
    bool dist = false;
    int interior_fwd = -1, interior_bwd = -1;      // frame whose interior grid blocks plmpm_grid_interior / _grad_gather_interior already did
    HaloIn halo_in[3];                // per halo field: where the neighbours' copies of the exchanged block planes arrive
    double target_outside = 0.0;      // sum of the target density over owned nodes outside the grid window (|0 - t| terms)
    // particle migration between z-slabs (plmpm_migrate_*): per storage epoch the global particle ids, the materials,
    // the map new slot -> old slot (or -1 - arrival index) and the old slots that left (down list, then up list)
    int *gid_store = nullptr, *mig_src = nullptr, *mig_leave = nullptr, *mig_dest = nullptr, *mig_cnt = nullptr, *iota = nullptr;
    char* mats_store = nullptr;
    double* mig_send[2] = {nullptr, nullptr};
    int mig_max_rows = 0, sort_cap = 0;
    std::vector<int32_t> ids0;            // global ids of the epoch-0 rows in caller order (plmpm_set_ids)
    struct MigInfo { int parent = 0, nout[2] = {0, 0}, nin[2] = {0, 0}; };
    std::vector<MigInfo> mig;             // per epoch
    int mig_pending_frame = -1, mig_pending_out[2] = {0, 0};
    int next_epoch = 1;
    int g2p_deferred = -1;            // slab path: frame whose g2p waits to run fused with the next frame's p2g
    double *ppos_l = nullptr, *prot_l = nullptr, *pgap_l = nullptr;
    // host state
    std::vector<int32_t> perm;
    double softness = 0.0;
    double w_sdf = 10, w_density = 10, w_contact = 1;
    int soft_contact = 0;
    bool have_target = false;
    double target_max = 0, target_sum = 0;
    int adj_frame[2] = {-1, -1};
    // per-frame grid_m / grid_v_in store (cfg.store_grid)
    bool store = false;
    char* gstore = nullptr;      // grid_m / grid_v_in per frame (SoA, 4 comps)
    char* vstore = nullptr;      // grid_v_out per frame (AoS T4)
    int* fstore = nullptr;
    int* contact = nullptr;      // [0] = n, [1..n]: blocks whose pose adjoints k_grid_op_grad left to the k_p2g_grad launch
    char* ptab = nullptr;            // [(F+1)][kMaxPrim] PrimT<T>: the primitives per substep, for the grid kernels (k_build_prims)
    int* tiles = nullptr;        // per-frame stencil boxes of the particle workgroups: [(F+1)][Npad/256][8]
    // Per-env-step storage order ("epochs").  Epoch 0 is the order chosen at reset (perm_d).  With cfg.resort_steps,
    // plmpm_step re-sorts the step's first frame along the Hilbert curve before it starts (epoch = step index); the
    // frames a step writes are in its epoch.  The reverse sweep converts the adjoint frame between epochs at the
    // step boundaries and reads v of the (re-sorted) boundary frame from the copy kept in the old order.
    bool resort = false;
    bool prof_no_resort = false;          // plmpm_set_resort(0): keep the current order (segment-checkpointed runs)
    int n_epochs = 1;
    int* perm_store = nullptr;            // [n_epochs - 1][Npad]: storage slot -> caller index, epochs 1..
    char* vend = nullptr;                 // [n_epochs][3 Npad] T: v of the frame that epoch e re-sorted, in the OLD order
    double* mats_master = nullptr;        // mu, lam, ys in caller order
    bool have_mats = false;
    unsigned *skey[2] = {nullptr, nullptr};
    int* sidx[2] = {nullptr, nullptr};
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    unsigned* cell_hist = nullptr;              // counting-sort flavour of the single-GPU re-sort: particles per cell (curve order)
    size_t cell_bins = 0;                       //   0: grid too fine for it, the radix sort is used
    char* frame_tmp = nullptr;
    std::vector<int> frame_epoch;
    int mats_epoch = 0;
    int adj_epoch[2] = {0, 0};
    int steps_since_sort = 0;             // env steps since the order of the current frames was chosen
    size_t gstride = 0;
    std::vector<char> dirty;          // frame f holds a scattered grid that has not been consumed/cleared
    // Device-side halo exchange (plmpm_peer.hip): per halo field and face a receive area in fine-grained device memory
    // that the neighbour on that face has mapped (IPC) -- [arrival counter | 2 x ncomp x count scalars] -- and the
    // neighbour's area for this rank's planes, mapped here.  seq counts the exchanges of a field; half seq & 1 is written.
    struct PeerField {
        int n = 0, ba[2] = {0, 0}, bb[2] = {0, 0};
        size_t count[2] = {0, 0};
        char *local[2] = {nullptr, nullptr}, *remote[2] = {nullptr, nullptr};
        unsigned seq = 0;
    };
    PeerField peer[3];
    unsigned* peer_done = nullptr;        // device: [0] workgroups of the running exchange that have finished their copies, [8] tag of the
                                          //   last exchange whose arrivals the poller saw, [16] device copy of the status word
    unsigned peer_tag = 0;                // tags of the exchanges of this engine (all fields): unique, never 0
    int* peer_status = nullptr;           // pinned host: 0, or field << 16 | face << 8 | 1 of an arrival that timed out
    std::vector<void*> peer_allocs, peer_mapped;
    bool peer_uncached = false;           // the receive areas are hipDeviceMallocUncached (else fine-grained)
    float peer_spoil = 1.0f;              // test hook (plmpm_debug_peer_spoil)
    // optional per-kernel timing with HIP events on the launch stream (plmpm_profile_*)
    bool prof = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<int, int>> ev_used;     // (kernel id, index of the start event)
    size_t ev_next = 0;
};

enum KernelId { K_P2G = 0, K_GRID_OP, K_G2P, K_P2G_RE, K_GRID_OP_RE, K_G2P_GRAD, K_GRID_OP_GRAD, K_P2G_GRAD, K_CLEAR, K_G2P_P2G,
                K_HALO_XCHG, K_GRID_OP_X, K_GRID_OP_GRAD_X, K_COUNT };
static const char* kKernelNames[K_COUNT] = {"p2g", "grid_op", "g2p", "p2g_recompute", "grid_op_recompute",
                                            "g2p_grad", "grid_op_grad", "p2g_grad", "clear_active", "g2p_p2g",
                                            "halo_exchange", "xchg+grid_op", "xchg+grid_op_grad"};

static void prof_begin(plmpm_sim* s, int id) {
    if (!s->prof) return;
    if (s->ev_next + 2 > s->ev_pool.size()) {
        size_t old = s->ev_pool.size();
        s->ev_pool.resize(old + 1024);
        for (size_t i = old; i < s->ev_pool.size(); ++i) (void)hipEventCreate(&s->ev_pool[i]);
    }
    s->ev_used.push_back({id, (int)s->ev_next});
    (void)hipEventRecord(s->ev_pool[s->ev_next], s->stream);
    s->ev_next += 2;
}
static void prof_end(plmpm_sim* s) {
    if (!s->prof) return;
    (void)hipEventRecord(s->ev_pool[s->ev_used.back().second + 1], s->stream);
}
// Experiment hook (profiles/r04_notes.md), compiled only with -DPLB_DYNLDS_HOOK=1: the environment variable
// PLB_DYNLDS="kernel:bytes,..." adds dynamic LDS to a kernel's launches, which lowers the workgroups a CU can hold -- how much
// does each kernel's time depend on its occupancy?  The product build launches with 0 bytes and never reads the environment.
#ifndef PLB_DYNLDS_HOOK
#define PLB_DYNLDS_HOOK 0
#endif
#if PLB_DYNLDS_HOOK
static inline unsigned dyn_lds(int id) {
    static unsigned tab[K_COUNT];
    static bool init = false;
    if (!init) {
        init = true;
        if (const char* e = getenv("PLB_DYNLDS")) {
            std::string v(e);
            size_t pos = 0;
            while (pos < v.size()) {
                size_t c = v.find(',', pos);
                if (c == std::string::npos) c = v.size();
                const std::string item = v.substr(pos, c - pos);
                const size_t q = item.find(':');
                if (q != std::string::npos)
                    for (int k = 0; k < K_COUNT; ++k)
                        if (item.substr(0, q) == kKernelNames[k]) tab[k] = (unsigned)atoi(item.c_str() + q + 1);
                pos = c + 1;
            }
        }
    }
    return tab[id];
}
#else
static inline unsigned dyn_lds(int) { return 0; }
#endif
#define LAUNCHG_CLEAR(s, D) LAUNCH(s, K_CLEAR, (k_clear_active<T>), dim3(nblocks_grid(s)), D)
#define LAUNCHB(s, id, kern, grid, block, ...)                                             \
    do {                                                                                   \
        if (dim3(grid).x == 0) break;            /* a slab rank may hold no particles for a while */ \
        prof_begin(s, id);                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(block), dyn_lds(id), (s)->stream, __VA_ARGS__); \
        prof_end(s);                                                                       \
    } while (0)
#define LAUNCH(s, id, kern, grid, ...) LAUNCHB(s, id, kern, grid, kBlock, __VA_ARGS__)

// Scatter launches.  Deterministic engines (cfg.deterministic) run the DET instantiation -- integer-limb accumulation,
// plmpm_kernels.h -- followed by the sweep that turns the limbs into the T sums the next kernel reads.
#define DET_RESOLVE(s, d0, d1, d2, d3) \
    hipLaunchKernelGGL((k_det_resolve<T>), dim3(1024), dim3(256), 0, (s)->stream, (s)->det_grid, (s)->G, d0, d1, d2, d3)
#define LAUNCH_P2G(s, id, WF, D, f)                                                                              \
    do {                                                                                                         \
        if ((s)->det) {                                                                                          \
            LAUNCH(s, id, (k_p2g<T, WF, true>), dim3(nblocks_particles(s, f)), D, f);                            \
            DET_RESOLVE(s, D.gin[0], D.gin[1], D.gin[2], D.gin[3]);                                              \
        } else LAUNCH(s, id, (k_p2g<T, WF>), dim3(nblocks_particles(s, f)), D, f);                               \
    } while (0)
#define LAUNCH_G2P_P2G(s, D, f, vprev)                                                                           \
    do {                                                                                                         \
        if ((s)->det) {                                                                                          \
            LAUNCH(s, K_G2P_P2G, (k_g2p_p2g<T, true>), dim3(nblocks_particles(s, f)), D, f, vprev);                \
            DET_RESOLVE(s, D.gin[0], D.gin[1], D.gin[2], D.gin[3]);                                              \
        } else LAUNCH(s, K_G2P_P2G, (k_g2p_p2g<T>), dim3(nblocks_particles(s, f)), D, f, vprev);                   \
    } while (0)
// (k_g2p_grad's workgroup 0 resets the contact list k_grid_op_grad(f) is about to fill.  A slab rank that holds no particles
// launches nothing here, while its grid kernel still appends the blocks of the exchanged planes in which a NEIGHBOUR's mass
// touches a manipulator: the list is then reset from the host side of the stream -- stale entries would be processed again by
// the pose workgroups of every later reverse substep and could outgrow the list's nblk + 1 words)
#define LAUNCH_G2P_GRAD(s, D, f, src, dst, vnext)                                                                \
    do {                                                                                                         \
        if (nblocks_particles(s, f) == 0) (void)hipMemsetAsync((s)->contact, 0, sizeof(int), (s)->stream);       \
        if ((s)->det) {                                                                                          \
            LAUNCH(s, K_G2P_GRAD, (k_g2p_grad<T, true>), dim3(nblocks_particles(s, f)), D, f, src, dst, vnext);  \
            DET_RESOLVE(s, D.goa[0], D.goa[1], D.goa[2], (T*)nullptr);                                           \
        } else LAUNCH(s, K_G2P_GRAD, (k_g2p_grad<T>), dim3(nblocks_particles(s, f)), D, f, src, dst, vnext);     \
    } while (0)
// p2g.grad with the pose adjoints of the blocks in contact: spare workgroups of the same launch, or -- deterministic
// engines -- one wave walking the contact list in block order first
#define LAUNCH_P2G_GRAD(s, D, f, src, dst)                                                                       \
    do {                                                                                                         \
        if ((s)->det) {                                                                                          \
            hipLaunchKernelGGL((k_pose_adjoint_det<T>), dim3(1), dim3(64), 0, (s)->stream, D, f);                \
            LAUNCH(s, K_P2G_GRAD, (k_p2g_grad<T>), dim3(nblocks_particles(s, f)), D, f, src, dst, 0);            \
        } else LAUNCH(s, K_P2G_GRAD, (k_p2g_grad<T>), dim3(nblocks_particles(s, f) + kPoseWG), D, f, src, dst, kPoseWG); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// frame >= 0 with the grid store on: that frame's own grid_in / flags; otherwise the shared scratch grid
template <class T> static Dev<T> make_dev(const plmpm_sim* s, int frame = -1) {
    Dev<T> D;
    const plmpm_config& c = s->cfg;
    double dx = 1.0 / c.n_grid;
    D.P.n = c.n_grid; D.P.dx = (T)dx; D.P.inv_dx = (T)c.n_grid; D.P.dt = (T)c.dt; D.P.p_mass = (T)c.p_mass;
    D.P.kappa = (T)(-c.dt * c.p_vol * 4.0 * (double)c.n_grid * (double)c.n_grid);
    for (int i = 0; i < 3; ++i) D.P.grav[i] = (T)(c.dt * c.gravity[i] * 30.0);
    D.P.x_hi = (T)(1.0 - 3.0 * dx);
    D.P.ground_friction = (T)c.ground_friction;
    D.P.svd_clamp = (T)c.svd_grad_clamp;
    D.P.softness = (T)s->softness;
    D.P.tie_first = c.minmax_tie != 0;
    const int epoch = frame >= 0 ? s->frame_epoch[frame] : 0;
    D.N = frame >= 0 ? s->epochN[epoch] : s->N; D.Npad = s->Npad; D.nprim = s->P;
    D.twg = s->Npad / kBlock;
    D.fgl = s->gwg_log2; D.fs = s->fs;
    for (int d = 0; d < 3; ++d) { D.go[d] = s->go[d]; D.rlo[d] = s->go[d]; D.rhi[d] = s->go[d] + 4 * s->nbw[d]; }
    D.nbx = s->nbw[0]; D.nby = s->nbw[1]; D.nbz = s->nbw[2];
    D.z0 = c.slab_z0; D.z1 = c.slab_z1;
    // interior slab faces: the neighbour only exchanges slab_halo node layers beyond the face
    if (c.slab_z0 > 0) D.rlo[2] = std::max(D.rlo[2], c.slab_z0 - c.slab_halo);
    if (c.slab_z1 < c.n_grid) D.rhi[2] = std::min(D.rhi[2], c.slab_z1 + c.slab_halo);
    D.err = s->err_d;
    D.frame_bytes = s->frame_bytes;
    D.state = s->state;
    D.adj[0] = (T*)s->adj[0]; D.adj[1] = (T*)s->adj[1];
    if (s->dist) {      // materials travel with the particles: one set per storage epoch
        T* m = (T*)(s->mats_store + (size_t)epoch * 3 * s->Npad * s->tsz);
        D.mu = m; D.lam = m + s->Npad; D.ys = m + 2 * (size_t)s->Npad;
    } else { D.mu = (T*)s->mu; D.lam = (T*)s->lam; D.ys = (T*)s->ys; }
    // one material for the whole body (the common case): the kernels take it from their arguments instead of loading 12 bytes
    // per particle -- p2g.grad follows its HBM bytes at ~4.7 TB/s (round 4: 30 MB fewer -> 6.4 us), so 6 MB are 1.3 us
#ifndef PLB_MATS_UNIFORM
#define PLB_MATS_UNIFORM 1
#endif
    D.mats_uniform = (PLB_MATS_UNIFORM && s->mats_uniform && !s->dist && s->have_mats) ? 1 : 0;
    for (int i = 0; i < 3; ++i) D.mat_u[i] = (T)s->mats_u[i];
    const bool framed = s->store && frame >= 0;
    char* gin_base = framed ? s->gstore + (size_t)frame * s->gstride : s->grid_in;
    for (int c = 0; c < 4; ++c) D.gin[c] = (T*)gin_base + (size_t)c * s->G;
    for (int c = 0; c < 3; ++c) D.goa[c] = (T*)s->grid_out_adj + (size_t)c * s->G;
    D.grid_out = (Vec4<T>*)(framed ? s->vstore + (size_t)frame * s->gstride : s->grid_out);
    D.grid_in_adj = (Vec4<T>*)s->grid_in_adj;
    D.flags = framed ? s->fstore + (size_t)frame * s->nflag : s->flags;
    D.tiles = s->tiles;
    D.contact = s->contact;
    D.ptab = (const PrimT<T>*)s->ptab;
    D.det = s->det_grid; D.det_stride = s->G;
    D.trace = (unsigned long long*)s->staging;      // profiling builds only (needs N * 24 * 8 >= 3 * 16384 * 128 bytes)
    D.ppos = s->ppos; D.prot = s->prot; D.pgap = s->pgap;
    D.ppos_a = s->dist ? s->ppos_l : s->ppos_a;
    D.prot_a = s->dist ? s->prot_l : s->prot_a;
    D.pgap_a = s->dist ? s->pgap_l : s->pgap_a;
    for (int i = 0; i < s->P; ++i) {
        D.prim[i].shape = s->prims[i].shape;
        D.prim[i].movable = s->prims[i].action_dim > 0;
        for (int k = 0; k < 3; ++k) D.prim[i].par[k] = s->prims[i].params[k];
        D.prim[i].friction = s->prims[i].friction;
    }
    return D;
}


#define DISPATCH(s, fn, ...) ((s)->cfg.dtype == PLMPM_F64 ? fn<double>(__VA_ARGS__) : fn<float>(__VA_ARGS__))
#define NEED_BOUND(s)                                                                                             \
    do {                                                                                                          \
        REQUIRE((s) && (s)->bound, "workspace not bound");                                                        \
        REQUIRE((s)->g2p_deferred < 0, "frame %d's g2p is deferred: call plmpm_p2g(frame + 1, chain = 1) next", (s)->g2p_deferred); \
    } while (0)
#define NEED_FRAME(s, f) REQUIRE((f) >= 0 && (f) <= (s)->F, "frame %d out of range [0,%d]", (f), (s)->F)

static inline int nblocks_particles(const plmpm_sim* s, int frame) { return (s->epochN[s->frame_epoch[frame]] + kBlock - 1) / kBlock; }
// ---------------------------------------------------------------------------------------------
// storage slot -> host row of state / gradient I/O.  Single GPU: the caller's particle index in every epoch.  Slab
// engines: caller order only in epoch 0; once particles have migrated the rows of a frame are its storage order
// (plmpm_get_ids names them)
static int* perm_of(const plmpm_sim* s, int epoch) {
    if (epoch <= 0) return s->perm_d;
    return s->dist ? s->iota : s->perm_store + (size_t)(epoch - 1) * s->Npad;
}

template <class T> __device__ __forceinline__ PrimT<T> prim_at(const Dev<T>& D, int q, int f) {
    PrimT<T> p;
    p.shape = D.prim[q].shape; p.movable = D.prim[q].movable; p.friction = (T)D.prim[q].friction;
    for (int i = 0; i < 3; ++i) { p.par[i] = D.prim[q].par[i]; p.pos[i] = p.pos1[i] = D.ppos[((size_t)f * D.nprim + q) * 3 + i]; }
    if (p.shape == SHAPE_CHOPSTICKS) p.par[2] = D.pgap[(size_t)f * D.nprim + q];
    p.rb = prim_bounding_radius(p.shape, p.par);
    for (int i = 0; i < 4; ++i) p.rot[i] = p.rot1[i] = D.prot[((size_t)f * D.nprim + q) * 4 + i];
    return p;
}

// ---- per-env-step re-sort ------------------------------------------------------------------------------------------
// Hilbert key of every storage slot of frame f (padding slots sort last and, the sort being stable, stay in place)
__device__ __forceinline__ unsigned hilbert_key_dev(unsigned x0, unsigned x1, unsigned x2, int bits) {
    unsigned X[3] = {x0, x1, x2};
    const unsigned M = 1u << (bits - 1);
    for (unsigned Q = M; Q > 1; Q >>= 1) {
        const unsigned P = Q - 1;
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { unsigned t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    for (int i = 1; i < 3; ++i) X[i] ^= X[i - 1];
    unsigned t = 0;
    for (unsigned Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
    for (int i = 0; i < 3; ++i) X[i] ^= t;
    unsigned h = 0;
    for (int bit = bits - 1; bit >= 0; --bit)
        for (int i = 0; i < 3; ++i) h = (h << 1) | ((X[i] >> bit) & 1u);
    return h;
}

// cross-unit entry points
extern "C" int plmpm_launch_fk(plmpm_sim* s, int first, int n);                      // plmpm_kinematics.hip
extern "C" void plmpm_launch_fk_grad(plmpm_sim* s, int first, int n, int step);
int plmpm_halo_field(plmpm_sim* s, int field, int frame, char** base, int* ncomp);   // plmpm_capi.hip: base of a halo field's SoA components
int plmpm_convert_adjoint(plmpm_sim* s, int which, int from, int to);                 // plmpm_capi.hip: adjoint frame `which` into another storage epoch
// plmpm_capi.hip: plmpm_grid_g2p / plmpm_grad_gather with the field's device-side exchange folded into the grid kernel (X built
// by plmpm_peer.hip: peer_prepare); X->n == 0 (no neighbour) falls back to the plain launch
int plmpm_grid_g2p_xchg(plmpm_sim* s, int frame, int chain, const PeerXchg* X);
int plmpm_grad_gather_xchg(plmpm_sim* s, int frame, const PeerXchg* X);
#ifndef PLB_PEER_FUSED_DEFAULT
#define PLB_PEER_FUSED_DEFAULT 0       // measured (round 5): no wall-clock gain for a thin rank of 8 at config 3 (91.0 against 89.3 us per substep)
#endif

