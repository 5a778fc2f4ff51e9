// C-ABI implementation (include/plmpm.h) on top of the kernels in plmpm_kernels.h.
// Host code here only carves workspaces, moves host<->device state and sequences launches; every
// arithmetic step of the hot path runs in a HIP kernel.  There is no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/plmpm.h"
#include "plmpm_kernels_pk.h"

// plmpm_sort.hip
extern "C" size_t plmpm_sort_temp_bytes(int n);
extern "C" size_t plmpm_scan_temp_bytes(size_t n);
extern "C" int plmpm_exclusive_scan(void* tmp, size_t bytes, const unsigned* in, unsigned* out, size_t n, void* stream);
extern "C" int plmpm_sort_pairs(void* tmp, size_t bytes, const unsigned* kin, unsigned* kout, const int* vin, int* vout, int n, int key_bits,
                                void* stream);

using namespace plb;

static thread_local std::string g_err;
static int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}
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
                                 // (two lists of nblk + 1: fused-grid engines alternate between them, frame by frame)
    // Fused-grid engines (one GPU, grid store, not deterministic): grid_op / grid_op.grad are evaluated inside the tile
    // fills of the particle kernels (plmpm_kernels.h: fg_node_vout / fg_node_gadj) -- 1 launch per forward substep and 2
    // per reverse substep instead of 2 and 3.  The grids of the frame the last reverse substep finished with
    // (fg_pending) are cleared by the next g2p.grad, or by k_clear_boxes when something else comes first.
    bool fg = false;
    int fg_pending = -1;
    std::vector<char> vnear;         // frame f: grid_v_out / contact bit of the nodes near a primitive are in the frame's grid_v_out store
    // two particles per lane with packed fp32 arithmetic (plmpm_kernels_pk.h): fp32 engines, floating-point atomics
    bool pk = false;
    char* grid_out_adj2 = nullptr;   // second grid_v_out.grad buffer (frames alternate)
    char* ptab = nullptr;            // [(F+1)][kMaxPrim] PrimT<T>: the primitives per substep, for the fills (k_build_prims)
    int* contact_mark = nullptr;     // [nblk] stamp of the g2p.grad launch that last listed the block as in contact
    int contact_stamp = 0;
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
    // optional per-kernel timing with HIP events on the launch stream (plmpm_profile_*)
    bool prof = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<int, int>> ev_used;     // (kernel id, index of the start event)
    size_t ev_next = 0;
};

enum KernelId { K_P2G = 0, K_GRID_OP, K_G2P, K_P2G_RE, K_GRID_OP_RE, K_G2P_GRAD, K_GRID_OP_GRAD, K_P2G_GRAD, K_CLEAR, K_G2P_P2G,
                // fused-grid engines: the same particle kernels with grid_op / grid_op.grad evaluated in their tile fills
                K_FG_G2P, K_FG_G2P_P2G, K_FG_G2P_GRAD, K_FG_P2G_GRAD, K_COUNT };
static const char* kKernelNames[K_COUNT] = {"p2g", "grid_op", "g2p", "p2g_recompute", "grid_op_recompute",
                                            "g2p_grad", "grid_op_grad", "p2g_grad", "clear_active", "g2p_p2g",
                                            "gridop+g2p", "gridop+g2p_p2g", "gridop+g2p_grad", "gridop_grad+p2g_grad"};

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
#define LAUNCHG_CLEAR(s, D) LAUNCH(s, K_CLEAR, (k_clear_active<T>), dim3(nblocks_grid(s)), D)
#define LAUNCHB(s, id, kern, grid, block, ...)                                             \
    do {                                                                                   \
        if (dim3(grid).x == 0) break;            /* a slab rank may hold no particles for a while */ \
        prof_begin(s, id);                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(block), 0, (s)->stream, __VA_ARGS__);         \
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
        PrevGrid<T> pg_;                                                                                         \
        memset(&pg_, 0, sizeof pg_);                                                                             \
        pg_.vout = vprev;                                                                                        \
        if ((s)->det) {                                                                                          \
            LAUNCH(s, K_G2P_P2G, (k_g2p_p2g<T, true>), dim3(nblocks_particles(s, f)), D, f, pg_);                \
            DET_RESOLVE(s, D.gin[0], D.gin[1], D.gin[2], D.gin[3]);                                              \
        } else LAUNCH(s, K_G2P_P2G, (k_g2p_p2g<T>), dim3(nblocks_particles(s, f)), D, f, pg_);                   \
    } while (0)
#define LAUNCH_G2P_GRAD(s, D, f, src, dst, vnext)                                                                \
    do {                                                                                                         \
        ClearArgs<T> ca_;                                                                                        \
        memset(&ca_, 0, sizeof ca_);                                                                             \
        ca_.frame = -1;                                                                                          \
        if ((s)->det) {                                                                                          \
            LAUNCH(s, K_G2P_GRAD, (k_g2p_grad<T, true>), dim3(nblocks_particles(s, f)), D, f, src, dst, vnext, ca_);  \
            DET_RESOLVE(s, D.goa[0], D.goa[1], D.goa[2], (T*)nullptr);                                           \
        } else LAUNCH(s, K_G2P_GRAD, (k_g2p_grad<T>), dim3(nblocks_particles(s, f)), D, f, src, dst, vnext, ca_);     \
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
// fg: a launch of the fused-grid path -- the frame's parity picks the grid_v_out.grad buffer and the contact list
template <class T> static Dev<T> make_dev(const plmpm_sim* s, int frame = -1, bool fg = false) {
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
    const bool framed = s->store && frame >= 0;
    char* gin_base = framed ? s->gstore + (size_t)frame * s->gstride : s->grid_in;
    for (int c = 0; c < 4; ++c) D.gin[c] = (T*)gin_base + (size_t)c * s->G;
    {
        char* ga = s->grid_out_adj;
        char* gb = s->grid_out_adj2 ? s->grid_out_adj2 : s->grid_out_adj;
        if (fg && (frame & 1)) std::swap(ga, gb);
        for (int c = 0; c < 3; ++c) { D.goa[c] = (T*)ga + (size_t)c * s->G; D.goa_prev[c] = (T*)gb + (size_t)c * s->G; }
    }
    D.grid_out = (Vec4<T>*)(framed ? s->vstore + (size_t)frame * s->gstride : s->grid_out);
    D.grid_in_adj = (Vec4<T>*)s->grid_in_adj;
    D.flags = framed ? s->fstore + (size_t)frame * s->nflag : s->flags;
    D.tiles = s->tiles;
    D.contact = s->contact + ((fg && (frame & 1)) ? s->nblk + 1 : 0);
    D.contact_next = s->contact + ((fg && (frame & 1)) ? 0 : s->nblk + 1);
    D.contact_mark = s->contact_mark; D.stamp = s->contact_stamp;
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

// ---------------------------------------------------------------------------------------------
// small kernels: state I/O, primitive chains, loss
// staging layout (double, original particle order): x[N*3] v[N*3] F[N*9] C[N*9]
template <class T>
__global__ void k_unpack_frame(Dev<T> D, int f, const double* st, const int* perm, int has_x, int has_v, int has_F, int has_C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.Npad) return;
    double* X = frame_x_w(D, f);
    T* R = frame_r(D, f);
    const int N = D.N, Np = D.Npad;
    if (i >= N) {       // padding lanes: harmless values
        for (int d = 0; d < 3; ++d) { X[d * Np + i] = 0.5; R[d * Np + i] = T(0); }
        for (int d = 0; d < 18; ++d) R[(3 + d) * Np + i] = T(0);
        return;
    }
    int o = perm[i];
    const double *sx = st, *sv = st + (size_t)3 * N, *sF = st + (size_t)6 * N, *sC = st + (size_t)15 * N;
    if (has_x) for (int d = 0; d < 3; ++d) X[d * Np + i] = sx[(size_t)3 * o + d];
    if (has_v) for (int d = 0; d < 3; ++d) R[d * Np + i] = (T)sv[(size_t)3 * o + d];
    if (has_C) for (int d = 0; d < 9; ++d) R[(3 + d) * Np + i] = (T)sC[(size_t)9 * o + d];
    if (has_F) for (int d = 0; d < 9; ++d) R[(12 + d) * Np + i] = (T)(sF[(size_t)9 * o + d] - ((d % 4 == 0) ? 1.0 : 0.0));
}
template <class T>
__global__ void k_pack_frame(Dev<T> D, int f, double* st, const int* perm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.N) return;
    const double* X = frame_x(D, f);
    const T* R = frame_r(D, f);
    const int N = D.N, Np = D.Npad;
    int o = perm[i];
    double *sx = st, *sv = st + (size_t)3 * N, *sF = st + (size_t)6 * N, *sC = st + (size_t)15 * N;
    for (int d = 0; d < 3; ++d) { sx[(size_t)3 * o + d] = X[d * Np + i]; sv[(size_t)3 * o + d] = (double)R[d * Np + i]; }
    for (int d = 0; d < 9; ++d) {
        sC[(size_t)9 * o + d] = (double)R[(3 + d) * Np + i];
        sF[(size_t)9 * o + d] = (double)R[(12 + d) * Np + i] + ((d % 4 == 0) ? 1.0 : 0.0);
    }
}
// adjoint frame <-> staging (same staging layout: xa, va, Fa, Ca)
template <class T>
__global__ void k_adj_io(Dev<T> D, int which, double* st, const int* perm, int add_from_staging, int has_x, int has_v, int has_F, int has_C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.N) return;
    T* A = D.adj[which];
    const int N = D.N, Np = D.Npad;
    int o = perm[i];
    double *sx = st, *sv = st + (size_t)3 * N, *sF = st + (size_t)6 * N, *sC = st + (size_t)15 * N;
    if (add_from_staging) {
        if (has_x) for (int d = 0; d < 3; ++d) A[d * Np + i] += (T)sx[(size_t)3 * o + d];
        if (has_v) for (int d = 0; d < 3; ++d) A[(3 + d) * Np + i] += (T)sv[(size_t)3 * o + d];
        if (has_C) for (int d = 0; d < 9; ++d) A[(6 + d) * Np + i] += (T)sC[(size_t)9 * o + d];
        if (has_F) for (int d = 0; d < 9; ++d) A[(15 + d) * Np + i] += (T)sF[(size_t)9 * o + d];
    } else {
        for (int d = 0; d < 3; ++d) { sx[(size_t)3 * o + d] = (double)A[d * Np + i]; sv[(size_t)3 * o + d] = (double)A[(3 + d) * Np + i]; }
        for (int d = 0; d < 9; ++d) { sC[(size_t)9 * o + d] = (double)A[(6 + d) * Np + i]; sF[(size_t)9 * o + d] = (double)A[(15 + d) * Np + i]; }
    }
}
template <class T> __global__ void k_set_mats(Dev<T> D, const double* st, const int* perm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.Npad) return;
    int o = i < D.N ? perm[i] : perm[0];
    D.mu[i] = (T)st[o]; D.lam[i] = (T)st[(size_t)D.N + o]; D.ys[i] = (T)st[(size_t)2 * D.N + o];
}
__global__ void k_copy_frame(char* state, size_t frame_bytes, int src, int dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n16 = frame_bytes / 16;
    const uint4* s = reinterpret_cast<const uint4*>(state + (size_t)src * frame_bytes);
    uint4* d = reinterpret_cast<uint4*>(state + (size_t)dst * frame_bytes);
    if (i < n16) d[i] = s[i];
}

// global += local; local = 0   (pose adjoints after the cross-rank sum)
__global__ void k_merge_pose_adj(double* g, double* l, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { g[i] += l[i]; l[i] = 0.0; }
}

struct PrimChainArgs {
    int P;
    int tie_first;                   // plmpm_config.minmax_tie (adjoint routing of the clamps)
    int action_dim[kMaxPrim];
    int kin[kMaxPrim];
    double scale[kMaxPrim][PLMPM_MAX_ACTION_DIM];
    double lo[kMaxPrim][3], hi[kMaxPrim][3];
    double min_gap[kMaxPrim];
};
// primitive trajectories the serial kinematics kernels walk (all double, [(F+1)][P][.])
struct ChainBufs {
    double *ppos, *prot, *pgap, *pv, *pw, *pgv;
    double *ppos_a, *prot_a, *pgap_a, *pv_a, *pw_a, *pgv_a, *act_a;
};
struct ActionArg { double a[kMaxPrim * PLMPM_MAX_ACTION_DIM]; };

// set_action: action_buffer[step] = clipped action; v,w for the step's frames (primive_base.py:166-198)
__global__ void k_set_action(PrimChainArgs A, ActionArg act, int step, int nsub, double* actbuf, double* pv, double* pw, double* pgv) {
    const int p = blockIdx.x;            // one workgroup per primitive; its threads share the substeps
    double ab[PLMPM_MAX_ACTION_DIM];
    for (int k = 0; k < PLMPM_MAX_ACTION_DIM; ++k) ab[k] = act.a[p * PLMPM_MAX_ACTION_DIM + k];
    if (threadIdx.x == 0) {
        double* o = actbuf + ((size_t)step * A.P + p) * PLMPM_MAX_ACTION_DIM;
        for (int k = 0; k < PLMPM_MAX_ACTION_DIM; ++k) o[k] = ab[k];
    }
    if (A.action_dim[p] <= 0) return;
    for (int j = step * nsub + threadIdx.x; j < (step + 1) * nsub; j += blockDim.x) {
        double* v = pv + ((size_t)j * A.P + p) * 3;
        double* w = pw + ((size_t)j * A.P + p) * 3;
        for (int k = 0; k < 3; ++k) v[k] = ab[k] * A.scale[p][k] / nsub;
        if (A.action_dim[p] > 3) for (int k = 0; k < 3; ++k) w[k] = ab[k + 3] * A.scale[p][k + 3] / nsub;
        if (A.kin[p] == PLMPM_KIN_CHOPSTICKS) pgv[(size_t)j * A.P + p] = ab[6] * A.scale[p][6] / nsub;   // primitives.py:109
    }
}
// forward_kinematics over frames [first, first+n) (primive_base.py:117-121)
// The chain is serial in the frame index; the pose (and, in reverse, its adjoint) is carried in registers from one
// frame to the next -- going through memory instead costs a store -> load round trip per frame (~1.5 us each, 39
// frames per env step).  The per-frame inputs that do not depend on the chain (velocities; in reverse also the poses
// and the kernels' share of the adjoints) are first staged in LDS by the whole workgroup, in parallel: read one frame
// ahead from global memory they still cost one L2 round trip per frame (0.65 us forward, 2.8 us in reverse).
constexpr int kChainThreads = 64;
constexpr int kChainFwdWords = 7, kChainBwdWords = 23;          // doubles staged per (frame, primitive)
constexpr size_t kChainMaxLds = 64 * 1024;                        // longer chains read global memory one frame ahead
template <bool STAGED>
__global__ __launch_bounds__(kChainThreads) void k_fk_chain(PrimChainArgs A, int first, int n, ChainBufs B) {
    extern __shared__ double sm[];
    const int p = blockIdx.x;            // one workgroup per primitive: p is wave-uniform, A.*[p] are scalar loads
    if (STAGED) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const size_t a = (size_t)(first + i) * A.P + p;
            double* q = sm + (size_t)i * kChainFwdWords;
            for (int k = 0; k < 3; ++k) { q[k] = B.pv[a * 3 + k]; q[3 + k] = B.pw[a * 3 + k]; }
            q[6] = B.pgv[a];
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const size_t a0 = (size_t)first * A.P + p;
    double pos[3], rot[4], gap = B.pgap[a0];
    for (int k = 0; k < 3; ++k) pos[k] = B.ppos[a0 * 3 + k];
    for (int k = 0; k < 4; ++k) rot[k] = B.prot[a0 * 4 + k];
    double v[3], w[3], gv;
    auto inputs = [&](int s, double* V3, double* W3, double& GV) {
        if (STAGED) {
            const double* q = sm + (size_t)(s - first) * kChainFwdWords;
            for (int k = 0; k < 3; ++k) { V3[k] = q[k]; W3[k] = q[3 + k]; }
            GV = q[6];
        } else {
            const size_t a = (size_t)s * A.P + p;
            for (int k = 0; k < 3; ++k) { V3[k] = B.pv[a * 3 + k]; W3[k] = B.pw[a * 3 + k]; }
            GV = B.pgv[a];
        }
    };
    inputs(first, v, w, gv);
    for (int s = first; s < first + n; ++s) {
        const size_t b = (size_t)(s + 1) * A.P + p;
        double vn[3] = {0, 0, 0}, wn[3] = {0, 0, 0}, gvn = 0.0;        // inputs of the next frame, in flight during this one
        if (s + 1 < first + n) inputs(s + 1, vn, wn, gvn);
        double pos1[3], rot1[4], gap1 = gap;
        if (A.kin[p] == PLMPM_KIN_CHOPSTICKS)
            fk_chopsticks_fwd_d(pos, rot, v, w, gap, gv, A.min_gap[p], A.lo[p], A.hi[p], pos1, rot1, &gap1);
        else if (A.kin[p] == PLMPM_KIN_ROLLINGPIN)
            fk_rollingpin_fwd_d(pos, rot, v, A.lo[p], A.hi[p], pos1, rot1);
        else
            fk_fwd_d(pos, rot, v, w, A.lo[p], A.hi[p], pos1, rot1);
        for (int k = 0; k < 3; ++k) { B.ppos[b * 3 + k] = pos1[k]; pos[k] = pos1[k]; v[k] = vn[k]; w[k] = wn[k]; }
        for (int k = 0; k < 4; ++k) { B.prot[b * 4 + k] = rot1[k]; rot[k] = rot1[k]; }
        if (A.kin[p] == PLMPM_KIN_CHOPSTICKS) B.pgap[b] = gap1;
        gap = gap1; gv = gvn;
    }
}
// forward_kinematics.grad for frames first+n-1..first, then set_velocity.grad for env step `step`.
// On entry X_a[frame] holds what the contact / loss kernels accumulated; on exit the complete adjoint.
template <bool STAGED>
__global__ __launch_bounds__(kChainThreads) void k_fk_chain_grad(PrimChainArgs A, int first, int n, int step, ChainBufs B) {
    extern __shared__ double sm[];
    if (STAGED) {
        // frame s, primitive p: pos 0-2, v 3-5, w 6-8, own pos adjoint 9-11, rot 12-15, own rot adjoint 16-19, gap, gap_vel, own gap adjoint
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const size_t a = (size_t)(first + i) * A.P + blockIdx.x;
            double* q = sm + (size_t)i * kChainBwdWords;
            for (int k = 0; k < 3; ++k) { q[k] = B.ppos[a * 3 + k]; q[3 + k] = B.pv[a * 3 + k]; q[6 + k] = B.pw[a * 3 + k]; q[9 + k] = B.ppos_a[a * 3 + k]; }
            for (int k = 0; k < 4; ++k) { q[12 + k] = B.prot[a * 4 + k]; q[16 + k] = B.prot_a[a * 4 + k]; }
            q[20] = B.pgap[a]; q[21] = B.pgv[a]; q[22] = B.pgap_a[a];
        }
        __syncthreads();
    }
    const int p = blockIdx.x;            // one workgroup per primitive (see k_fk_chain)
    if (threadIdx.x != 0 || A.action_dim[p] <= 0) return;
    double va_sum[3] = {0, 0, 0}, wa_sum[3] = {0, 0, 0}, ga_sum = 0.0;
    const size_t bl = (size_t)(first + n) * A.P + p;
    double pos1_a[3], rot1_a[4], gap1_a = B.pgap_a[bl];        // complete adjoint of frame s+1, carried
    for (int k = 0; k < 3; ++k) pos1_a[k] = B.ppos_a[bl * 3 + k];
    for (int k = 0; k < 4; ++k) rot1_a[k] = B.prot_a[bl * 4 + k];
    // frame s: pose, velocities and the kernels' share of its adjoint, loaded one frame ahead
    double pos[3], rot[4], v[3], w[3], gap, gv, own_p[3], own_r[4], own_g;
    auto load = [&](size_t a, double* P3, double* R4, double* V3, double* W3, double& G, double& GV, double* OP, double* OR, double& OG) {
        if (STAGED) {
            const double* q = sm + (a / A.P - (size_t)first) * kChainBwdWords;
            for (int k = 0; k < 3; ++k) { P3[k] = q[k]; V3[k] = q[3 + k]; W3[k] = q[6 + k]; OP[k] = q[9 + k]; }
            for (int k = 0; k < 4; ++k) { R4[k] = q[12 + k]; OR[k] = q[16 + k]; }
            G = q[20]; GV = q[21]; OG = q[22];
            return;
        }
        for (int k = 0; k < 3; ++k) { P3[k] = B.ppos[a * 3 + k]; V3[k] = B.pv[a * 3 + k]; W3[k] = B.pw[a * 3 + k]; OP[k] = B.ppos_a[a * 3 + k]; }
        for (int k = 0; k < 4; ++k) { R4[k] = B.prot[a * 4 + k]; OR[k] = B.prot_a[a * 4 + k]; }
        G = B.pgap[a]; GV = B.pgv[a]; OG = B.pgap_a[a];
    };
    load((size_t)(first + n - 1) * A.P + p, pos, rot, v, w, gap, gv, own_p, own_r, own_g);
    for (int s = first + n - 1; s >= first; --s) {
        const size_t a = (size_t)s * A.P + p;
        double posn[3] = {0, 0, 0}, rotn[4] = {1, 0, 0, 0}, vn[3] = {0, 0, 0}, wn[3] = {0, 0, 0}, gapn = 0, gvn = 0, opn[3] = {0, 0, 0},
               orn[4] = {0, 0, 0, 0}, ogn = 0;
        if (s > first) load((size_t)(s - 1) * A.P + p, posn, rotn, vn, wn, gapn, gvn, opn, orn, ogn);
        double va[3], wa[3] = {0.0, 0.0, 0.0}, pa[3] = {own_p[0], own_p[1], own_p[2]}, ra[4] = {own_r[0], own_r[1], own_r[2], own_r[3]}, ga = own_g;
        if (A.kin[p] == PLMPM_KIN_CHOPSTICKS) {
            double gva = 0.0;
            fk_chopsticks_bwd_d(pos, rot, v, w, gap, gv, A.min_gap[p], A.lo[p], A.hi[p], pos1_a, rot1_a, gap1_a, pa, ra, &ga, va, wa, &gva, A.tie_first);
            B.pgv_a[a] = gva;
            B.pgap_a[a] = ga;
            ga_sum += gva;
        } else if (A.kin[p] == PLMPM_KIN_ROLLINGPIN)
            fk_rollingpin_bwd_d(pos, rot, v, A.lo[p], A.hi[p], pos1_a, rot1_a, pa, ra, va, A.tie_first);
        else
            fk_bwd_d(pos, rot, v, w, A.lo[p], A.hi[p], pos1_a, rot1_a, pa, ra, va, wa, A.tie_first);
        for (int k = 0; k < 3; ++k) {
            B.pv_a[a * 3 + k] = va[k]; B.pw_a[a * 3 + k] = wa[k]; va_sum[k] += va[k]; wa_sum[k] += wa[k];
            B.ppos_a[a * 3 + k] = pa[k]; pos1_a[k] = pa[k];
            pos[k] = posn[k]; v[k] = vn[k]; w[k] = wn[k]; own_p[k] = opn[k];
        }
        for (int k = 0; k < 4; ++k) { B.prot_a[a * 4 + k] = ra[k]; rot1_a[k] = ra[k]; rot[k] = rotn[k]; own_r[k] = orn[k]; }
        gap1_a = ga; gap = gapn; gv = gvn; own_g = ogn;
    }
    double* aa = B.act_a + ((size_t)step * A.P + p) * PLMPM_MAX_ACTION_DIM;
    for (int k = 0; k < 3; ++k) aa[k] += va_sum[k] * A.scale[p][k] / n;
    if (A.action_dim[p] > 3) for (int k = 0; k < 3; ++k) aa[k + 3] += wa_sum[k] * A.scale[p][k + 3] / n;
    if (A.kin[p] == PLMPM_KIN_CHOPSTICKS) aa[6] += ga_sum * A.scale[p][6] / n;
}

// ---- loss -----------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double r = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_max(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double r = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmax(r, sh[i]);
    __syncthreads();
    return r;
}
// density / sdf losses (loss.py:145-153) + IoU sums (loss.py:239-254)
// dl: deterministic mode only (else null) -- two integer limbs per loss scalar, see det_add (plmpm_kernels.h)
__device__ __forceinline__ void ls_add(double* ls, long long* dl, int slot, double v) {
    if (dl) det_add(dl + 2 * slot, dl + 2 * slot + 1, v);
    else atomicAdd(&ls[slot], v);
}
template <class T> __global__ void k_loss_reduce(size_t G, int nbxy, int gz, int z0, int z1, const T* gm, const T* td, const T* ts, double* ls, long long* dl) {
    __shared__ double sh[8];
    double dens = 0, sdf = 0, mx = 0, dot = 0, sum = 0;
    const unsigned nb2 = (unsigned)nbxy;                      // blocks per z-plane of the window (32-bit division)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += (size_t)gridDim.x * blockDim.x) {
        int z = gz + (int)((unsigned)(i >> 6) / nb2) * 4 + (int)((i & 63) >> 4);
        if (z < z0 || z >= z1) continue;                      // nodes owned by another rank
        double g = (double)gm[i], t = (double)td[i];
        dens += fabs(g - t); sdf += (double)ts[i] * g; mx = fmax(mx, g); dot += g * t; sum += g;
    }
    dens = block_sum(dens, sh); sdf = block_sum(sdf, sh); dot = block_sum(dot, sh); sum = block_sum(sum, sh);
    mx = block_max(mx, sh);
    if (threadIdx.x == 0) {
        ls_add(ls, dl, LS_DENSITY, dens); ls_add(ls, dl, LS_SDF, sdf); ls_add(ls, dl, LS_DOT, dot); ls_add(ls, dl, LS_SUMGM, sum);
        atomicMax(reinterpret_cast<unsigned long long*>(&ls[LS_MAXGM]), (unsigned long long)__double_as_longlong(mx));
    }
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
// contact distance passes (loss.py:116-135).  mode 0: hard min, 1: soft normaliser, 2: soft weighted sum.
// Grid-stride over the particles with a bounded number of workgroups, one result per workgroup and primitive: the
// partial results all land on the same word, and same-address atomics cost ~5 ns EACH on this chip (one per wave
// made this kernel 180 us at 500k particles).
__device__ __forceinline__ double block_min(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double r = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmin(r, sh[i]);
    __syncthreads();
    return r;
}
template <class T> __global__ void k_contact(Dev<T> D, int f, int mode, double* ls, long long* dl) {
    __shared__ double sh[8];
    const double* X = frame_x(D, f);
    for (int q = 0; q < D.nprim; ++q) {
        if (!D.prim[q].movable) continue;
        const PrimT<T> pr = prim_at(D, q, f);
        const double dn = mode == 2 ? ls[LS_DNORM + q] : 1.0;
        double acc = mode == 0 ? 1e30 : 0.0;
        for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < D.N; p += gridDim.x * blockDim.x) {
            const double x[3] = {X[p], X[D.Npad + p], X[2 * D.Npad + p]};
            const double d = fmax(prim_sdf(pr, x), 0.0);
            if (mode == 0) acc = fmin(acc, d);
            else {
                const double sw = 1.0 / (1.0 + d * d * 10000.0);
                acc += mode == 1 ? sw : d * sw / dn;
            }
        }
        if (mode == 0) {
            const double m = block_min(acc, sh);
            // non-negative doubles order like their bit patterns; skip the atomic when it cannot lower the minimum
            if (threadIdx.x == 0 && m < ls[LS_MIND + q])
                atomicMin(reinterpret_cast<unsigned long long*>(&ls[LS_MIND + q]), (unsigned long long)__double_as_longlong(m));
        } else {
            const double v = block_sum(acc, sh);
            if (threadIdx.x == 0) ls_add(ls, dl, (mode == 1 ? LS_DNORM : LS_MIND) + q, v);
        }
    }
}
// compute_loss_kernel_grad (loss.py:210-237) per particle: density + sdf through grid_m, contact through sdf.
template <class T>
__global__ void k_loss_grad(Dev<T> D, int f, int which, const T* gm, const T* td, const T* ts, const double* ls,
                            double w_sdf, double w_density, double w_contact, int soft, long long* dl, int argmin) {
    __shared__ double sacc[kMaxPrim * 8];
    __shared__ long long sdet[kMaxPrim * 8 * 2];          // deterministic mode: integer limbs instead of sacc
    if (threadIdx.x < kMaxPrim * 8) { sacc[threadIdx.x] = 0.0; sdet[2 * threadIdx.x] = 0; sdet[2 * threadIdx.x + 1] = 0; }
    __syncthreads();
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < D.N) {
        const double* X = frame_x(D, f);
        double x[3] = {X[p], X[D.Npad + p], X[2 * D.Npad + p]};
        int base[3];
        T fx[3], w[3][3], dw[3][3];
        stencil<T, double>(x, D.P.inv_dx, base, fx, w, dw);
        clamp_to_reach(D, base);
        double fxa[3] = {0, 0, 0};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                for (int l = 0; l < 3; ++l) {
                    int idx = node_index(D, base[0] + i, base[1] + j, base[2] + l);
                    double diff = (double)gm[idx] - (double)td[idx];
                    double sg = diff > 0 ? 1.0 : (diff < 0 ? -1.0 : 0.0);          // d|x|/dx with sgn(0) = 0
                    double ga = (w_density * sg + w_sdf * (double)ts[idx]) * (double)D.P.p_mass;
                    fxa[0] += ga * (double)(dw[i][0] * w[j][1] * w[l][2]);
                    fxa[1] += ga * (double)(w[i][0] * dw[j][1] * w[l][2]);
                    fxa[2] += ga * (double)(w[i][0] * w[j][1] * dw[l][2]);
                }
        double xa[3] = {fxa[0] * (double)D.P.inv_dx, fxa[1] * (double)D.P.inv_dx, fxa[2] * (double)D.P.inv_dx};
        for (int q = 0; q < D.nprim; ++q) {
            if (!D.prim[q].movable) continue;
            PrimT<T> pr = prim_at(D, q, f);
            double sd = prim_sdf(pr, x);
            if (!max_to_lhs(sd, 0.0, D.P.tie_first)) continue;          // max(sdf, 0): adjoint to sdf iff 0 < sdf
            double md = ls[LS_MIND + q];
            double coef;
            // hard contact, ti.atomic_min(min_dist, d) (loss.py:123-128): differentiated as an add by Taichi 0.7.x as far as
            // it is known (unverified) -- every particle gets min_dist's adjoint; plmpm_config.contact_min_adjoint
            // = 1 sends it to the particle(s) that attain the minimum instead (the mathematical derivative)
            if (!soft) { if (argmin && fmax(sd, 0.0) != md) continue; coef = w_contact * 2.0 * md; }
            else {
                double dn = ls[LS_DNORM + q];
                double den = 1.0 + sd * sd * 10000.0;
                double sw = 1.0 / den, dsw = -20000.0 * sd / (den * den);
                coef = w_contact * 2.0 * md * (sw + sd * dsw - md * dsw) / dn;
            }
            double pa[3] = {0, 0, 0}, ra[4] = {0, 0, 0, 0}, ga = 0.0;
            if (pr.shape == SHAPE_SPHERE) {                 // d sdf/dx = (x - c)/len ; d sdf/dc = -that
                double dvec[3] = {x[0] - pr.pos[0], x[1] - pr.pos[1], x[2] - pr.pos[2]};
                double L = len14(dvec[0], dvec[1], dvec[2]);
                for (int d = 0; d < 3; ++d) { double g = coef * dvec[d] / L; xa[d] += g; pa[d] = -g; }
            } else {                                        // sdf = sdf_local(inv_trans(x, pos, rot))
                double loc[3], iq[4], na0[3] = {0, 0, 0}, loca[3] = {0, 0, 0};
                inv_trans(x, pr.pos, pr.rot, loc, iq);
                shape_local_adj(pr.shape, pr.par, loc, coef, na0, loca, &ga);
                inv_trans_adj(x, pr.pos, pr.rot, iq, loca, pa, ra);
                for (int d = 0; d < 3; ++d) xa[d] -= pa[d];  // d/dx = -d/dpos
            }
            if (dl) {
                for (int d = 0; d < 3; ++d) det_add(&sdet[2 * (q * 8 + d)], &sdet[2 * (q * 8 + d) + 1], pa[d]);
                for (int d = 0; d < 4; ++d) det_add(&sdet[2 * (q * 8 + 3 + d)], &sdet[2 * (q * 8 + 3 + d) + 1], ra[d]);
                det_add(&sdet[2 * (q * 8 + 7)], &sdet[2 * (q * 8 + 7) + 1], ga);
            } else {
                for (int d = 0; d < 3; ++d) if (pa[d] != 0.0) atomicAdd(&sacc[q * 8 + d], pa[d]);
                for (int d = 0; d < 4; ++d) if (ra[d] != 0.0) atomicAdd(&sacc[q * 8 + 3 + d], ra[d]);
                if (ga != 0.0) atomicAdd(&sacc[q * 8 + 7], ga);
            }
        }
        T* A = D.adj[which];
        for (int d = 0; d < 3; ++d) A[d * D.Npad + p] += (T)xa[d];
    }
    __syncthreads();
    if (dl) {
        // the workgroup's integer sums go on into the global limbs (slot LS_COUNT + q * 8 + c); k_det_small_resolve adds
        // the totals into the pose adjoints
        if (threadIdx.x < D.nprim * 8 * 2 && sdet[threadIdx.x] != 0)
            atomicAdd(reinterpret_cast<unsigned long long*>(dl + 2 * LS_COUNT + threadIdx.x), (unsigned long long)sdet[threadIdx.x]);
    } else if (threadIdx.x < D.nprim * 8) {
        double v = sacc[threadIdx.x];
        int q = threadIdx.x / 8, c = threadIdx.x % 8;
        if (v != 0.0) {
            if (c < 3) atomicAdd(&D.ppos_a[((size_t)f * D.nprim + q) * 3 + c], v);
            else if (c < 7) atomicAdd(&D.prot_a[((size_t)f * D.nprim + q) * 4 + (c - 3)], v);
            else atomicAdd(&D.pgap_a[(size_t)f * D.nprim + q], v);
        }
    }
}
// deterministic mode: the integer limbs of the loss scalars and of k_loss_grad's pose adjoints -> their double targets
template <class T> __global__ void k_det_small_resolve(Dev<T> D, int f, long long* dl, double* ls) {
    const int t = threadIdx.x;
    if (t >= LS_COUNT + D.nprim * 8) return;
    const long long hi = dl[2 * t], lo = dl[2 * t + 1];
    if (!(hi | lo)) return;
    dl[2 * t] = 0; dl[2 * t + 1] = 0;
    const double v = det_value(hi, lo);
    if (t < LS_COUNT) { ls[t] += v; return; }
    const int q = (t - LS_COUNT) / 8, c = (t - LS_COUNT) % 8;
    if (c < 3) D.ppos_a[((size_t)f * D.nprim + q) * 3 + c] += v;
    else if (c < 7) D.prot_a[((size_t)f * D.nprim + q) * 4 + (c - 3)] += v;
    else D.pgap_a[(size_t)f * D.nprim + q] += v;
}
// target SDF sweep (loss.py:81-101), double, linear [i][j][k] layout
__global__ void k_sdf_sweep(int n, double dx, double inf, const double* dens, const double* sdf_c, const double* np_c,
                            double* sdf, double* npn, int* changed) {
    size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t G = (size_t)n * n * n;
    if (I >= G) return;
    int k = I % n, j = (I / n) % n, i = I / ((size_t)n * n);
    double gx = i * dx, gy = j * dx, gz = k * dx;
    double best = inf, bx = npn[3 * I], by = npn[3 * I + 1], bz = npn[3 * I + 2];
    if (dens[I] > 1e-4) { best = 0.0; bx = gx; by = gy; bz = gz; }
    else {
        for (int a = -3; a < 3; ++a)
            for (int b = -3; b < 3; ++b)
                for (int c = -3; c < 3; ++c) {
                    int vi = i + a, vj = j + b, vk = k + c;
                    if (vi < 0 || vj < 0 || vk < 0 || vi >= n || vj >= n || vk >= n) continue;
                    if (a == 0 && b == 0 && c == 0) continue;
                    size_t V = ((size_t)vi * n + vj) * n + vk;
                    if (sdf_c[V] < inf) {
                        double ex = gx - np_c[3 * V], ey = gy - np_c[3 * V + 1], ez = gz - np_c[3 * V + 2];
                        double dist = sqrt(ex * ex + ey * ey + ez * ez + 1e-8);
                        if (dist < best) { best = dist; bx = np_c[3 * V]; by = np_c[3 * V + 1]; bz = np_c[3 * V + 2]; }
                    }
                }
    }
    if (best != sdf_c[I] || bx != np_c[3 * I] || by != np_c[3 * I + 1] || bz != np_c[3 * I + 2]) *changed = 1;
    sdf[I] = best; npn[3 * I] = bx; npn[3 * I + 1] = by; npn[3 * I + 2] = bz;
}
// host grids are dense (n,n,n) [i][j][k]; the device holds the blocked window.  Upload: the window's part of the dense
// grid; download: the dense grid is zeroed first, the window's nodes written over it.
template <class T> __global__ void k_upload_grid(Dev<T> D, int n, const double* lin, T* blocked) {
    size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= (size_t)D.nbx * D.nby * D.nbz * 64) return;
    int nd[3];
    block_nodes(D, (int)(I >> 6), (int)(I & 63), nd);
    blocked[I] = (T)lin[((size_t)nd[0] * n + nd[1]) * n + nd[2]];
}
template <class T> __global__ void k_download_grid(Dev<T> D, int n, const T* blocked, double* lin) {
    size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= (size_t)D.nbx * D.nby * D.nbz * 64) return;
    int nd[3];
    block_nodes(D, (int)(I >> 6), (int)(I & 63), nd);
    lin[((size_t)nd[0] * n + nd[1]) * n + nd[2]] = (double)blocked[I];
}
// dst += src (the neighbour's copy of exchanged block planes, fields that no grid kernel adds on first touch)
template <class T> __global__ void k_add_region(T* dst, const T* src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
template <class T> __global__ void k_grid_stats(Dev<T> D, unsigned long long* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t G = (size_t)D.nbx * D.nby * D.nbz * 64;
    if (i < G && D.gin[0][i] > T(0)) atomicAdd(&out[0], 1ULL);
    if (i < G / 64 && D.flags[flag_slot(D, (int)i)]) atomicAdd(&out[1], 1ULL);
}

// ---------------------------------------------------------------------------------------------
static PrimChainArgs chain_args(const plmpm_sim* s) {
    PrimChainArgs A;
    memset(&A, 0, sizeof A);
    A.P = s->P;
    A.tie_first = s->cfg.minmax_tie != 0;
    for (int p = 0; p < s->P; ++p) {
        A.action_dim[p] = s->prims[p].action_dim;
        A.kin[p] = s->prims[p].kinematics;
        for (int k = 0; k < PLMPM_MAX_ACTION_DIM; ++k) A.scale[p][k] = s->prims[p].action_scale[k];
        for (int k = 0; k < 3; ++k) { A.lo[p][k] = s->prims[p].lower_bound[k]; A.hi[p][k] = s->prims[p].upper_bound[k]; }
        A.min_gap[p] = s->prims[p].params[2];           // Chopsticks: params = h, r, minimal_gap
    }
    return A;
}
static ChainBufs chain_bufs(const plmpm_sim* s) {
    ChainBufs B;
    B.ppos = s->ppos; B.prot = s->prot; B.pgap = s->pgap; B.pv = s->pv; B.pw = s->pw; B.pgv = s->pgv;
    B.ppos_a = s->ppos_a; B.prot_a = s->prot_a; B.pgap_a = s->pgap_a; B.pv_a = s->pv_a; B.pw_a = s->pw_a;
    B.pgv_a = s->pgv_a; B.act_a = s->act_a;
    return B;
}
static inline int nblocks_particles(const plmpm_sim* s, int frame) { return (s->epochN[s->frame_epoch[frame]] + kBlock - 1) / kBlock; }
static const HaloIn kNoHalo = {0, {0, 0}, {0, 0}, {nullptr, nullptr}, 0};
#ifndef PLB_POSE_WG
#define PLB_POSE_WG 16
#endif
// workgroups at the head of every k_p2g_grad launch that finish grid_op.grad's pose adjoints (64 waves: the blocks in
// contact with a manipulator number a few dozen)
constexpr int kPoseWG = PLB_POSE_WG;
constexpr int kClearWG = 64;
static inline int nblocks_grid(const plmpm_sim* s) { return (s->nblk + (kBlock / 64) - 1) / (kBlock / 64); }
// persistent grid kernels: a fixed number of workgroups, each striding over its share of the block flags
static inline int nwg_grid(const plmpm_sim* s) { return s->gwg; }

// clear arguments for the grids of frame `frame` (fused-grid engines)
#ifndef PLB_PK_DEFAULT
#define PLB_PK_DEFAULT 0
#endif
#ifndef PLB_FUSE_GRID_DEFAULT
#define PLB_FUSE_GRID_DEFAULT 0      // measured (round 3, profiles/r03_notes.md): not yet faster than the grid kernels at config 3
#endif
template <class T> static ClearArgs<T> clear_args(const plmpm_sim* s, int frame) {
    ClearArgs<T> A;
    memset(&A, 0, sizeof A);
    A.frame = frame;
    if (frame < 0) return A;
    const Dev<T> Df = make_dev<T>(s, frame, true);
    A.nwg = nblocks_particles(s, frame);
    A.nwg_clear = 0;
    for (int c = 0; c < 4; ++c) A.gin[c] = Df.gin[c];
    for (int c = 0; c < 3; ++c) A.goa[c] = Df.goa[c];
    A.flags = Df.flags;
    return A;
}
// the frame a fused-grid reverse substep left behind, when no g2p.grad of the frame before it follows
template <class T> static int fg_flush_t(plmpm_sim* s) {
    if (s->fg_pending < 0) return 0;
    const int f = s->fg_pending;
    Dev<T> D = make_dev<T>(s, f, true);
    hipLaunchKernelGGL((k_clear_boxes<T>), dim3(nblocks_particles(s, f)), dim3(kBlock), 0, s->stream, D, clear_args<T>(s, f));
    s->dirty[f] = 0;
    s->fg_pending = -1;
    return 0;
}
#define FG_FLUSH(s) do { if ((s)->fg_pending >= 0) fg_flush_t<T>(s); } while (0)

template <class T> static int substep_fwd(plmpm_sim* s, int f) {
    FG_FLUSH(s);
    Dev<T> D = make_dev<T>(s, f);
    if (s->fg) {                 // p2g | g2p with grid_op in its tile fill
        if (s->dirty[f]) LAUNCH(s, K_CLEAR, (k_clear_active<T>), dim3(nblocks_grid(s)), D);
        LAUNCH_P2G(s, K_P2G, true, D, f);
        s->dirty[f] = 1;
        LAUNCH(s, K_FG_G2P, (k_g2p<T, true>), dim3(nblocks_particles(s, f)), D, f);
        s->vnear[f] = 1;
        return 0;
    }
    s->vnear[f] = 0;
    if (s->store) {
        if (s->dirty[f]) LAUNCH(s, K_CLEAR, (k_clear_active<T>), dim3(nblocks_grid(s)), D);   // frame reused without a backward pass
        LAUNCH_P2G(s, K_P2G, true, D, f);
        LAUNCH(s, K_GRID_OP, (k_grid_op<T, false>), dim3(nwg_grid(s)), D, f, s->halo_in[PLMPM_HALO_GRID_IN]);                    // keep grid_in for substep_grad
        s->dirty[f] = 1;
    } else {
        LAUNCH_P2G(s, K_P2G, true, D, f);
        LAUNCH(s, K_GRID_OP, (k_grid_op<T, true>), dim3(nwg_grid(s)), D, f, kNoHalo);
    }
    LAUNCH(s, K_G2P, (k_g2p<T>), dim3(nblocks_particles(s, f)), D, f);
    return 0;
}
template <class T> static int substep_bwd(plmpm_sim* s, int f) {
    Dev<T> D = make_dev<T>(s, f);
    const int src = (f + 1) & 1, dst = f & 1;
    // frame f+1 re-sorted by the env step that starts there: its v in THIS frame's order was kept aside
    const T* vnext = s->frame_epoch[f + 1] != s->frame_epoch[f] ? (const T*)(s->vend + (size_t)s->frame_epoch[f + 1] * 3 * s->Npad * s->tsz) : nullptr;
    if (s->fg && s->dirty[f]) {
        // g2p.grad (+ grid_op in its tile fill, + the clear of frame f+1's grids) | p2g.grad (+ grid_op.grad in its tile fill)
        const bool chained = s->fg_pending == f + 1;
        if (!chained) {
            FG_FLUSH(s);
            hipMemsetAsync(s->contact, 0, 4, s->stream);                          // both contact counters: no g2p.grad reset them
            hipMemsetAsync(s->contact + s->nblk + 1, 0, 4, s->stream);
        }
        ++s->contact_stamp;                  // (the marks start at 0 and the first stamp is 1)
        Dev<T> Dg = make_dev<T>(s, f, true);
        ClearArgs<T> ca = clear_args<T>(s, chained ? f + 1 : -1);
        ca.nwg_clear = chained ? kClearWG : 0;                            // workgroups at the head of the launch do the clear
        const int nwg = nblocks_particles(s, f) + ca.nwg_clear;
        if (s->vnear[f]) LAUNCH(s, K_FG_G2P_GRAD, (k_g2p_grad<T, false, 1 + NEAR_LOAD>), dim3(nwg), Dg, f, src, dst, vnext, ca);
        else LAUNCH(s, K_FG_G2P_GRAD, (k_g2p_grad<T, false, 1 + NEAR_EVAL>), dim3(nwg), Dg, f, src, dst, vnext, ca);
        if (chained) s->dirty[f + 1] = 0;
        LAUNCH(s, K_FG_P2G_GRAD, (k_p2g_grad<T, true>), dim3(nblocks_particles(s, f) + kPoseWG), Dg, f, src, dst, kPoseWG);
        s->fg_pending = f;                   // dirty[f] stays set until the frame's grids are cleared
        s->adj_frame[dst] = f;
        return 0;
    }
    FG_FLUSH(s);
    if (!(s->store && s->dirty[f])) {        // this frame's grid is not resident: recompute it (mpm_simulator.py:265-268)
        LAUNCH_P2G(s, K_P2G_RE, false, D, f);
        LAUNCH(s, K_GRID_OP_RE, (k_grid_op<T, false>), dim3(nwg_grid(s)), D, f, kNoHalo);
    }
    LAUNCH_G2P_GRAD(s, D, f, src, dst, vnext);
    LAUNCH(s, K_GRID_OP_GRAD, (k_grid_op_grad<T>), dim3(nwg_grid(s)), D, f, s->halo_in[PLMPM_HALO_GRID_OUT_ADJ]);
    LAUNCH_P2G_GRAD(s, D, f, src, dst);
    if (s->store) s->dirty[f] = 0;           // k_grid_op_grad left grid_in / flags of this frame clean
    s->adj_frame[dst] = f;
    return 0;
}

// Whole env step forward in store mode: p2g(f0) | grid_op(f0) | [g2p(f-1)+p2g(f) fused | grid_op(f)] ... | g2p(last)
template <class T> static int step_fwd_fused(plmpm_sim* s, int first, int n) {
    FG_FLUSH(s);
    if (s->fg) {                 // p2g(f0) | [g2p(f-1) + p2g(f) with grid_op(f-1) in the tile fill] ... | g2p(last) with grid_op(last)
        for (int f = first; f < first + n; ++f) {
            Dev<T> D = make_dev<T>(s, f);
            if (s->dirty[f]) LAUNCHG_CLEAR(s, D);
            if (f == first) LAUNCH_P2G(s, K_P2G, true, D, f);
            else {
                PrevGrid<T> pg;
                memset(&pg, 0, sizeof pg);
                for (int c = 0; c < 4; ++c) pg.gin[c] = (const T*)(s->gstore + (size_t)(f - 1) * s->gstride) + (size_t)c * s->G;
                pg.vout = (const Vec4<T>*)(s->vstore + (size_t)(f - 1) * s->gstride);      // nodes near a primitive; all nodes of boxes that exceed the LDS tile
                s->vnear[f - 1] = 1;
                if constexpr (sizeof(T) == 4) {
                    if (s->pk) { LAUNCHB(s, K_FG_G2P_P2G, (k_g2p_p2g_pk<true>), dim3(nblocks_particles(s, f)), kBlockPk, D, f, pg); }
                    else LAUNCH(s, K_FG_G2P_P2G, (k_g2p_p2g<T, false, true>), dim3(nblocks_particles(s, f)), D, f, pg);
                } else LAUNCH(s, K_FG_G2P_P2G, (k_g2p_p2g<T, false, true>), dim3(nblocks_particles(s, f)), D, f, pg);
            }
            s->dirty[f] = 1;
        }
        Dev<T> D = make_dev<T>(s, first + n - 1);
        LAUNCH(s, K_FG_G2P, (k_g2p<T, true>), dim3(nblocks_particles(s, first + n - 1)), D, first + n - 1);
        s->vnear[first + n - 1] = 1;
        return 0;
    }
    for (int f = first; f < first + n; ++f) {
        Dev<T> D = make_dev<T>(s, f);
        s->vnear[f] = 0;
        if (s->dirty[f]) LAUNCHG_CLEAR(s, D);
        if (f == first) {
            LAUNCH_P2G(s, K_P2G, true, D, f);
        } else {
            const Vec4<T>* vprev = (const Vec4<T>*)(s->vstore + (size_t)(f - 1) * s->gstride);
            bool done = false;
            if constexpr (sizeof(T) == 4) {
                if (s->pk) {
                    PrevGrid<T> pg;
                    memset(&pg, 0, sizeof pg);
                    pg.vout = vprev;
                    LAUNCHB(s, K_G2P_P2G, (k_g2p_p2g_pk<false>), dim3(nblocks_particles(s, f)), kBlockPk, D, f, pg);
                    done = true;
                }
            }
            if (!done) LAUNCH_G2P_P2G(s, D, f, vprev);
        }
        LAUNCH(s, K_GRID_OP, (k_grid_op<T, false>), dim3(nwg_grid(s)), D, f, s->halo_in[PLMPM_HALO_GRID_IN]);
        s->dirty[f] = 1;
    }
    Dev<T> D = make_dev<T>(s, first + n - 1);
    LAUNCH(s, K_G2P, (k_g2p<T>), dim3(nblocks_particles(s, first + n - 1)), D, first + n - 1);
    return 0;
}

// phase-split variants used by the multi-GPU driver (store_grid mode only)
template <class T> static int phase_p2g(plmpm_sim* s, int f) {
    FG_FLUSH(s);
    Dev<T> D = make_dev<T>(s, f);
    if (s->dirty[f]) LAUNCH(s, K_CLEAR, (k_clear_active<T>), dim3(nblocks_grid(s)), D);
    LAUNCH_P2G(s, K_P2G, true, D, f);
    s->dirty[f] = 1;
    return 0;
}
// part: 0 every active block | 1 only the blocks outside the exchanged planes (nothing else: the halos may still be in
// flight) | 2 the blocks of the exchanged planes, then g2p
template <class T> static int phase_grid_g2p(plmpm_sim* s, int f, bool defer_g2p = false, int part = 0) {
    Dev<T> D = make_dev<T>(s, f);
    s->frame_epoch[f + 1] = s->frame_epoch[f];                 // g2p (now or fused into the next p2g) writes frame f + 1 in this order
    HaloIn H = s->halo_in[PLMPM_HALO_GRID_IN];
    H.part = part;
    s->vnear[f] = 0;
    LAUNCH(s, K_GRID_OP, (k_grid_op<T, false>), dim3(nwg_grid(s)), D, f, H);
    if (part == 1) return 0;
    if (!defer_g2p) LAUNCH(s, K_G2P, (k_g2p<T>), dim3(nblocks_particles(s, f)), D, f);
    return 0;
}
// g2p(f-1), deferred by the previous phase_grid_g2p, fused with p2g(f) exactly as in step_fwd_fused
template <class T> static int phase_g2p_p2g(plmpm_sim* s, int f) {
    FG_FLUSH(s);
    Dev<T> D = make_dev<T>(s, f);
    if (s->dirty[f]) LAUNCHG_CLEAR(s, D);
    const Vec4<T>* vprev = (const Vec4<T>*)(s->vstore + (size_t)(f - 1) * s->gstride);
    LAUNCH_G2P_P2G(s, D, f, vprev);
    s->dirty[f] = 1;
    return 0;
}
template <class T> static int phase_grad_scatter(plmpm_sim* s, int f) {
    FG_FLUSH(s);
    Dev<T> D = make_dev<T>(s, f);
    const T* vnext = s->frame_epoch[f + 1] != s->frame_epoch[f] ? (const T*)(s->vend + (size_t)s->frame_epoch[f + 1] * 3 * s->Npad * s->tsz) : nullptr;
    LAUNCH_G2P_GRAD(s, D, f, (f + 1) & 1, f & 1, vnext);
    return 0;
}
template <class T> static int phase_grad_gather(plmpm_sim* s, int f, int part = 0) {
    Dev<T> D = make_dev<T>(s, f);
    HaloIn H = s->halo_in[PLMPM_HALO_GRID_OUT_ADJ];
    H.part = part;
    LAUNCH(s, K_GRID_OP_GRAD, (k_grid_op_grad<T>), dim3(nwg_grid(s)), D, f, H);
    if (part == 1) return 0;
    LAUNCH_P2G_GRAD(s, D, f, (f + 1) & 1, f & 1);
    s->dirty[f] = 0;
    s->adj_frame[f & 1] = f;
    s->adj_epoch[f & 1] = s->frame_epoch[f];
    return 0;
}

template <class T> static int build_prims_t(plmpm_sim* s, int first, int n) {
    Dev<T> D = make_dev<T>(s);
    hipLaunchKernelGGL((k_build_prims<T>), dim3((n * s->P + 63) / 64), dim3(64), 0, s->stream, D, first, n, (PrimT<T>*)s->ptab);
    return 0;
}
#define DISPATCH(s, fn, ...) ((s)->cfg.dtype == PLMPM_F64 ? fn<double>(__VA_ARGS__) : fn<float>(__VA_ARGS__))

// ---------------------------------------------------------------------------------------------
// storage slot -> host row of state / gradient I/O.  Single GPU: the caller's particle index in every epoch.  Slab
// engines: caller order only in epoch 0; once particles have migrated the rows of a frame are its storage order
// (plmpm_get_ids names them)
static int* perm_of(const plmpm_sim* s, int epoch) {
    if (epoch <= 0) return s->perm_d;
    return s->dist ? s->iota : s->perm_store + (size_t)(epoch - 1) * s->Npad;
}
// material arrays in the storage order of `epoch`, from the caller-order master copy
template <class T> static int set_materials_t(plmpm_sim* s, int epoch) {
    const bool filled = s->mats_filled;
    s->mats_epoch = epoch;
    if (s->dist && epoch != 0) return 0;          // slab engines: epochs > 0 got their materials with the migrating rows
    if (s->mats_uniform && filled && !s->dist) return 0;      // every particle the same: a permutation changes nothing
    s->mats_filled = true;
    Dev<T> D = make_dev<T>(s);
    hipLaunchKernelGGL((k_set_mats<T>), dim3(s->Npad / 256), dim3(256), 0, s->stream, D, s->mats_master, perm_of(s, epoch));
    return 0;
}

template <class T> static int unpack_t(plmpm_sim* s, int f, int hx, int hv, int hF, int hC) {
    Dev<T> D = make_dev<T>(s, f);
    hipLaunchKernelGGL((k_unpack_frame<T>), dim3(s->Npad / 256), dim3(256), 0, s->stream, D, f, s->staging, perm_of(s, s->frame_epoch[f]), hx, hv, hF, hC);
    return 0;
}

template <class T> static int pack_t(plmpm_sim* s, int f) {
    Dev<T> D = make_dev<T>(s, f);
    hipLaunchKernelGGL((k_pack_frame<T>), dim3(s->Npad / 256), dim3(256), 0, s->stream, D, f, s->staging, perm_of(s, s->frame_epoch[f]));
    return 0;
}

template <class T> static int adj_io_t(plmpm_sim* s, int which, int add, int hx, int hv, int hF, int hC, int epoch = -1) {
    Dev<T> D = make_dev<T>(s);
    if (epoch < 0) epoch = s->adj_epoch[which];
    D.N = s->epochN[epoch];
    hipLaunchKernelGGL((k_adj_io<T>), dim3(s->Npad / 256), dim3(256), 0, s->stream, D, which, s->staging, perm_of(s, epoch), add, hx, hv, hF, hC);
    return 0;
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
template <class T> __global__ void k_hilbert_keys(Dev<T> D, int f, int bits, unsigned* keys, int* idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.Npad) return;
    idx[i] = i;
    if (i >= D.N) { keys[i] = 1u << (3 * bits); return; }          // one past the largest cell key: padding sorts last
    const double* X = frame_x(D, f);
    int b[3];
    for (int d = 0; d < 3; ++d) {
        b[d] = (int)(X[d * D.Npad + i] * (double)D.P.inv_dx - 0.5);
        b[d] = min(max(b[d], 0), D.P.n - 1);
    }
    keys[i] = hilbert_key_dev((unsigned)b[0], (unsigned)b[1], (unsigned)b[2], bits);
}
// frame f gathered through `order` (new slot i <- old slot order[i]) into `out`; v of the old frame kept in `vend`
template <class T> __global__ void k_permute_frame(Dev<T> D, int f, const int* order, char* out, T* vend, const int* perm_old, int* perm_new) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.Npad) return;
    const int Np = D.Npad, j = order[i];
    const double* X = frame_x(D, f);
    const T* R = frame_r(D, f);
    double* Xo = reinterpret_cast<double*>(out);
    T* Ro = reinterpret_cast<T*>(out + (size_t)3 * 8 * Np);
    for (int d = 0; d < 3; ++d) Xo[d * Np + i] = X[d * Np + j];
    for (int d = 0; d < 21; ++d) Ro[d * Np + i] = R[d * Np + j];
    for (int d = 0; d < 3; ++d) vend[d * Np + i] = R[d * Np + i];          // old order, same slot
    if (i < D.N) perm_new[i] = perm_old[j];
}
// Counting sort by cell along the curve: a particle's key IS its bin, so the order is histogram -> exclusive scan ->
// scatter through a per-bin cursor -- 4 launches instead of the ~20 of the library radix sort (0.13 ms for 500k pairs,
// launch-bound).  The order inside a cell is whatever the cursor hands out (the in-wave sort does not care); the
// deterministic engine and grids with more than 2^25 cells take the radix sort.
template <class T> __global__ void k_cell_hist(Dev<T> D, int f, int bits, unsigned* keys, unsigned* hist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.N) return;
    const double* X = frame_x(D, f);
    int b[3];
    for (int d = 0; d < 3; ++d) b[d] = min(max((int)(X[d * D.Npad + i] * (double)D.P.inv_dx - 0.5), 0), D.P.n - 1);
    const unsigned k = hilbert_key_dev((unsigned)b[0], (unsigned)b[1], (unsigned)b[2], bits);
    keys[i] = k;
    atomicAdd(&hist[k], 1u);
}
__global__ void k_cell_scatter(int n, int npad, const unsigned* keys, unsigned* cursor, int* order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    if (i >= n) { order[i] = i; return; }                   // padding slots stay where they are
    order[atomicAdd(&cursor[keys[i]], 1u)] = i;
}
template <class T> static int resort_frame_t(plmpm_sim* s, int f, int epoch) {
    Dev<T> D = make_dev<T>(s, f);
    s->epochN[epoch] = D.N;
    const int nb = s->Npad / 256;
    int bits = 1;
    while ((1 << bits) < s->n) ++bits;
    if (s->cell_bins && !s->det) {
        HIPCHK(hipMemsetAsync(s->cell_hist, 0, s->cell_bins * 4, s->stream));
        hipLaunchKernelGGL((k_cell_hist<T>), dim3(nb), dim3(256), 0, s->stream, D, f, bits, s->skey[0], s->cell_hist);
        if (plmpm_exclusive_scan(s->sort_tmp, s->sort_tmp_bytes, s->cell_hist, s->cell_hist, s->cell_bins, s->stream) != 0) return fail("resort: device scan failed");
        hipLaunchKernelGGL(k_cell_scatter, dim3(nb), dim3(256), 0, s->stream, D.N, s->Npad, s->skey[0], s->cell_hist, s->sidx[1]);
        T* vend = (T*)(s->vend + (size_t)epoch * 3 * s->Npad * s->tsz);
        hipLaunchKernelGGL((k_permute_frame<T>), dim3(nb), dim3(256), 0, s->stream, D, f, s->sidx[1], s->frame_tmp, vend,
                           perm_of(s, s->frame_epoch[f]), perm_of(s, epoch));
        HIPCHK(hipMemcpyAsync(s->state + (size_t)f * s->frame_bytes, s->frame_tmp, s->frame_bytes, hipMemcpyDeviceToDevice, s->stream));
        s->frame_epoch[f] = epoch;
        return 0;
    }
    // (keys without the lowest 1 / 2 curve levels -- fewer radix passes -- were measured: the kernels lose more to the
    // coarser order than the sort saves, profiles/r02_notes.md)
    hipLaunchKernelGGL((k_hilbert_keys<T>), dim3(nb), dim3(256), 0, s->stream, D, f, bits, s->skey[0], s->sidx[0]);
    if (plmpm_sort_pairs(s->sort_tmp, s->sort_tmp_bytes, s->skey[0], s->skey[1], s->sidx[0], s->sidx[1], s->Npad, 3 * bits + 1, s->stream) != 0)
        return fail("resort: device sort failed");
    T* vend = (T*)(s->vend + (size_t)epoch * 3 * s->Npad * s->tsz);
    hipLaunchKernelGGL((k_permute_frame<T>), dim3(nb), dim3(256), 0, s->stream, D, f, s->sidx[1], s->frame_tmp, vend,
                       perm_of(s, s->frame_epoch[f]), perm_of(s, epoch));
    HIPCHK(hipMemcpyAsync(s->state + (size_t)f * s->frame_bytes, s->frame_tmp, s->frame_bytes, hipMemcpyDeviceToDevice, s->stream));
    s->frame_epoch[f] = epoch;
    return 0;
}
// adjoint frame `which` from the storage order of epoch `from` to that of epoch `to`: slot j of `to` holds the particle
// (caller index perm_to[j]) that sat in slot inv_from[perm_to[j]] of `from` -- one gather pass over the 24 rows.  (Through
// the float64 caller-order staging buffer, as state I/O goes, this was two passes and 0.18 ms per re-sort boundary.)
__global__ void k_inv_perm(const int* perm, int n, int* inv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[perm[i]] = i;
}
template <class T> __global__ void k_adj_regather(const T* in, T* out, const int* inv_from, const int* perm_to, int n, int Np) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Np) return;
    if (j >= n) { for (int d = 0; d < 24; ++d) out[(size_t)d * Np + j] = T(0); return; }
    const int i = inv_from[perm_to[j]];
    for (int d = 0; d < 24; ++d) out[(size_t)d * Np + j] = in[(size_t)d * Np + i];
}
template <class T> static int convert_adjoint_t(plmpm_sim* s, int which, int from, int to) {
    if (from == to) return 0;
    if (s->dist) return fail("slab engine: the adjoint frame is in storage epoch %d but epoch %d is needed -- particles migrated in between; "
                             "run plmpm_migrate_adjoint_begin / _finish on the boundary frame first", from, to);
    const int nb = s->Npad / 256;
    int* inv = s->sidx[0];                               // sort scratch: idle during the reverse sweep
    hipLaunchKernelGGL(k_inv_perm, dim3(nb), dim3(256), 0, s->stream, perm_of(s, from), s->N, inv);
    hipLaunchKernelGGL((k_adj_regather<T>), dim3(nb), dim3(256), 0, s->stream, (const T*)s->adj[which], (T*)s->frame_tmp, inv, perm_of(s, to), s->N, s->Npad);
    HIPCHK(hipMemcpyAsync(s->adj[which], s->frame_tmp, (size_t)24 * s->Npad * s->tsz, hipMemcpyDeviceToDevice, s->stream));
    s->adj_epoch[which] = to;
    return 0;
}

template <class T> static int upload_grid_t(plmpm_sim* s, const double* lin_d, char* dst) {
    hipLaunchKernelGGL((k_upload_grid<T>), dim3((unsigned)((s->G + 255) / 256)), dim3(256), 0, s->stream, make_dev<T>(s), s->n, lin_d, (T*)dst);
    return 0;
}

template <class T> static int download_grid_t(plmpm_sim* s, const char* src, double* lin_d) {
    (void)hipMemsetAsync(lin_d, 0, s->Gfull * 8, s->stream);
    hipLaunchKernelGGL((k_download_grid<T>), dim3((unsigned)((s->G + 255) / 256)), dim3(256), 0, s->stream, make_dev<T>(s), s->n, (const T*)src, lin_d);
    return 0;
}

template <class T> static int loss_scatter_t(plmpm_sim* s, int f) {
    Dev<T> D = make_dev<T>(s, f);
    hipMemsetAsync(s->loss_gm, 0, s->G * s->tsz, s->stream);
    if (nblocks_particles(s, f) > 0) {
        if (s->det) {
            hipLaunchKernelGGL((k_grid_mass<T, true>), dim3(nblocks_particles(s, f)), dim3(kBlock), 0, s->stream, D, f, (T*)s->loss_gm);
            DET_RESOLVE(s, (T*)s->loss_gm, (T*)nullptr, (T*)nullptr, (T*)nullptr);
        } else
            hipLaunchKernelGGL((k_grid_mass<T>), dim3(nblocks_particles(s, f)), dim3(kBlock), 0, s->stream, D, f, (T*)s->loss_gm);
    }
    return 0;
}

// mode 0: hard min, 1: soft normaliser, 2: soft weighted sum (needs the global normaliser in lscal)
template <class T> static int loss_contact_pass_t(plmpm_sim* s, int f, int mode) {
    Dev<T> D = make_dev<T>(s, f);
    bool any = false;
    for (int p = 0; p < s->P; ++p) any |= s->prims[p].action_dim > 0;
    if (any) {
        hipLaunchKernelGGL((k_contact<T>), dim3(std::min(s->Npad / 256, 512)), dim3(256), 0, s->stream, D, f, mode, s->lscal, s->det_small);
        if (s->det) hipLaunchKernelGGL((k_det_small_resolve<T>), dim3(1), dim3(128), 0, s->stream, D, f, s->det_small, s->lscal);
    }
    return 0;
}
// loss scalars to their start values, on the device (no host buffer to keep alive, no synchronisation)
__global__ void k_ls_init(double* ls, int soft) {
    const int i = threadIdx.x;
    if (i < LS_COUNT) ls[i] = (!soft && i >= LS_MIND && i < LS_MIND + kMaxPrim) ? 100000.0 : 0.0;      // loss.py:189-191
}
static int loss_reset_scalars(plmpm_sim* s) {
    hipLaunchKernelGGL(k_ls_init, dim3(1), dim3(64), 0, s->stream, s->lscal, s->soft_contact ? 1 : 0);
    return 0;
}

template <class T> static int loss_reduce_t(plmpm_sim* s) {
    hipLaunchKernelGGL((k_loss_reduce<T>), dim3(256), dim3(256), 0, s->stream, s->G, s->nbw[0] * s->nbw[1], s->go[2], s->cfg.slab_z0, s->cfg.slab_z1,
                       (const T*)s->loss_gm, (const T*)s->loss_td, (const T*)s->loss_ts, s->lscal, s->det_small);
    if (s->det) hipLaunchKernelGGL((k_det_small_resolve<T>), dim3(1), dim3(128), 0, s->stream, make_dev<T>(s), 0, s->det_small, s->lscal);
    return 0;
}

template <class T> static int loss_grad_t(plmpm_sim* s, int f) {
    Dev<T> D = make_dev<T>(s, f);
    hipLaunchKernelGGL((k_loss_grad<T>), dim3(s->Npad / 256), dim3(256), 0, s->stream, D, f, f & 1, (const T*)s->loss_gm,
                       (const T*)s->loss_td, (const T*)s->loss_ts, s->lscal, s->w_sdf, s->w_density, s->w_contact, s->soft_contact, s->det_small,
                       s->cfg.contact_min_adjoint);
    if (s->det) hipLaunchKernelGGL((k_det_small_resolve<T>), dim3(1), dim3(128), 0, s->stream, D, f, s->det_small, s->lscal);
    return 0;
}

template <class T> static int grid_stats_t(plmpm_sim* s, int f, unsigned long long* d_out) {
    Dev<T> D = make_dev<T>(s);
    D.N = s->epochN[s->frame_epoch[f]];
    // recompute the scatter of frame f without consuming it, count, then clear
    LAUNCH_P2G(s, K_P2G_RE, false, D, f);
    hipLaunchKernelGGL((k_grid_stats<T>), dim3((unsigned)((s->G + 255) / 256)), dim3(256), 0, s->stream, D, d_out);
    hipLaunchKernelGGL((k_clear_active<T>), dim3(nblocks_grid(s)), dim3(kBlock), 0, s->stream, D);
    return 0;
}

extern "C" {

const char* plmpm_last_error(void) { return g_err.c_str(); }
int plmpm_version(void) { return 1; }

int plmpm_create(const plmpm_config* cfg, const plmpm_primitive* prims, plmpm_handle* out) {
    REQUIRE(cfg && out, "null argument");
    REQUIRE(cfg->dtype == PLMPM_F32 || cfg->dtype == PLMPM_F64, "dtype must be PLMPM_F32 or PLMPM_F64");
    REQUIRE(cfg->n_grid >= 8 && cfg->n_grid % 4 == 0, "n_grid must be a multiple of 4 (got %d)", cfg->n_grid);
    REQUIRE(cfg->n_particles > 0, "n_particles must be positive");
    REQUIRE(cfg->n_primitives >= 0 && cfg->n_primitives <= PLMPM_MAX_PRIMITIVES, "at most %d primitives", PLMPM_MAX_PRIMITIVES);
    REQUIRE(cfg->max_frames >= 1, "max_frames must be >= 1");
    REQUIRE(cfg->n_primitives == 0 || prims, "prims is null");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device visible: this engine has no CPU path");
    plmpm_sim* s = new plmpm_sim();
    s->cfg = *cfg;
    if (s->cfg.slab_z1 <= s->cfg.slab_z0) { s->cfg.slab_z0 = 0; s->cfg.slab_z1 = cfg->n_grid; }
    s->P = cfg->n_primitives;
    s->act_ofs[0] = 0;
    for (int p = 0; p < s->P; ++p) {
        s->prims[p] = prims[p];
        if (prims[p].action_dim < 0 || prims[p].action_dim > PLMPM_MAX_ACTION_DIM) { delete s; return fail("bad action_dim"); }
        if (prims[p].kinematics < PLMPM_KIN_DEFAULT || prims[p].kinematics > PLMPM_KIN_CHOPSTICKS) { delete s; return fail("unknown kinematics %d", prims[p].kinematics); }
        if ((prims[p].shape == PLMPM_CHOPSTICKS) != (prims[p].kinematics == PLMPM_KIN_CHOPSTICKS)) {
            delete s;
            return fail("primitive %d: the Chopsticks shape and PLMPM_KIN_CHOPSTICKS go together", p);
        }
        if (prims[p].shape == PLMPM_CHOPSTICKS && prims[p].action_dim != 7) {
            delete s;
            return fail("primitive %d: Chopsticks take a 7-dim action (3 linear, 3 angular, 1 grasp; primitives.py:92)", p);
        }
        if (prims[p].shape < PLMPM_SPHERE || prims[p].shape > PLMPM_CHOPSTICKS) { delete s; return fail("primitive %d: unknown shape %d", p, prims[p].shape); }
        s->act_ofs[p + 1] = s->act_ofs[p] + prims[p].action_dim;
    }
    s->act_total = s->act_ofs[s->P];
    s->N = cfg->n_particles;
    const int cap = std::max(cfg->particle_capacity, cfg->n_particles);
    s->Npad = (int)align_up(cap, kBlock);
    s->n = cfg->n_grid; s->Gfull = (size_t)s->n * s->n * s->n;
    // grid window: the box of 4^3 blocks that is allocated and swept (all-zero grid_lo / grid_hi = the whole grid)
    for (int d = 0; d < 3; ++d) {
        int lo = cfg->grid_lo[d], hi = cfg->grid_hi[d];
        if (hi <= lo) { lo = 0; hi = s->n; }
        lo = std::max(0, lo) / 4 * 4;
        hi = std::min(s->n, (hi + 3) / 4 * 4);
        if (hi - lo < 4) { delete s; return fail("grid window axis %d is empty: [%d, %d)", d, cfg->grid_lo[d], cfg->grid_hi[d]); }
        s->go[d] = lo; s->nbw[d] = (hi - lo) / 4;
    }
    s->nblk = s->nbw[0] * s->nbw[1] * s->nbw[2]; s->G = (size_t)s->nblk * 64;
    // persistent grid kernels: a power-of-two number of workgroups (<= kGridWG, about one wave per 1-4 blocks), each
    // with its blocks' flags side by side
    s->gwg = 1; s->gwg_log2 = 0;
    while (s->gwg * 2 <= kGridWG && s->gwg * 2 * (kBlock / 64) <= s->nblk) { s->gwg *= 2; ++s->gwg_log2; }
    s->fs = (s->nblk + s->gwg - 1) / s->gwg;
    s->nflag = s->gwg * s->fs;
    s->F = cfg->max_frames;
    s->tsz = cfg->dtype == PLMPM_F64 ? 8 : 4;
    s->frame_bytes = (size_t)s->Npad * (24 + 21 * s->tsz);
    size_t P1 = std::max(s->P, 1);
    s->ws.state_bytes = (size_t)(s->F + 1) * s->frame_bytes;
    s->ws.adjoint_bytes = align_up(2 * 24 * s->Npad * s->tsz, 256) + 3 * align_up(s->Npad * s->tsz, 256) + align_up((size_t)s->Npad * 4, 256);
    s->ws.grid_bytes = 4 * align_up(s->G * 4 * s->tsz, 256) + align_up((size_t)s->nflag * 4, 256) + 3 * align_up(s->G * s->tsz, 256);
    s->dist = s->cfg.slab_z0 > 0 || s->cfg.slab_z1 < cfg->n_grid || cfg->slab_halo > 0;
    s->resort = cfg->resort_steps > 0 && !s->dist && cfg->substeps > 0;
    // storage epochs: single GPU one per re-sort (+2: the alternating pair of copy-mode episodes); slab engines one per
    // migration, at most one per env step
    s->n_epochs = s->resort ? s->F / (cfg->substeps * cfg->resort_steps) + 3 : 1;
    if (s->dist && cfg->substeps > 0) s->n_epochs = s->F / cfg->substeps + 4;
    s->epochN.assign(s->n_epochs + 1, s->N);
    s->mig.assign(s->n_epochs + 1, plmpm_sim::MigInfo());
    s->frame_epoch.assign(s->F + 2, 0);
    const bool sorts = s->resort || s->dist;
    s->sort_cap = s->dist ? s->Npad + s->Npad / 2 : s->Npad;          // slab engines sort stayers + arrivals
    s->mig_max_rows = s->Npad / 4;
    s->sort_tmp_bytes = sorts ? plmpm_sort_temp_bytes(s->sort_cap) : 0;
    s->cell_bins = 0;
    if (s->resort) {                                       // counting-sort re-sort: one bin per cell of the 2^bits cube
        int cb = 1;
        while ((1 << cb) < s->n) ++cb;
        if (3 * cb <= 25) {
            s->cell_bins = (size_t)1 << (3 * cb);
            s->sort_tmp_bytes = std::max(s->sort_tmp_bytes, plmpm_scan_temp_bytes(s->cell_bins));
            s->ws.adjoint_bytes += align_up(s->cell_bins * 4, 256);
        }
    }
    s->ws.adjoint_bytes += align_up((size_t)3 * s->Npad * 8, 256);                   // material master copy
    if (sorts)
        s->ws.adjoint_bytes += align_up((size_t)(s->n_epochs - 1) * s->Npad * 4, 256) + align_up((size_t)s->n_epochs * 3 * s->Npad * s->tsz, 256)
                               + 4 * align_up((size_t)s->sort_cap * 4, 256) + align_up(s->sort_tmp_bytes, 256) + align_up(s->frame_bytes, 256);
    if (s->dist)
        s->ws.adjoint_bytes += 3 * align_up((size_t)s->n_epochs * s->Npad * 4, 256)            // gid_store, mig_src, mig_leave
                               + align_up((size_t)s->n_epochs * 3 * s->Npad * s->tsz, 256)    // mats_store
                               + 2 * align_up((size_t)s->Npad * 4, 256) + 256                  // mig_dest, iota, mig_cnt
                               + 2 * align_up((size_t)s->mig_max_rows * 28 * 8, 256);          // packed rows of the leavers, per direction
    s->store = cfg->store_grid != 0;
    s->gstride = align_up(s->G * 4 * s->tsz, 256);
    if (s->store) s->ws.grid_bytes += 2 * (size_t)s->F * s->gstride + align_up((size_t)s->F * s->nflag * 4, 256);
    s->ws.grid_bytes += align_up((size_t)(s->F + 1) * (s->Npad / kBlock) * 8 * 4, 256) + align_up((size_t)2 * (s->nblk + 1) * 4, 256);
    // fused-grid path (grid_op inside the particle kernels' tile fills): one GPU, grid store, floating-point atomics.
    // PLMPM_FUSE_GRID=0 keeps the grid kernels (A/B measurements, and the reference for the parity test of the fused path)
    {
        const char* e = getenv("PLMPM_FUSE_GRID");
        const bool want = e ? e[0] != '0' : (PLB_FUSE_GRID_DEFAULT != 0);
        s->fg = s->store && !s->dist && cfg->deterministic == 0 && want;
    }
    if (s->fg) s->ws.grid_bytes += align_up(s->G * 4 * s->tsz, 256) + align_up((size_t)s->nblk * 4, 256);
    s->ws.grid_bytes += align_up((size_t)(s->F + 1) * kMaxPrim * sizeof(PrimT<double>), 256);
    {
        const char* e = getenv("PLMPM_PK");
        s->pk = cfg->dtype == PLMPM_F32 && cfg->deterministic == 0 && (e ? e[0] != '0' : (PLB_PK_DEFAULT != 0));
    }
    s->dirty.assign(s->F + 1, 0);
    s->vnear.assign(s->F + 1, 0);
    s->ws.misc_bytes = 2 * align_up((size_t)(s->F + 1) * P1 * 7 * 8, 256) + 2 * align_up((size_t)(s->F + 1) * P1 * 8 * 8, 256)  // poses(+adj), padded
                       + 4 * align_up((size_t)(s->F + 1) * P1 * 3 * 8, 256)                       // v,w (+adj)
                       + 4 * align_up((size_t)(s->F + 1) * P1 * 8, 256)                           // gap, gap_vel (+adj)
                       + 2 * align_up((size_t)(s->F + 1) * P1 * PLMPM_MAX_ACTION_DIM * 8, 256)    // action buffers (+adj)
                       + align_up(LS_COUNT * 8, 256) + align_up((size_t)s->Npad * 24 * 8, 256) + 512;
    if (s->dist) s->ws.misc_bytes += 3 * align_up((size_t)(s->F + 1) * P1 * 4 * 8, 256);
    s->det = cfg->deterministic != 0;
    if (s->det) {
        s->ws.grid_bytes += align_up(s->G * 8 * 8, 256);                                  // [8][G] integer limbs
        s->ws.misc_bytes += align_up((size_t)(LS_COUNT + kMaxPrim * 8) * 2 * 8, 256);
    }
    memset(s->halo_in, 0, sizeof s->halo_in);
    s->perm.resize(s->N);
    for (int i = 0; i < s->N; ++i) s->perm[i] = i;
    *out = s;
    return 0;
}

int plmpm_destroy(plmpm_handle s) {
    if (s) for (auto e : s->ev_pool) (void)hipEventDestroy(e);
    delete s;
    return 0;
}

int plmpm_workspace_bytes(plmpm_handle s, plmpm_workspace* out) {
    REQUIRE(s && out, "null argument");
    *out = s->ws;
    return 0;
}

int plmpm_bind_workspace(plmpm_handle s, void* state, void* adjoint, void* grid, void* misc) {
    REQUIRE(s && state && adjoint && grid && misc, "null argument");
    REQUIRE(((uintptr_t)state | (uintptr_t)adjoint | (uintptr_t)grid | (uintptr_t)misc) % 256 == 0, "workspaces must be 256-byte aligned");
    s->state = (char*)state; s->adjw = (char*)adjoint; s->gridw = (char*)grid; s->miscw = (char*)misc;
    char* p = s->adjw;
    auto take = [&](size_t bytes) { char* r = p; p += align_up(bytes, 256); return r; };
    s->adj[0] = take((size_t)24 * s->Npad * s->tsz);
    s->adj[1] = s->adj[0] + (size_t)24 * s->Npad * s->tsz;
    p = s->adjw + align_up(2 * 24 * s->Npad * s->tsz, 256);
    s->mu = take(s->Npad * s->tsz); s->lam = take(s->Npad * s->tsz); s->ys = take(s->Npad * s->tsz);
    s->perm_d = (int*)take((size_t)s->Npad * 4);
    s->mats_master = (double*)take((size_t)3 * s->Npad * 8);
    if (s->resort || s->dist) {
        s->perm_store = (int*)take((size_t)(s->n_epochs - 1) * s->Npad * 4);
        s->vend = take((size_t)s->n_epochs * 3 * s->Npad * s->tsz);
        for (int i = 0; i < 2; ++i) { s->skey[i] = (unsigned*)take((size_t)s->sort_cap * 4); s->sidx[i] = (int*)take((size_t)s->sort_cap * 4); }
        s->sort_tmp = take(s->sort_tmp_bytes);
        if (s->cell_bins) s->cell_hist = (unsigned*)take(s->cell_bins * 4);
        s->frame_tmp = take(s->frame_bytes);
    }
    if (s->dist) {
        s->gid_store = (int*)take((size_t)s->n_epochs * s->Npad * 4);
        s->mig_src = (int*)take((size_t)s->n_epochs * s->Npad * 4);
        s->mig_leave = (int*)take((size_t)s->n_epochs * s->Npad * 4);
        s->mats_store = take((size_t)s->n_epochs * 3 * s->Npad * s->tsz);
        s->mig_dest = (int*)take((size_t)s->Npad * 4);
        s->iota = (int*)take((size_t)s->Npad * 4);
        s->mig_cnt = (int*)take(256);
        for (int i = 0; i < 2; ++i) s->mig_send[i] = (double*)take((size_t)s->mig_max_rows * 28 * 8);
    }
    REQUIRE((size_t)(p - s->adjw) <= s->ws.adjoint_bytes, "internal: adjoint workspace overflow");
    p = s->gridw;
    s->grid_in = take(s->G * 4 * s->tsz); s->grid_out = take(s->G * 4 * s->tsz);
    s->grid_out_adj = take(s->G * 4 * s->tsz); s->grid_in_adj = take(s->G * 4 * s->tsz);
    s->flags = (int*)take((size_t)s->nflag * 4);
    s->loss_gm = take(s->G * s->tsz); s->loss_td = take(s->G * s->tsz); s->loss_ts = take(s->G * s->tsz);
    if (s->store) {
        s->gstore = take((size_t)s->F * s->gstride);
        s->vstore = take((size_t)s->F * s->gstride);
        s->fstore = (int*)take((size_t)s->F * s->nflag * 4);
    }
    s->tiles = (int*)take((size_t)(s->F + 1) * (s->Npad / kBlock) * 8 * 4);
    s->contact = (int*)take((size_t)2 * (s->nblk + 1) * 4);
    if (s->fg) {
        s->grid_out_adj2 = take(s->G * 4 * s->tsz);
        s->contact_mark = (int*)take((size_t)s->nblk * 4);
    }
    s->ptab = take((size_t)(s->F + 1) * kMaxPrim * sizeof(PrimT<double>));
    s->det_grid = s->det ? (long long*)take(s->G * 8 * 8) : nullptr;
    REQUIRE((size_t)(p - s->gridw) <= s->ws.grid_bytes, "internal: grid workspace overflow");
    p = s->miscw;
    size_t P1 = std::max(s->P, 1), F1 = s->F + 1;
    s->ppos = (double*)take(F1 * P1 * 3 * 8); s->prot = (double*)take(F1 * P1 * 4 * 8);
    s->ppos_a = (double*)take(F1 * P1 * 3 * 8); s->prot_a = (double*)take(F1 * P1 * 4 * 8);
    s->pv = (double*)take(F1 * P1 * 3 * 8); s->pw = (double*)take(F1 * P1 * 3 * 8);
    s->pv_a = (double*)take(F1 * P1 * 3 * 8); s->pw_a = (double*)take(F1 * P1 * 3 * 8);
    s->pgap = (double*)take(F1 * P1 * 8); s->pgap_a = (double*)take(F1 * P1 * 8);
    s->pgv = (double*)take(F1 * P1 * 8); s->pgv_a = (double*)take(F1 * P1 * 8);
    s->act = (double*)take(F1 * P1 * PLMPM_MAX_ACTION_DIM * 8); s->act_a = (double*)take(F1 * P1 * PLMPM_MAX_ACTION_DIM * 8);
    s->lscal = (double*)take(LS_COUNT * 8);
    s->staging = (double*)take((size_t)s->Npad * 24 * 8);
    s->err_d = (int*)take(512);
    if (s->dist) { s->ppos_l = (double*)take(F1 * P1 * 3 * 8); s->prot_l = (double*)take(F1 * P1 * 4 * 8); s->pgap_l = (double*)take(F1 * P1 * 8); }
    s->det_small = s->det ? (long long*)take((size_t)(LS_COUNT + kMaxPrim * 8) * 2 * 8) : nullptr;
    REQUIRE((size_t)(p - s->miscw) <= s->ws.misc_bytes, "internal: misc workspace overflow");
    // initial contents: zero grids / adjoints / primitive buffers, identity order
    HIPCHK(hipMemsetAsync(s->state, 0, s->ws.state_bytes, s->stream));       // frames are first touched here, not inside a caller's timed region
    HIPCHK(hipMemsetAsync(s->adjw, 0, s->ws.adjoint_bytes, s->stream));
    HIPCHK(hipMemsetAsync(s->gridw, 0, s->ws.grid_bytes, s->stream));
    HIPCHK(hipMemsetAsync(s->miscw, 0, s->ws.misc_bytes, s->stream));
    HIPCHK(hipMemcpyAsync(s->perm_d, s->perm.data(), (size_t)s->N * 4, hipMemcpyHostToDevice, s->stream));
    if (s->dist) {
        std::vector<int> io(s->Npad);
        for (int i = 0; i < s->Npad; ++i) io[i] = i;
        HIPCHK(hipMemcpyAsync(s->iota, io.data(), (size_t)s->Npad * 4, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->gid_store, io.data(), (size_t)s->N * 4, hipMemcpyHostToDevice, s->stream));      // default ids: the row index
        HIPCHK(hipStreamSynchronize(s->stream));          // `io` goes out of scope
        s->ids0.assign(io.begin(), io.begin() + s->N);
    }
    // unit quaternions everywhere so an unset primitive frame is still a valid pose
    std::vector<double> rot(F1 * P1 * 4, 0.0);
    for (size_t i = 0; i < F1 * P1; ++i) rot[4 * i] = 1.0;
    HIPCHK(hipMemcpyAsync(s->prot, rot.data(), rot.size() * 8, hipMemcpyHostToDevice, s->stream));
    if (s->resort || s->dist) {
        // first use of the library sort loads its code object (~20 ms): pay that here, not in the first re-sorted step
        if (plmpm_sort_pairs(s->sort_tmp, s->sort_tmp_bytes, s->skey[0], s->skey[1], s->sidx[0], s->sidx[1], s->Npad, 8, s->stream) != 0)
            return fail("device sort unavailable");
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    s->bound = true;
    return 0;
}

int plmpm_set_stream(plmpm_handle s, void* hip_stream) {
    REQUIRE(s, "null handle");
    s->stream = (hipStream_t)hip_stream;
    return 0;
}

#define NEED_BOUND(s)                                                                                             \
    do {                                                                                                          \
        REQUIRE((s) && (s)->bound, "workspace not bound");                                                        \
        REQUIRE((s)->g2p_deferred < 0, "frame %d's g2p is deferred: call plmpm_p2g(frame + 1, chain = 1) next", (s)->g2p_deferred); \
    } while (0)
#define NEED_FRAME(s, f) REQUIRE((f) >= 0 && (f) <= (s)->F, "frame %d out of range [0,%d]", (f), (s)->F)

int plmpm_set_materials(plmpm_handle s, const double* mu, const double* lam, const double* ys) {
    NEED_BOUND(s);
    REQUIRE(mu && lam && ys, "null argument");
    size_t nb = (size_t)s->N * 8;
    // the common case -- one material for the whole body -- needs no re-gather when the storage order changes
    bool uni = true;
    for (int i = 1; i < s->N && uni; ++i) uni = mu[i] == mu[0] && lam[i] == lam[0] && ys[i] == ys[0];
    s->mats_uniform = uni;
    s->mats_filled = false;
    HIPCHK(hipMemcpyAsync(s->mats_master, mu, nb, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->mats_master + s->N, lam, nb, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->mats_master + 2 * (size_t)s->N, ys, nb, hipMemcpyHostToDevice, s->stream));
    s->have_mats = true;
    DISPATCH(s, set_materials_t, s, s->mats_epoch);
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}


// Index of cell (b0,b1,b2) along the 3-D Hilbert curve over a 2^bits cube (Skilling, "Programming the Hilbert
// curve", AIP Conf. Proc. 707, 2004).  Consecutive indices are face-adjacent cells, so any run of the sorted
// particle list -- a wavefront, a workgroup -- covers a compact box of cells.
static uint64_t hilbert_index(int b0, int b1, int b2, int bits) {
    uint32_t X[3] = {(uint32_t)b0, (uint32_t)b1, (uint32_t)b2};
    const uint32_t M = 1u << (bits - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    for (int i = 1; i < 3; ++i) X[i] ^= X[i - 1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
    for (int i = 0; i < 3; ++i) X[i] ^= t;
    uint64_t h = 0;
    for (int bit = bits - 1; bit >= 0; --bit)
        for (int i = 0; i < 3; ++i) h = (h << 1) | ((X[i] >> bit) & 1u);
    return h;
}

// Storage order chosen at reset: particles sorted along the Hilbert curve of their stencil-base cell (ties keep
// caller order).  With a Morton order ~4 % of the 256-particle workgroups straddle a long jump of the curve and
// their stencil box overflows the LDS tile; along the Hilbert curve the mean box is 230 nodes instead of 410 and
// 0.3 % overflow (config-3 cloud).
static void compute_order(plmpm_sim* s, const double* x) {
    const int n = s->n;
    int bits = 1;
    while ((1 << bits) < n) ++bits;
    std::vector<std::pair<uint64_t, int32_t>> key(s->N);
    for (int i = 0; i < s->N; ++i) {
        int b[3];
        for (int d = 0; d < 3; ++d) {
            b[d] = (int)(x[(size_t)3 * i + d] * n - 0.5);
            b[d] = std::min(std::max(b[d], 0), n - 1);
        }
        key[i] = {hilbert_index(b[0], b[1], b[2], bits), (int32_t)i};
    }
    std::sort(key.begin(), key.end());
    for (int i = 0; i < s->N; ++i) s->perm[i] = key[i].second;
}

int plmpm_set_frame(plmpm_handle s, int frame, const double* x, const double* v, const double* F, const double* C, int resort) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    if (resort) {
        REQUIRE(x && v && F && C, "resort needs the full state (all of x, v, F, C)");
        compute_order(s, x);
        HIPCHK(hipMemcpyAsync(s->perm_d, s->perm.data(), (size_t)s->N * 4, hipMemcpyHostToDevice, s->stream));
        std::fill(s->frame_epoch.begin(), s->frame_epoch.end(), 0);          // a new episode: every frame in the reset order
        s->epochN[0] = s->N;
        s->next_epoch = 1;
        s->steps_since_sort = 0;
        s->mats_epoch = 0;                       // whatever epoch the last rollout ended in: the materials that follow are epoch 0's
        if (s->dist) {       // global ids in the reset order
            std::vector<int32_t> g(s->N);
            for (int i = 0; i < s->N; ++i) g[i] = s->ids0[s->perm[i]];
            HIPCHK(hipMemcpyAsync(s->gid_store, g.data(), (size_t)s->N * 4, hipMemcpyHostToDevice, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
        }
        if (s->have_mats) DISPATCH(s, set_materials_t, s, 0);
    }
    const size_t N = s->epochN[s->frame_epoch[frame]];
    if (x) HIPCHK(hipMemcpyAsync(s->staging, x, N * 3 * 8, hipMemcpyHostToDevice, s->stream));
    if (v) HIPCHK(hipMemcpyAsync(s->staging + 3 * N, v, N * 3 * 8, hipMemcpyHostToDevice, s->stream));
    if (F) HIPCHK(hipMemcpyAsync(s->staging + 6 * N, F, N * 9 * 8, hipMemcpyHostToDevice, s->stream));
    if (C) HIPCHK(hipMemcpyAsync(s->staging + 15 * N, C, N * 9 * 8, hipMemcpyHostToDevice, s->stream));
    DISPATCH(s, unpack_t, s, frame, x != nullptr, v != nullptr, F != nullptr, C != nullptr);
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int plmpm_get_frame(plmpm_handle s, int frame, double* x, double* v, double* F, double* C) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    const size_t N = s->epochN[s->frame_epoch[frame]];
    DISPATCH(s, pack_t, s, frame);
    if (x) HIPCHK(hipMemcpyAsync(x, s->staging, N * 3 * 8, hipMemcpyDeviceToHost, s->stream));
    if (v) HIPCHK(hipMemcpyAsync(v, s->staging + 3 * N, N * 3 * 8, hipMemcpyDeviceToHost, s->stream));
    if (F) HIPCHK(hipMemcpyAsync(F, s->staging + 6 * N, N * 9 * 8, hipMemcpyDeviceToHost, s->stream));
    if (C) HIPCHK(hipMemcpyAsync(C, s->staging + 15 * N, N * 9 * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int plmpm_copy_frame(plmpm_handle s, int source, int target) {
    NEED_BOUND(s);
    NEED_FRAME(s, source);
    NEED_FRAME(s, target);
    size_t n16 = s->frame_bytes / 16;
    hipLaunchKernelGGL(k_copy_frame, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s->stream, s->state, s->frame_bytes, source, target);
    s->frame_epoch[target] = s->frame_epoch[source];
    if (s->P > 0) {
        HIPCHK(hipMemcpyAsync(s->ppos + (size_t)target * s->P * 3, s->ppos + (size_t)source * s->P * 3, (size_t)s->P * 3 * 8, hipMemcpyDeviceToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->prot + (size_t)target * s->P * 4, s->prot + (size_t)source * s->P * 4, (size_t)s->P * 4 * 8, hipMemcpyDeviceToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->pgap + (size_t)target * s->P, s->pgap + (size_t)source * s->P, (size_t)s->P * 8, hipMemcpyDeviceToDevice, s->stream));
    }
    return 0;
}

int plmpm_set_primitive_state(plmpm_handle s, int prim, int frame, const double* st) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(prim >= 0 && prim < s->P && st, "bad primitive index");
    HIPCHK(hipMemcpyAsync(s->ppos + ((size_t)frame * s->P + prim) * 3, st, 3 * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->prot + ((size_t)frame * s->P + prim) * 4, st + 3, 4 * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->pgap + (size_t)frame * s->P + prim, st + 7, 8, hipMemcpyHostToDevice, s->stream));
    {   // the per-substep primitive records that hold this pose (substeps frame-1 and frame)
        const int a = std::max(frame - 1, 0), b = std::min(frame, s->F - 1);
        if (b >= a) { if (s->cfg.dtype == PLMPM_F64) build_prims_t<double>(s, a, b - a + 1); else build_prims_t<float>(s, a, b - a + 1); }
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}
int plmpm_get_primitive_state(plmpm_handle s, int prim, int frame, double* st) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(prim >= 0 && prim < s->P && st, "bad primitive index");
    HIPCHK(hipMemcpyAsync(st, s->ppos + ((size_t)frame * s->P + prim) * 3, 3 * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(st + 3, s->prot + ((size_t)frame * s->P + prim) * 4, 4 * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(st + 7, s->pgap + (size_t)frame * s->P + prim, 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}
int plmpm_get_primitive_grad(plmpm_handle s, int prim, int frame, double* g) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(prim >= 0 && prim < s->P && g, "bad primitive index");
    HIPCHK(hipMemcpyAsync(g, s->ppos_a + ((size_t)frame * s->P + prim) * 3, 3 * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(g + 3, s->prot_a + ((size_t)frame * s->P + prim) * 4, 4 * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(g + 7, s->pgap_a + (size_t)frame * s->P + prim, 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}
__global__ void k_add_doubles(double* dst, double a0, double a1, double a2, double a3, int n) {
    const double a[4] = {a0, a1, a2, a3};
    if ((int)threadIdx.x < n) dst[threadIdx.x] += a[threadIdx.x];
}
int plmpm_add_primitive_grad(plmpm_handle s, int prim, int frame, const double* g) {
    NEED_BOUND(s);
    REQUIRE(s->adj_frame[0] >= 0 || s->adj_frame[1] >= 0, "add_primitive_grad: no reverse sweep has begun (plmpm_grad_begin clears the pose adjoints: call it first)");
    NEED_FRAME(s, frame);
    REQUIRE(prim >= 0 && prim < s->P && g, "bad primitive index");
    const size_t a = (size_t)frame * s->P + prim;
    hipLaunchKernelGGL(k_add_doubles, dim3(1), dim3(4), 0, s->stream, s->ppos_a + a * 3, g[0], g[1], g[2], 0.0, 3);
    hipLaunchKernelGGL(k_add_doubles, dim3(1), dim3(4), 0, s->stream, s->prot_a + a * 4, g[3], g[4], g[5], g[6], 4);
    hipLaunchKernelGGL(k_add_doubles, dim3(1), dim3(4), 0, s->stream, s->pgap_a + a, g[7], 0.0, 0.0, 0.0, 1);
    HIPCHK(hipGetLastError());
    return 0;
}
}  // extern "C"
// Primitive.sdf (primive_base.py:57-60; a ti.func in the reference): signed distance of n points to primitive `prim`
// at its pose of `frame`, evaluated by the same device function the collide / loss kernels use
template <class T> __global__ void k_prim_sdf(Dev<T> D, int q, int f, const double* pts, int n, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PrimT<T> pr = prim_at(D, q, f);
    const double x[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
    out[i] = prim_sdf(pr, x);
}
extern "C" {
int plmpm_primitive_sdf(plmpm_handle s, int prim, int frame, const double* points, int n, double* out) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(prim >= 0 && prim < s->P && points && out && n >= 0, "primitive_sdf: bad arguments");
    if (n == 0) return 0;
    double* d_in;
    HIPCHK(hipMalloc(&d_in, (size_t)n * 4 * 8));
    double* d_out = d_in + (size_t)n * 3;
    HIPCHK(hipMemcpyAsync(d_in, points, (size_t)n * 3 * 8, hipMemcpyHostToDevice, s->stream));
    if (s->cfg.dtype == PLMPM_F64) hipLaunchKernelGGL((k_prim_sdf<double>), dim3((n + 255) / 256), dim3(256), 0, s->stream, make_dev<double>(s), prim, frame, d_in, n, d_out);
    else hipLaunchKernelGGL((k_prim_sdf<float>), dim3((n + 255) / 256), dim3(256), 0, s->stream, make_dev<float>(s), prim, frame, d_in, n, d_out);
    HIPCHK(hipMemcpyAsync(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    (void)hipFree(d_in);
    return 0;
}
// Loss.min_dist / dist_norm of the movable primitives after the last loss evaluation (loss.py:116-135)
int plmpm_loss_contact_scalars(plmpm_handle s, double* min_dist, double* dist_norm) {
    NEED_BOUND(s);
    double ls[LS_COUNT];
    HIPCHK(hipMemcpyAsync(ls, s->lscal, sizeof ls, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (int q = 0; q < s->P; ++q) {
        if (min_dist) min_dist[q] = ls[LS_MIND + q];
        if (dist_norm) dist_norm[q] = ls[LS_DNORM + q];
    }
    return 0;
}
// measured HBM roof of this device: float4 copy of `bytes` (read + write), best of `reps`; GB/s of bytes moved
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_read16(const uint4* __restrict__ in, unsigned* __restrict__ out, size_t n) {
    uint4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = in[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x9e3779b9u) out[0] = a.x;
}
int plmpm_measure_hbm(void* src, void* dst, size_t bytes, int reps, void* hip_stream, double* copy_gbs, double* read_gbs) {
    REQUIRE(src && dst && bytes >= (1u << 20) && reps > 0 && copy_gbs && read_gbs, "measure_hbm: bad arguments");
    hipStream_t st = (hipStream_t)hip_stream;
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    const size_t n = bytes / 16;
    double best[2] = {0, 0};
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < reps + 1; ++r) {
            HIPCHK(hipEventRecord(a, st));
            if (k == 0) hipLaunchKernelGGL(k_copy16, dim3(16384), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, n);
            else hipLaunchKernelGGL(k_read16, dim3(2048), dim3(256), 0, st, (const uint4*)src, (unsigned*)dst, n);
            HIPCHK(hipEventRecord(b, st));
            HIPCHK(hipEventSynchronize(b));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, a, b));
            if (r > 0) best[k] = std::max(best[k], (k == 0 ? 2.0 : 1.0) * (double)bytes / (ms * 1e6));      // first run warms up
        }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *copy_gbs = best[0]; *read_gbs = best[1];
    return 0;
}
int plmpm_set_softness(plmpm_handle s, double softness) {
    REQUIRE(s, "null handle");
    s->softness = softness;
    return 0;
}

int plmpm_set_action(plmpm_handle s, int step, int n_substeps, const double* action) {
    NEED_BOUND(s);
    REQUIRE(n_substeps > 0 && step >= 0 && (step + 1) * n_substeps <= s->F, "set_action: frames [%d,%d) exceed max_frames %d",
            step * n_substeps, (step + 1) * n_substeps, s->F);
    if (s->P == 0) return 0;
    REQUIRE(action || s->act_total == 0, "null action");
    ActionArg a;
    memset(&a, 0, sizeof a);
    for (int p = 0; p < s->P; ++p)
        for (int k = 0; k < s->prims[p].action_dim; ++k) {
            double v = action[s->act_ofs[p] + k];
            a.a[p * PLMPM_MAX_ACTION_DIM + k] = std::min(1.0, std::max(-1.0, v));      // primitives.py:290
        }
    hipLaunchKernelGGL(k_set_action, dim3(s->P), dim3(kChainThreads), 0, s->stream, chain_args(s), a, step, n_substeps, s->act, s->pv, s->pw, s->pgv);
    return 0;
}

// Primitive.set_velocity (primive_base.py:184-192): v, w of the step's frames from action_buffer[step] as stored
__global__ void k_set_velocity(PrimChainArgs A, int prim, int step, int nsub, const double* actbuf, double* pv, double* pw, double* pgv) {
    const int p = prim;
    const double* ab = actbuf + ((size_t)step * A.P + p) * PLMPM_MAX_ACTION_DIM;
    if (A.action_dim[p] <= 0) return;
    for (int j = step * nsub + threadIdx.x; j < (step + 1) * nsub; j += blockDim.x) {
        double* v = pv + ((size_t)j * A.P + p) * 3;
        double* w = pw + ((size_t)j * A.P + p) * 3;
        for (int k = 0; k < 3; ++k) v[k] = ab[k] * A.scale[p][k] / nsub;
        if (A.action_dim[p] > 3) for (int k = 0; k < 3; ++k) w[k] = ab[k + 3] * A.scale[p][k + 3] / nsub;
        if (A.kin[p] == PLMPM_KIN_CHOPSTICKS) pgv[(size_t)j * A.P + p] = ab[6] * A.scale[p][6] / nsub;
    }
}
int plmpm_set_velocity(plmpm_handle s, int prim, int step, int n_substeps) {
    NEED_BOUND(s);
    REQUIRE(prim >= 0 && prim < s->P, "set_velocity: bad primitive index");
    REQUIRE(n_substeps > 0 && step >= 0 && (step + 1) * n_substeps <= s->F, "set_velocity: frames exceed max_frames");
    hipLaunchKernelGGL(k_set_velocity, dim3(1), dim3(kChainThreads), 0, s->stream, chain_args(s), prim, step, n_substeps, s->act, s->pv, s->pw, s->pgv);
    HIPCHK(hipGetLastError());
    return 0;
}

int plmpm_get_action_grad(plmpm_handle s, int n_steps, double* out) {
    NEED_BOUND(s);
    REQUIRE(out && n_steps >= 0 && n_steps <= s->F, "bad arguments");
    if (s->P == 0 || s->act_total == 0) return 0;
    std::vector<double> buf((size_t)n_steps * s->P * PLMPM_MAX_ACTION_DIM);
    HIPCHK(hipMemcpyAsync(buf.data(), s->act_a, buf.size() * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (int i = 0; i < n_steps; ++i)
        for (int p = 0; p < s->P; ++p)
            for (int k = 0; k < s->prims[p].action_dim; ++k)
                out[(size_t)i * s->act_total + s->act_ofs[p] + k] = buf[((size_t)i * s->P + p) * PLMPM_MAX_ACTION_DIM + k];
    return 0;
}

static void launch_fk_grad(plmpm_sim* s, int first, int n, int step) {
    const size_t lds = (size_t)n * kChainBwdWords * 8;
    if (lds <= kChainMaxLds) hipLaunchKernelGGL(k_fk_chain_grad<true>, dim3(s->P), dim3(kChainThreads), lds, s->stream, chain_args(s), first, n, step, chain_bufs(s));
    else hipLaunchKernelGGL(k_fk_chain_grad<false>, dim3(s->P), dim3(kChainThreads), 0, s->stream, chain_args(s), first, n, step, chain_bufs(s));
}
static int launch_fk(plmpm_sim* s, int first, int n) {
    if (s->P > 0) {
        const size_t lds = (size_t)n * kChainFwdWords * 8;
        if (lds <= kChainMaxLds) hipLaunchKernelGGL(k_fk_chain<true>, dim3(s->P), dim3(kChainThreads), lds, s->stream, chain_args(s), first, n, chain_bufs(s));
        else hipLaunchKernelGGL(k_fk_chain<false>, dim3(s->P), dim3(kChainThreads), 0, s->stream, chain_args(s), first, n, chain_bufs(s));
        // the primitives of substeps first .. first+n-1 as the grid kernels / fused-grid fills read them
        if (s->cfg.dtype == PLMPM_F64) build_prims_t<double>(s, first, n); else build_prims_t<float>(s, first, n);
    }
    return 0;
}

int plmpm_substep(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    REQUIRE(frame >= 0 && frame < s->F, "substep: frame %d out of range", frame);
    launch_fk(s, frame, 1);
    if (s->have_mats && s->mats_epoch != s->frame_epoch[frame]) DISPATCH(s, set_materials_t, s, s->frame_epoch[frame]);
    s->frame_epoch[frame + 1] = s->frame_epoch[frame];
    DISPATCH(s, substep_fwd, s, frame);
    HIPCHK(hipGetLastError());
    return 0;
}

int plmpm_step(plmpm_handle s, int first_frame, int n_substeps) {
    NEED_BOUND(s);
    REQUIRE(first_frame >= 0 && n_substeps > 0 && first_frame + n_substeps <= s->F, "step: frames [%d,%d] exceed max_frames %d",
            first_frame, first_frame + n_substeps, s->F);
    launch_fk(s, first_frame, n_substeps);
    // re-sort the step's first frame along the Hilbert curve (not the episode's first step: set_frame just sorted it)
    const int span = std::max(s->cfg.substeps, 1) * std::max(s->cfg.resort_steps, 1);       // frames per storage order
    int epoch = first_frame / span;
    bool sort_now = s->resort && first_frame > 0 && n_substeps == s->cfg.substeps && first_frame % span == 0 && epoch < s->n_epochs;
    if (s->resort && first_frame == 0 && s->steps_since_sort >= s->cfg.resort_steps && s->n_epochs >= 3) {
        // copy-mode episodes (Gym step(): every env step starts again at frame 0): same cadence, two alternating epochs
        epoch = s->frame_epoch[0] == 1 ? 2 : 1;
        sort_now = true;
    }
    if (sort_now && !s->prof_no_resort && s->frame_epoch[first_frame] != epoch) {
        if (DISPATCH(s, resort_frame_t, s, first_frame, epoch)) return -1;
        s->steps_since_sort = 0;
    }
    ++s->steps_since_sort;
    const int e = s->frame_epoch[first_frame];
    if (s->have_mats && s->mats_epoch != e) DISPATCH(s, set_materials_t, s, e);
    for (int f = first_frame + 1; f <= first_frame + n_substeps; ++f) s->frame_epoch[f] = e;      // before the launches: they size by epoch
    if (s->store && n_substeps > 1) DISPATCH(s, step_fwd_fused, s, first_frame, n_substeps);
    else for (int f = first_frame; f < first_frame + n_substeps; ++f) DISPATCH(s, substep_fwd, s, f);
    HIPCHK(hipGetLastError());
    return 0;
}
int plmpm_set_resort(plmpm_handle s, int on) {
    REQUIRE(s, "null handle");
    s->prof_no_resort = !on;
    return 0;
}

int plmpm_grad_begin(plmpm_handle s, int last_frame) {
    NEED_BOUND(s);
    NEED_FRAME(s, last_frame);
    HIPCHK(hipMemsetAsync(s->adj[0], 0, (size_t)2 * 24 * s->Npad * s->tsz, s->stream));
    size_t P1 = std::max(s->P, 1), F1 = s->F + 1;
    HIPCHK(hipMemsetAsync(s->ppos_a, 0, F1 * P1 * 3 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->prot_a, 0, F1 * P1 * 4 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->pv_a, 0, F1 * P1 * 3 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->pw_a, 0, F1 * P1 * 3 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->pgap_a, 0, F1 * P1 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->pgv_a, 0, F1 * P1 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->act_a, 0, F1 * P1 * PLMPM_MAX_ACTION_DIM * 8, s->stream));
    if (s->dist) {
        HIPCHK(hipMemsetAsync(s->ppos_l, 0, F1 * P1 * 3 * 8, s->stream));
        HIPCHK(hipMemsetAsync(s->prot_l, 0, F1 * P1 * 4 * 8, s->stream));
        HIPCHK(hipMemsetAsync(s->pgap_l, 0, F1 * P1 * 8, s->stream));
    }
    s->adj_frame[last_frame & 1] = last_frame;
    s->adj_frame[(last_frame + 1) & 1] = -1;
    s->adj_epoch[0] = s->adj_epoch[1] = s->frame_epoch[last_frame];     // all zero: any order
    return 0;
}

// Segment-checkpointed backward (plb/optimizer/long_term_gradient.ipynb cell 2, copy_and_clear): the adjoint held
// for frame `from_frame` (start of the later segment) becomes the adjoint of `to_frame` (end of the earlier
// segment, whose forward was just re-run); primitive pose adjoints move with it, every other pose / velocity /
// action adjoint is cleared.
__global__ void k_move_double(double* dst, const double* src, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
int plmpm_segment_carry(plmpm_handle s, int from_frame, int to_frame) {
    NEED_BOUND(s);
    NEED_FRAME(s, from_frame);
    NEED_FRAME(s, to_frame);
    REQUIRE(s->adj_frame[from_frame & 1] == from_frame, "segment_carry: adjoint of frame %d is not resident", from_frame);
    if ((from_frame & 1) != (to_frame & 1))
        HIPCHK(hipMemcpyAsync(s->adj[to_frame & 1], s->adj[from_frame & 1], (size_t)24 * s->Npad * s->tsz, hipMemcpyDeviceToDevice, s->stream));
    s->adj_epoch[to_frame & 1] = s->adj_epoch[from_frame & 1];
    s->adj_frame[to_frame & 1] = to_frame;
    s->adj_frame[(to_frame + 1) & 1] = -1;
    size_t P1 = std::max(s->P, 1), F1 = s->F + 1;
    if (s->P > 0) {
        double* tmp = s->staging;                       // 8 * P doubles of scratch (staging holds >= 24 * N)
        hipLaunchKernelGGL(k_move_double, dim3(1), dim3(64), 0, s->stream, tmp, s->ppos_a + (size_t)from_frame * s->P * 3, s->P * 3);
        hipLaunchKernelGGL(k_move_double, dim3(1), dim3(64), 0, s->stream, tmp + 32, s->prot_a + (size_t)from_frame * s->P * 4, s->P * 4);
        hipLaunchKernelGGL(k_move_double, dim3(1), dim3(64), 0, s->stream, tmp + 64, s->pgap_a + (size_t)from_frame * s->P, s->P);
        HIPCHK(hipMemsetAsync(s->ppos_a, 0, F1 * P1 * 3 * 8, s->stream));
        HIPCHK(hipMemsetAsync(s->prot_a, 0, F1 * P1 * 4 * 8, s->stream));
        HIPCHK(hipMemsetAsync(s->pgap_a, 0, F1 * P1 * 8, s->stream));
        hipLaunchKernelGGL(k_move_double, dim3(1), dim3(64), 0, s->stream, s->ppos_a + (size_t)to_frame * s->P * 3, tmp, s->P * 3);
        hipLaunchKernelGGL(k_move_double, dim3(1), dim3(64), 0, s->stream, s->prot_a + (size_t)to_frame * s->P * 4, tmp + 32, s->P * 4);
        hipLaunchKernelGGL(k_move_double, dim3(1), dim3(64), 0, s->stream, s->pgap_a + (size_t)to_frame * s->P, tmp + 64, s->P);
    }
    HIPCHK(hipMemsetAsync(s->pv_a, 0, F1 * P1 * 3 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->pw_a, 0, F1 * P1 * 3 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->pgv_a, 0, F1 * P1 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->act_a, 0, F1 * P1 * PLMPM_MAX_ACTION_DIM * 8, s->stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// reverse of the substep that starts at `frame`: bring the incoming adjoint (of frame+1) and the materials into the
// storage order of `frame`, then run it
static int bwd_prepare(plmpm_sim* s, int frame) {
    const int e = s->frame_epoch[frame], slot = (frame + 1) & 1;
    if (s->adj_epoch[slot] != e && DISPATCH(s, convert_adjoint_t, s, slot, s->adj_epoch[slot], e)) return -1;
    if (s->have_mats && s->mats_epoch != e) DISPATCH(s, set_materials_t, s, e);
    s->adj_epoch[frame & 1] = e;
    return 0;
}
int plmpm_substep_grad(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    REQUIRE(frame >= 0 && frame < s->F, "substep_grad: frame %d out of range", frame);
    REQUIRE(s->adj_frame[(frame + 1) & 1] == frame + 1, "substep_grad(%d): adjoint of frame %d is not resident (call grad_begin / go in reverse order)", frame, frame + 1);
    if (bwd_prepare(s, frame)) return -1;
    DISPATCH(s, substep_bwd, s, frame);
    HIPCHK(hipGetLastError());
    return 0;
}

int plmpm_step_grad(plmpm_handle s, int first_frame, int n_substeps, int step) {
    NEED_BOUND(s);
    REQUIRE(first_frame >= 0 && n_substeps > 0 && first_frame + n_substeps <= s->F, "step_grad: bad frame range");
    for (int f = first_frame + n_substeps - 1; f >= first_frame; --f) {
        REQUIRE(s->adj_frame[(f + 1) & 1] == f + 1, "step_grad: adjoint of frame %d is not resident", f + 1);
        if (bwd_prepare(s, f)) return -1;
        DISPATCH(s, substep_bwd, s, f);
    }
    if (s->P > 0) launch_fk_grad(s, first_frame, n_substeps, step);
    HIPCHK(hipGetLastError());
    return 0;
}

int plmpm_add_frame_grad(plmpm_handle s, int frame, const double* xa, const double* va, const double* Fa, const double* Ca) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->adj_frame[frame & 1] == frame, "add_frame_grad: adjoint of frame %d is not resident", frame);
    const size_t N = s->epochN[s->frame_epoch[frame]];
    if (s->adj_epoch[frame & 1] != s->frame_epoch[frame] &&
        DISPATCH(s, convert_adjoint_t, s, frame & 1, s->adj_epoch[frame & 1], s->frame_epoch[frame])) return -1;
    if (xa) HIPCHK(hipMemcpyAsync(s->staging, xa, N * 3 * 8, hipMemcpyHostToDevice, s->stream));
    if (va) HIPCHK(hipMemcpyAsync(s->staging + 3 * N, va, N * 3 * 8, hipMemcpyHostToDevice, s->stream));
    if (Fa) HIPCHK(hipMemcpyAsync(s->staging + 6 * N, Fa, N * 9 * 8, hipMemcpyHostToDevice, s->stream));
    if (Ca) HIPCHK(hipMemcpyAsync(s->staging + 15 * N, Ca, N * 9 * 8, hipMemcpyHostToDevice, s->stream));
    DISPATCH(s, adj_io_t, s, frame & 1, 1, xa != nullptr, va != nullptr, Fa != nullptr, Ca != nullptr);
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}
int plmpm_get_frame_grad(plmpm_handle s, int frame, double* xa, double* va, double* Fa, double* Ca) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->adj_frame[frame & 1] == frame, "get_frame_grad: adjoint of frame %d is not resident", frame);
    const size_t N = s->epochN[s->adj_epoch[frame & 1]];
    DISPATCH(s, adj_io_t, s, frame & 1, 0, 1, 1, 1, 1);
    if (xa) HIPCHK(hipMemcpyAsync(xa, s->staging, N * 3 * 8, hipMemcpyDeviceToHost, s->stream));
    if (va) HIPCHK(hipMemcpyAsync(va, s->staging + 3 * N, N * 3 * 8, hipMemcpyDeviceToHost, s->stream));
    if (Fa) HIPCHK(hipMemcpyAsync(Fa, s->staging + 6 * N, N * 9 * 8, hipMemcpyDeviceToHost, s->stream));
    if (Ca) HIPCHK(hipMemcpyAsync(Ca, s->staging + 15 * N, N * 9 * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

// ---- loss -----------------------------------------------------------------------------------

int plmpm_loss_set_target(plmpm_handle s, const double* density) {
    NEED_BOUND(s);
    REQUIRE(density, "null density");
    const size_t G = s->Gfull;                   // the sweeps run on the dense n^3 grid; only the window's part is kept
    double *d_dens, *d_sdf[2], *d_np[2];
    int* d_changed;
    HIPCHK(hipMalloc(&d_dens, G * 8));
    for (int i = 0; i < 2; ++i) { HIPCHK(hipMalloc(&d_sdf[i], G * 8)); HIPCHK(hipMalloc(&d_np[i], G * 24)); }
    HIPCHK(hipMalloc(&d_changed, 4));
    HIPCHK(hipMemcpyAsync(d_dens, density, G * 8, hipMemcpyHostToDevice, s->stream));
    std::vector<double> inf(G, 1000.0);
    HIPCHK(hipMemcpyAsync(d_sdf[0], inf.data(), G * 8, hipMemcpyHostToDevice, s->stream));     // target_sdf_copy.fill(inf)
    HIPCHK(hipMemsetAsync(d_np[0], 0, G * 24, s->stream));
    HIPCHK(hipMemsetAsync(d_np[1], 0, G * 24, s->stream));
    int cur = 0, last = 0;
    for (int it = 0; it < 2 * s->n; ++it) {                                                  // loss.py:103-106
        HIPCHK(hipMemsetAsync(d_changed, 0, 4, s->stream));
        // nearest_point persists across sweeps where nothing improves: carry the previous field over
        HIPCHK(hipMemcpyAsync(d_np[1 - cur], d_np[cur], G * 24, hipMemcpyDeviceToDevice, s->stream));
        hipLaunchKernelGGL(k_sdf_sweep, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s->stream, s->n, 1.0 / s->n, 1000.0,
                           d_dens, d_sdf[cur], d_np[cur], d_sdf[1 - cur], d_np[1 - cur], d_changed);
        int changed = 0;
        HIPCHK(hipMemcpyAsync(&changed, d_changed, 4, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        cur = 1 - cur;
        last = cur;
        if (!changed) break;
    }
    DISPATCH(s, upload_grid_t, s, d_dens, s->loss_td);
    DISPATCH(s, upload_grid_t, s, d_sdf[last], s->loss_ts);
    HIPCHK(hipStreamSynchronize(s->stream));
    hipFree(d_dens); hipFree(d_changed);
    for (int i = 0; i < 2; ++i) { hipFree(d_sdf[i]); hipFree(d_np[i]); }
    s->target_max = 0; s->target_sum = 0; s->target_outside = 0;
    const int n = s->n;
    for (size_t i = 0; i < G; ++i) {
        s->target_max = std::max(s->target_max, density[i]);
        s->target_sum += density[i];
        if (density[i] != 0.0) {
            // |grid_m - target| of an owned node outside the grid window is |0 - target|: a constant of the density loss
            const int k = (int)(i % n), j = (int)((i / n) % n), ii = (int)(i / ((size_t)n * n));
            const int nd[3] = {ii, j, k};
            bool inside = true;
            for (int d = 0; d < 3; ++d) inside &= nd[d] >= s->go[d] && nd[d] < s->go[d] + 4 * s->nbw[d];
            if (!inside && k >= s->cfg.slab_z0 && k < s->cfg.slab_z1) s->target_outside += std::fabs(density[i]);
        }
    }
    s->have_target = true;
    return 0;
}

int plmpm_loss_set_weights(plmpm_handle s, double sdf, double density, double contact, int soft_contact) {
    REQUIRE(s, "null handle");
    s->w_sdf = sdf; s->w_density = density; s->w_contact = contact; s->soft_contact = soft_contact;
    return 0;
}


int plmpm_loss_scatter(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    DISPATCH(s, loss_scatter_t, s, frame);
    HIPCHK(hipGetLastError());
    return 0;
}

int plmpm_loss_partials(plmpm_handle s, int frame, int phase, double* out32) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->have_target, "loss: no target density set");
    REQUIRE(out32 && (phase == 0 || phase == 1), "bad arguments");
    if (phase == 0) {
        if (loss_reset_scalars(s)) return -1;
        DISPATCH(s, loss_reduce_t, s);
        DISPATCH(s, loss_contact_pass_t, s, frame, s->soft_contact ? 1 : 0);
    } else {
        REQUIRE(s->soft_contact, "phase 1 only exists for the soft contact loss");
        DISPATCH(s, loss_contact_pass_t, s, frame, 2);
    }
    HIPCHK(hipMemcpyAsync(out32, s->lscal, LS_COUNT * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (phase == 0) out32[LS_DENSITY] += s->target_outside;
    return 0;
}

int plmpm_loss_set_globals(plmpm_handle s, const double* in32) {
    NEED_BOUND(s);
    REQUIRE(in32, "null argument");
    HIPCHK(hipMemcpyAsync(s->lscal, in32, LS_COUNT * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int plmpm_loss_finish(plmpm_handle s, const double* ls, double* out6) {
    REQUIRE(s && ls && out6, "null argument");
    double contact = 0;
    for (int p = 0; p < s->P; ++p)
        if (s->prims[p].action_dim > 0) contact += ls[LS_MIND + p] * ls[LS_MIND + p];          // loss.py:137-140
    double ma = ls[LS_MAXGM], mb = s->target_max;
    double I = ls[LS_DOT] / ma / mb, U = ls[LS_SUMGM] / ma + s->target_sum / mb;                 // loss.py:252-254
    out6[0] = contact * s->w_contact + ls[LS_DENSITY] * s->w_density + ls[LS_SDF] * s->w_sdf;   // loss.py:158-162
    out6[1] = ls[LS_SDF]; out6[2] = ls[LS_DENSITY]; out6[3] = contact; out6[4] = I / (U - I); out6[5] = 0;
    return 0;
}

int plmpm_loss_backward_local(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->have_target, "loss: no target density set");
    REQUIRE(s->adj_frame[frame & 1] == frame, "loss_backward: adjoint of frame %d is not resident", frame);
    if (s->adj_epoch[frame & 1] != s->frame_epoch[frame] &&
        DISPATCH(s, convert_adjoint_t, s, frame & 1, s->adj_epoch[frame & 1], s->frame_epoch[frame])) return -1;
    DISPATCH(s, loss_grad_t, s, frame);
    HIPCHK(hipGetLastError());
    return 0;
}

// single-rank composition of the phases above
static int loss_globals_single(plmpm_sim* s, int frame, double* ls) {
    if (plmpm_loss_scatter(s, frame)) return -1;
    if (plmpm_loss_partials(s, frame, 0, ls)) return -1;
    if (s->soft_contact) {
        if (plmpm_loss_set_globals(s, ls)) return -1;
        if (plmpm_loss_partials(s, frame, 1, ls)) return -1;
    }
    return 0;
}

int plmpm_loss_forward(plmpm_handle s, int frame, double* out6) {
    NEED_BOUND(s);
    REQUIRE(out6, "null output");
    double ls[LS_COUNT];
    if (loss_globals_single(s, frame, ls)) return -1;
    return plmpm_loss_finish(s, ls, out6);
}

int plmpm_loss_backward(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->have_target, "loss: no target density set");
    // recompute grid_m and the contact scalars (loss.py:210-237) -- all of it stays on the device: the adjoint needs the
    // mass grid and the per-primitive contact scalars, not the density / sdf sums, and nothing of it on the host
    if (plmpm_loss_scatter(s, frame)) return -1;
    if (loss_reset_scalars(s)) return -1;
    DISPATCH(s, loss_contact_pass_t, s, frame, s->soft_contact ? 1 : 0);
    if (s->soft_contact) DISPATCH(s, loss_contact_pass_t, s, frame, 2);
    return plmpm_loss_backward_local(s, frame);
}

int plmpm_get_grid_mass(plmpm_handle s, int frame, double* out) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(out, "null output");
    double* d_lin;
    HIPCHK(hipMalloc(&d_lin, s->Gfull * 8));
    DISPATCH(s, loss_scatter_t, s, frame);
    DISPATCH(s, download_grid_t, s, s->loss_gm, d_lin);
    HIPCHK(hipMemcpyAsync(out, d_lin, s->Gfull * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    hipFree(d_lin);
    return 0;
}
int plmpm_loss_get_target_sdf(plmpm_handle s, double* out) {
    NEED_BOUND(s);
    REQUIRE(out && s->have_target, "no target set");
    double* d_lin;
    HIPCHK(hipMalloc(&d_lin, s->Gfull * 8));
    DISPATCH(s, download_grid_t, s, s->loss_ts, d_lin);
    HIPCHK(hipMemcpyAsync(out, d_lin, s->Gfull * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    hipFree(d_lin);
    return 0;
}

int plmpm_grid_stats(plmpm_handle s, int frame, int64_t* active_nodes, int64_t* active_blocks) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    unsigned long long* d_out;
    unsigned long long h[2] = {0, 0};
    HIPCHK(hipMalloc(&d_out, 16));
    HIPCHK(hipMemsetAsync(d_out, 0, 16, s->stream));
    DISPATCH(s, grid_stats_t, s, frame, d_out);
    HIPCHK(hipMemcpyAsync(h, d_out, 16, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    hipFree(d_out);
    if (active_nodes) *active_nodes = (int64_t)h[0];
    if (active_blocks) *active_blocks = (int64_t)h[1];
    return 0;
}

int plmpm_tile_boxes(plmpm_handle s, int frame, int32_t* out, int max_workgroups, int* n_workgroups) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    const int nwg = nblocks_particles(s, frame);
    if (n_workgroups) *n_workgroups = nwg;
    if (!out) return 0;
    REQUIRE(max_workgroups >= nwg, "tile_boxes: room for %d workgroups, need %d", max_workgroups, nwg);
    std::vector<int> h((size_t)nwg * 8);
    HIPCHK(hipMemcpyAsync(h.data(), s->tiles + (size_t)frame * nwg * 8, h.size() * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (int w = 0; w < nwg; ++w)
        for (int k = 0; k < 6; ++k) out[w * 6 + k] = h[(size_t)w * 8 + k];
    return 0;
}

int plmpm_fk(plmpm_handle s, int first_frame, int n_substeps) {
    NEED_BOUND(s);
    REQUIRE(first_frame >= 0 && n_substeps > 0 && first_frame + n_substeps <= s->F, "fk: bad frame range");
    return launch_fk(s, first_frame, n_substeps);
}
int plmpm_p2g(plmpm_handle s, int frame, int chain) {
    REQUIRE(s && s->bound, "workspace not bound");
    REQUIRE(s->store, "the phase-split substep needs store_grid = 1");
    REQUIRE(frame >= 0 && frame < s->F, "p2g: frame out of range");
    if (chain) {
        REQUIRE(frame >= 1 && s->g2p_deferred == frame - 1, "p2g(%d, chain): frame %d's g2p was not deferred (deferred: %d)", frame, frame - 1, s->g2p_deferred);
        s->g2p_deferred = -1;
        DISPATCH(s, phase_g2p_p2g, s, frame);
    } else {
        REQUIRE(s->g2p_deferred < 0, "frame %d's g2p is deferred: call plmpm_p2g(frame + 1, chain = 1) next", s->g2p_deferred);
        DISPATCH(s, phase_p2g, s, frame);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
int plmpm_grid_interior(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    REQUIRE(s->store && frame >= 0 && frame < s->F, "grid_interior: bad call");
    REQUIRE(s->halo_in[PLMPM_HALO_GRID_IN].n > 0, "grid_interior: no halo planes are registered for PLMPM_HALO_GRID_IN (nothing to overlap with)");
    DISPATCH(s, phase_grid_g2p, s, frame, false, 1);
    HIPCHK(hipGetLastError());
    s->interior_fwd = frame;
    return 0;
}
int plmpm_grid_g2p(plmpm_handle s, int frame, int chain) {
    NEED_BOUND(s);
    REQUIRE(s->store && frame >= 0 && frame < s->F, "grid_g2p: bad call");
    REQUIRE(!chain || frame + 1 < s->F, "grid_g2p: only a substep with a successor chains");
    const int part = s->interior_fwd == frame ? 2 : 0;         // the interior blocks were done by plmpm_grid_interior
    s->interior_fwd = -1;
    DISPATCH(s, phase_grid_g2p, s, frame, chain != 0, part);
    HIPCHK(hipGetLastError());
    if (chain) s->g2p_deferred = frame;
    return 0;
}
int plmpm_grad_scatter(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    REQUIRE(s->store && frame >= 0 && frame < s->F, "grad_scatter: bad call");
    REQUIRE(s->dirty[frame], "grad_scatter(%d): the frame's grid is not resident (run the forward substep first)", frame);
    REQUIRE(s->adj_frame[(frame + 1) & 1] == frame + 1, "grad_scatter(%d): adjoint of frame %d is not resident", frame, frame + 1);
    REQUIRE(s->adj_epoch[(frame + 1) & 1] == s->frame_epoch[frame], "grad_scatter(%d): the adjoint of frame %d is in storage epoch %d, this substep ran in "
            "epoch %d (particles migrated at that frame: run plmpm_migrate_adjoint_begin / _finish first)", frame, frame + 1,
            s->adj_epoch[(frame + 1) & 1], s->frame_epoch[frame]);
    DISPATCH(s, phase_grad_scatter, s, frame);
    HIPCHK(hipGetLastError());
    return 0;
}
int plmpm_grad_gather_interior(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    REQUIRE(s->store && frame >= 0 && frame < s->F, "grad_gather_interior: bad call");
    REQUIRE(s->halo_in[PLMPM_HALO_GRID_OUT_ADJ].n > 0, "grad_gather_interior: no halo planes are registered for PLMPM_HALO_GRID_OUT_ADJ");
    DISPATCH(s, phase_grad_gather, s, frame, 1);
    HIPCHK(hipGetLastError());
    s->interior_bwd = frame;
    return 0;
}
int plmpm_grad_gather(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    REQUIRE(s->store && frame >= 0 && frame < s->F, "grad_gather: bad call");
    const int part = s->interior_bwd == frame ? 2 : 0;
    s->interior_bwd = -1;
    DISPATCH(s, phase_grad_gather, s, frame, part);
    HIPCHK(hipGetLastError());
    return 0;
}
int plmpm_chain_grad(plmpm_handle s, int first_frame, int n_substeps, int step) {
    NEED_BOUND(s);
    REQUIRE(first_frame >= 0 && n_substeps > 0 && first_frame + n_substeps <= s->F, "chain_grad: bad frame range");
    if (s->dist && s->P > 0) {          // fold this step's (already rank-summed) local pose adjoints into the global ones
        size_t np = (size_t)(n_substeps + 1) * s->P * 3, nr = (size_t)(n_substeps + 1) * s->P * 4;
        hipLaunchKernelGGL(k_merge_pose_adj, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s->stream,
                           s->ppos_a + (size_t)first_frame * s->P * 3, s->ppos_l + (size_t)first_frame * s->P * 3, np);
        hipLaunchKernelGGL(k_merge_pose_adj, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, s->stream,
                           s->prot_a + (size_t)first_frame * s->P * 4, s->prot_l + (size_t)first_frame * s->P * 4, nr);
        size_t ng = (size_t)(n_substeps + 1) * s->P;
        hipLaunchKernelGGL(k_merge_pose_adj, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s->stream,
                           s->pgap_a + (size_t)first_frame * s->P, s->pgap_l + (size_t)first_frame * s->P, ng);
    }
    if (s->P > 0) launch_fk_grad(s, first_frame, n_substeps, step);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- halos: zero-copy exchange of whole block planes ---------------------------------------------------------------
static int halo_field(plmpm_sim* s, int field, int frame, char** base, int* ncomp) {
    if (field == PLMPM_HALO_GRID_IN) {
        REQUIRE(s->store && frame >= 0 && frame < s->F, "halo: grid_in needs store_grid and a valid frame");
        *base = s->gstore + (size_t)frame * s->gstride; *ncomp = 4;
    } else if (field == PLMPM_HALO_GRID_OUT_ADJ) { *base = s->grid_out_adj; *ncomp = 3; }
    else if (field == PLMPM_HALO_LOSS_MASS) { *base = s->loss_gm; *ncomp = 1; }
    else return fail("unknown halo field %d", field);
    return 0;
}
int plmpm_grid_window(plmpm_handle s, int32_t* origin3, int32_t* blocks3) {
    REQUIRE(s && origin3 && blocks3, "null argument");
    for (int d = 0; d < 3; ++d) { origin3[d] = s->go[d]; blocks3[d] = s->nbw[d]; }
    return 0;
}
int plmpm_halo_region(plmpm_handle s, int field, int frame, int comp, int bz_a, int bz_b, void** dev_ptr, size_t* count) {
    NEED_BOUND(s);
    REQUIRE(dev_ptr && count, "null argument");
    char* base; int nc;
    if (halo_field(s, field, frame, &base, &nc)) return -1;
    REQUIRE(comp >= 0 && comp < nc, "halo_region: field %d has %d components", field, nc);
    const int ra = bz_a - s->go[2] / 4, rb = bz_b - s->go[2] / 4;       // block planes relative to the window
    REQUIRE(ra >= 0 && rb <= s->nbw[2] && ra < rb, "halo_region: block planes [%d,%d) outside this rank's grid window (planes [%d,%d))",
            bz_a, bz_b, s->go[2] / 4, s->go[2] / 4 + s->nbw[2]);
    const size_t plane = (size_t)s->nbw[0] * s->nbw[1] * 64;
    *dev_ptr = base + ((size_t)comp * s->G + (size_t)ra * plane) * s->tsz;
    *count = (size_t)(rb - ra) * plane;
    return 0;
}
int plmpm_halo_set_recv(plmpm_handle s, int field, int n_faces, const int* bz_a, const int* bz_b, void* const* recv) {
    REQUIRE(s, "null handle");
    REQUIRE(field >= 0 && field < 3 && n_faces >= 0 && n_faces <= 2, "halo_set_recv: a z-slab has at most 2 faces");
    HaloIn& H = s->halo_in[field];
    memset(&H, 0, sizeof H);
    for (int i = 0; i < n_faces; ++i) {
        const int ra = bz_a[i] - s->go[2] / 4, rb = bz_b[i] - s->go[2] / 4;
        REQUIRE(recv[i] && ra >= 0 && rb <= s->nbw[2] && ra < rb, "halo_set_recv: block planes [%d,%d) outside the grid window", bz_a[i], bz_b[i]);
        H.ba[i] = ra; H.bb[i] = rb; H.buf[i] = recv[i];
    }
    H.n = n_faces;
    return 0;
}
int plmpm_halo_apply(plmpm_handle s, int field, int frame) {
    NEED_BOUND(s);
    REQUIRE(field == PLMPM_HALO_LOSS_MASS, "halo_apply: the substep fields are added by grid_op / grid_op.grad themselves");
    char* base; int nc;
    if (halo_field(s, field, frame, &base, &nc)) return -1;
    const HaloIn& H = s->halo_in[field];
    const size_t plane = (size_t)s->nbw[0] * s->nbw[1] * 64;
    for (int i = 0; i < H.n; ++i) {
        const size_t cnt = (size_t)(H.bb[i] - H.ba[i]) * plane;
        for (int c = 0; c < nc; ++c) {
            char* dst = base + ((size_t)c * s->G + (size_t)H.ba[i] * plane) * s->tsz;
            const char* src = (const char*)H.buf[i] + (size_t)c * cnt * s->tsz;
            if (s->cfg.dtype == PLMPM_F64) hipLaunchKernelGGL((k_add_region<double>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s->stream, (double*)dst, (const double*)src, cnt);
            else hipLaunchKernelGGL((k_add_region<float>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s->stream, (float*)dst, (const float*)src, cnt);
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
// ---- particle migration between z-slabs (env-step boundaries) ----------------------------------------------------------
// A rank owns the particles whose stencil CENTRE node lies in its slab.  At the first frame of an env step the rows
// that left are packed and sent to the neighbour, the arrivals are merged in, and the whole set is re-sorted along
// the Hilbert curve into a new storage epoch (so this is also the slab engines' cell re-sort).  The reverse sweep
// sends the adjoint rows of the arrivals back where they came from.  Row = 28 doubles: global id, x(3), v(3), C(9),
// E(9), mu, lam, yield stress; adjoint row = 24 doubles.
constexpr int kMigRow = 28, kMigAdjRow = 24;
template <class T> __global__ void k_mig_classify(Dev<T> D, int f, int* dest, int* cnt, int* list0, int* list1, int maxlist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.N) return;
    const double* X = frame_x(D, f);
    const int cz = (int)(X[2 * (size_t)D.Npad + i] * (double)D.P.inv_dx - 0.5) + 1;
    const int d = cz < D.z0 ? 0 : (cz >= D.z1 ? 1 : -1);
    dest[i] = d;
    if (d >= 0) {
        const int k = atomicAdd(&cnt[d], 1);
        if (k < maxlist) (d == 0 ? list0 : list1)[k] = i;
    }
}
template <class T> __global__ void k_mig_pack(Dev<T> D, int f, const int* list, int n, const int* gid, double* out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int i = list[k], Np = D.Npad;
    const double* X = frame_x(D, f);
    const T* R = frame_r(D, f);
    double* r = out + (size_t)k * kMigRow;
    r[0] = (double)gid[i];
    for (int d = 0; d < 3; ++d) r[1 + d] = X[(size_t)d * Np + i];
    for (int d = 0; d < 21; ++d) r[4 + d] = (double)R[(size_t)d * Np + i];
    r[25] = (double)D.mu[i]; r[26] = (double)D.lam[i]; r[27] = (double)D.ys[i];
}
// sort keys of the candidates of the new frame: old slots [0, n_old) (leavers and padding sort last), then the arrivals
template <class T> __global__ void k_mig_keys(Dev<T> D, int f, int bits, const int* dest, int n_old, const double* in0, int n_in0, const double* in1,
                                              int n_in1, unsigned* keys, int* idx, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    idx[i] = i;
    const unsigned last = 1u << (3 * bits);
    double x[3];
    if (i < n_old) {
        if (dest[i] >= 0) { keys[i] = last; return; }
        const double* X = frame_x(D, f);
        for (int d = 0; d < 3; ++d) x[d] = X[(size_t)d * D.Npad + i];
    } else if (i < n_old + n_in0 + n_in1) {
        const int a = i - n_old;
        const double* r = a < n_in0 ? in0 + (size_t)a * kMigRow : in1 + (size_t)(a - n_in0) * kMigRow;
        for (int d = 0; d < 3; ++d) x[d] = r[1 + d];
    } else { keys[i] = last; return; }
    int b[3];
    for (int d = 0; d < 3; ++d) b[d] = min(max((int)(x[d] * (double)D.P.inv_dx - 0.5), 0), D.P.n - 1);
    keys[i] = hilbert_key_dev((unsigned)b[0], (unsigned)b[1], (unsigned)b[2], bits);
}
// the new frame (into `out`), its materials, ids and the slot map; v of the old frame is kept in the old order (vend)
template <class T> __global__ void k_mig_build(Dev<T> D, int f, int n_new, int n_old, const int* order, const double* in0, int n_in0, const double* in1,
                                               char* out, T* vend, const int* gid_old, int* gid_new, T* mats_new, int* src) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.Npad) return;
    const int Np = D.Npad;
    const double* X = frame_x(D, f);
    const T* R = frame_r(D, f);
    double* Xo = reinterpret_cast<double*>(out);
    T* Ro = reinterpret_cast<T*>(out + (size_t)3 * 8 * Np);
    for (int d = 0; d < 3; ++d) vend[(size_t)d * Np + i] = i < n_old ? R[(size_t)d * Np + i] : T(0);
    if (i >= n_new) {                               // padding rows: harmless values
        for (int d = 0; d < 3; ++d) Xo[(size_t)d * Np + i] = 0.5;
        for (int d = 0; d < 21; ++d) Ro[(size_t)d * Np + i] = T(0);
        for (int d = 0; d < 3; ++d) mats_new[(size_t)d * Np + i] = T(1);
        return;
    }
    const int j = order[i];
    if (j < n_old) {
        for (int d = 0; d < 3; ++d) Xo[(size_t)d * Np + i] = X[(size_t)d * Np + j];
        for (int d = 0; d < 21; ++d) Ro[(size_t)d * Np + i] = R[(size_t)d * Np + j];
        mats_new[i] = D.mu[j]; mats_new[(size_t)Np + i] = D.lam[j]; mats_new[2 * (size_t)Np + i] = D.ys[j];
        gid_new[i] = gid_old[j];
        src[i] = j;
    } else {
        const int a = j - n_old;
        const double* r = a < n_in0 ? in0 + (size_t)a * kMigRow : in1 + (size_t)(a - n_in0) * kMigRow;
        gid_new[i] = (int)r[0];
        for (int d = 0; d < 3; ++d) Xo[(size_t)d * Np + i] = r[1 + d];
        for (int d = 0; d < 21; ++d) Ro[(size_t)d * Np + i] = (T)r[4 + d];
        for (int d = 0; d < 3; ++d) mats_new[(size_t)d * Np + i] = (T)r[25 + d];
        src[i] = -1 - a;
    }
}
// reverse: adjoint rows of the arrivals -> back buffers; rows of the stayers -> their old slots
template <class T> __global__ void k_mig_adj_split(const T* adj, T* tmp, int Np, int n_new, const int* src, int n_in0, double* back0, double* back1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_new) return;
    const int j = src[i];
    if (j >= 0) {
        for (int c = 0; c < kMigAdjRow; ++c) tmp[(size_t)c * Np + j] = adj[(size_t)c * Np + i];
    } else {
        const int a = -1 - j;
        double* r = a < n_in0 ? back0 + (size_t)a * kMigAdjRow : back1 + (size_t)(a - n_in0) * kMigAdjRow;
        for (int c = 0; c < kMigAdjRow; ++c) r[c] = (double)adj[(size_t)c * Np + i];
    }
}
template <class T> __global__ void k_mig_adj_recv(T* tmp, int Np, const int* list, int n, const double* rows) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int i = list[k];
    for (int c = 0; c < kMigAdjRow; ++c) tmp[(size_t)c * Np + i] = (T)rows[(size_t)k * kMigAdjRow + c];
}

template <class T> static int migrate_begin_t(plmpm_sim* s, int frame, int epoch_new) {
    Dev<T> D = make_dev<T>(s, frame);
    int* leave = s->mig_leave + (size_t)epoch_new * s->Npad;
    HIPCHK(hipMemsetAsync(s->mig_cnt, 0, 8, s->stream));
    if (D.N > 0)
        hipLaunchKernelGGL((k_mig_classify<T>), dim3((D.N + 255) / 256), dim3(256), 0, s->stream, D, frame, s->mig_dest, s->mig_cnt, leave,
                           leave + s->mig_max_rows, s->mig_max_rows);
    int cnt[2];
    HIPCHK(hipMemcpyAsync(cnt, s->mig_cnt, 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    REQUIRE(cnt[0] <= s->mig_max_rows && cnt[1] <= s->mig_max_rows, "migrate: %d / %d rows leave at once, room for %d per direction", cnt[0], cnt[1], s->mig_max_rows);
    if (s->det)
        // the lists were filled through an atomic cursor: put them in slot order, so that the rows leave -- and arrive,
        // and tie-break the neighbour's stable re-sort -- in the same order in every run
        for (int d = 0; d < 2; ++d)
            if (cnt[d] > 1) {
                int* list = leave + (size_t)d * s->mig_max_rows;
                HIPCHK(hipMemcpyAsync(s->skey[0], list, (size_t)cnt[d] * 4, hipMemcpyDeviceToDevice, s->stream));
                if (plmpm_sort_pairs(s->sort_tmp, s->sort_tmp_bytes, s->skey[0], s->skey[1], s->sidx[0], s->sidx[1], cnt[d], 32, s->stream) != 0)
                    return fail("migrate: device sort failed");
                HIPCHK(hipMemcpyAsync(list, s->skey[1], (size_t)cnt[d] * 4, hipMemcpyDeviceToDevice, s->stream));
            }
    const int* gid = s->gid_store + (size_t)s->frame_epoch[frame] * s->Npad;
    for (int d = 0; d < 2; ++d)
        if (cnt[d] > 0)
            hipLaunchKernelGGL((k_mig_pack<T>), dim3((cnt[d] + 255) / 256), dim3(256), 0, s->stream, D, frame, leave + (size_t)d * s->mig_max_rows, cnt[d], gid,
                               s->mig_send[d]);
    s->mig_pending_out[0] = cnt[0]; s->mig_pending_out[1] = cnt[1];
    return 0;
}
template <class T> static int migrate_finish_t(plmpm_sim* s, int frame, int e_new, int n_in0, const double* in0, int n_in1, const double* in1) {
    Dev<T> D = make_dev<T>(s, frame);
    const int e_old = s->frame_epoch[frame], n_old = D.N;
    const int n_new = n_old - s->mig_pending_out[0] - s->mig_pending_out[1] + n_in0 + n_in1;
    REQUIRE(n_new >= 0 && n_new <= s->Npad, "migrate: %d particles after the exchange, capacity %d (raise particle_capacity)", n_new, s->Npad);
    const int total = n_old + n_in0 + n_in1;
    REQUIRE(total <= s->sort_cap, "migrate: %d candidate rows, room for %d", total, s->sort_cap);
    // the reverse sweep packs the adjoint rows of these arrivals into this rank's own send buffers (mig_max_rows rows each)
    REQUIRE(n_in0 <= s->mig_max_rows && n_in1 <= s->mig_max_rows, "migrate: %d / %d rows arrive at once, the row buffers hold %d (raise particle_capacity)",
            n_in0, n_in1, s->mig_max_rows);
    int bits = 1;
    while ((1 << bits) < s->n) ++bits;
    if (total > 0) {
        hipLaunchKernelGGL((k_mig_keys<T>), dim3((total + 255) / 256), dim3(256), 0, s->stream, D, frame, bits, s->mig_dest, n_old, in0, n_in0, in1, n_in1,
                           s->skey[0], s->sidx[0], total);
        if (plmpm_sort_pairs(s->sort_tmp, s->sort_tmp_bytes, s->skey[0], s->skey[1], s->sidx[0], s->sidx[1], total, 3 * bits + 1, s->stream) != 0)
            return fail("migrate: device sort failed");
    }
    T* vend = (T*)(s->vend + (size_t)e_new * 3 * s->Npad * s->tsz);
    T* mats_new = (T*)(s->mats_store + (size_t)e_new * 3 * s->Npad * s->tsz);
    hipLaunchKernelGGL((k_mig_build<T>), dim3(s->Npad / 256), dim3(256), 0, s->stream, D, frame, n_new, n_old, s->sidx[1], in0, n_in0, in1, s->frame_tmp, vend,
                       s->gid_store + (size_t)e_old * s->Npad, s->gid_store + (size_t)e_new * s->Npad, mats_new, s->mig_src + (size_t)e_new * s->Npad);
    HIPCHK(hipMemcpyAsync(s->state + (size_t)frame * s->frame_bytes, s->frame_tmp, s->frame_bytes, hipMemcpyDeviceToDevice, s->stream));
    plmpm_sim::MigInfo& m = s->mig[e_new];
    m.parent = e_old; m.nout[0] = s->mig_pending_out[0]; m.nout[1] = s->mig_pending_out[1]; m.nin[0] = n_in0; m.nin[1] = n_in1;
    s->epochN[e_new] = n_new;
    s->frame_epoch[frame] = e_new;
    s->mats_epoch = e_new;
    return 0;
}
template <class T> static int migrate_adjoint_begin_t(plmpm_sim* s, int frame) {
    const int slot = frame & 1, e = s->adj_epoch[slot];
    const plmpm_sim::MigInfo& m = s->mig[e];
    T* tmp = (T*)s->frame_tmp;
    HIPCHK(hipMemsetAsync(tmp, 0, (size_t)kMigAdjRow * s->Npad * s->tsz, s->stream));
    if (s->epochN[e] > 0)
        hipLaunchKernelGGL((k_mig_adj_split<T>), dim3((s->epochN[e] + 255) / 256), dim3(256), 0, s->stream, (const T*)s->adj[slot], tmp, s->Npad, s->epochN[e],
                           s->mig_src + (size_t)e * s->Npad, m.nin[0], s->mig_send[0], s->mig_send[1]);
    return 0;
}
template <class T> static int migrate_adjoint_finish_t(plmpm_sim* s, int frame, const double* rows0, const double* rows1) {
    const int slot = frame & 1, e = s->adj_epoch[slot];
    const plmpm_sim::MigInfo& m = s->mig[e];
    T* tmp = (T*)s->frame_tmp;
    const int* leave = s->mig_leave + (size_t)e * s->Npad;
    const double* rows[2] = {rows0, rows1};
    for (int d = 0; d < 2; ++d)
        if (m.nout[d] > 0) {
            REQUIRE(rows[d], "migrate_adjoint_finish: %d rows went %s at this boundary, their adjoints are missing", m.nout[d], d ? "up" : "down");
            hipLaunchKernelGGL((k_mig_adj_recv<T>), dim3((m.nout[d] + 255) / 256), dim3(256), 0, s->stream, tmp, s->Npad, leave + (size_t)d * s->mig_max_rows,
                               m.nout[d], rows[d]);
        }
    HIPCHK(hipMemcpyAsync(s->adj[slot], tmp, (size_t)kMigAdjRow * s->Npad * s->tsz, hipMemcpyDeviceToDevice, s->stream));
    s->adj_epoch[slot] = m.parent;
    return 0;
}

extern "C" {
int plmpm_set_ids(plmpm_handle s, const int32_t* ids) {
    REQUIRE(s && ids, "null argument");
    REQUIRE(s->dist, "set_ids: global particle ids only exist on slab engines");
    s->ids0.assign(ids, ids + s->N);
    return 0;
}
int plmpm_frame_info(plmpm_handle s, int frame, int32_t* count, int32_t* epoch, int32_t* adjoint_epoch) {
    REQUIRE(s, "null handle");
    NEED_FRAME(s, frame);
    const int e = s->frame_epoch[frame];
    if (count) *count = s->epochN[e];
    if (epoch) *epoch = e;
    if (adjoint_epoch) *adjoint_epoch = s->adj_frame[frame & 1] == frame ? s->adj_epoch[frame & 1] : -1;
    return 0;
}
int plmpm_set_population(plmpm_handle s, int n_rows) {
    NEED_BOUND(s);
    REQUIRE(s->dist, "set_population: slab engines only (a single-GPU engine keeps its particle count)");
    REQUIRE(n_rows >= 0 && n_rows <= s->Npad, "set_population: %d rows, capacity %d (raise particle_capacity)", n_rows, s->Npad);
    s->N = n_rows;
    s->perm.resize(n_rows);
    for (int i = 0; i < n_rows; ++i) s->perm[i] = i;
    s->ids0.assign(n_rows, 0);
    s->have_mats = false;                       // the caller sets ids, frame 0 (resort) and materials of the new population next
    return 0;
}
int plmpm_get_materials(plmpm_handle s, int frame, double* mu, double* lam, double* ys) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(mu && lam && ys, "null argument");
    REQUIRE(s->have_mats, "get_materials: no materials were set");
    const int e = s->frame_epoch[frame];
    const size_t n = s->epochN[e];
    if (!(s->dist && e > 0)) {                 // rows in caller order: the master copy
        HIPCHK(hipMemcpyAsync(mu, s->mats_master, n * 8, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(lam, s->mats_master + s->N, n * 8, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(ys, s->mats_master + 2 * (size_t)s->N, n * 8, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        return 0;
    }
    // a migrated epoch of a slab engine: the materials travelled with the rows, storage order, engine scalar type
    std::vector<char> h((size_t)3 * s->Npad * s->tsz);
    HIPCHK(hipMemcpyAsync(h.data(), s->mats_store + (size_t)e * 3 * s->Npad * s->tsz, h.size(), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    double* out[3] = {mu, lam, ys};
    for (int c = 0; c < 3; ++c)
        for (size_t i = 0; i < n; ++i)
            out[c][i] = s->tsz == 8 ? ((const double*)h.data())[(size_t)c * s->Npad + i] : (double)((const float*)h.data())[(size_t)c * s->Npad + i];
    return 0;
}
int plmpm_adjoint_rows(plmpm_handle s, int frame, int32_t* rows) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(rows, "null argument");
    REQUIRE(s->adj_frame[frame & 1] == frame, "adjoint_rows: adjoint of frame %d is not resident", frame);
    *rows = s->epochN[s->adj_epoch[frame & 1]];
    return 0;
}
int plmpm_get_ids(plmpm_handle s, int frame, int32_t* ids) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(ids && s->dist, "get_ids: slab engines only");
    const int e = s->frame_epoch[frame];
    if (e == 0) { memcpy(ids, s->ids0.data(), (size_t)s->N * 4); return 0; }       // epoch 0: rows are in caller order
    HIPCHK(hipMemcpyAsync(ids, s->gid_store + (size_t)e * s->Npad, (size_t)s->epochN[e] * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}
int plmpm_migrate_begin(plmpm_handle s, int frame, int32_t* out2, void** rows_down, void** rows_up) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->dist && out2 && rows_down && rows_up, "migrate_begin: slab engines only; null argument");
    REQUIRE(s->mig_pending_frame < 0, "migrate_begin: the migration of frame %d is still open", s->mig_pending_frame);
    // tape mode: a fresh epoch per migration; copy mode (frame 0 over and over): two alternating epochs
    int e_new = frame == 0 ? (s->frame_epoch[0] == 1 ? 2 : 1) : std::max(s->next_epoch, 3);
    REQUIRE(e_new < s->n_epochs, "migrate: out of storage epochs (%d)", s->n_epochs);
    if (DISPATCH(s, migrate_begin_t, s, frame, e_new)) return -1;
    HIPCHK(hipGetLastError());
    s->mig_pending_frame = frame;
    out2[0] = s->mig_pending_out[0]; out2[1] = s->mig_pending_out[1];
    *rows_down = s->mig_send[0]; *rows_up = s->mig_send[1];
    return 0;
}
int plmpm_migrate_finish(plmpm_handle s, int frame, int n_in_down, const void* rows_down, int n_in_up, const void* rows_up, int32_t* new_count) {
    NEED_BOUND(s);
    REQUIRE(s->mig_pending_frame == frame, "migrate_finish(%d): call migrate_begin on that frame first", frame);
    REQUIRE(n_in_down >= 0 && n_in_up >= 0 && (n_in_down == 0 || rows_down) && (n_in_up == 0 || rows_up), "migrate_finish: bad arrival lists");
    const int e_new = frame == 0 ? (s->frame_epoch[0] == 1 ? 2 : 1) : std::max(s->next_epoch, 3);
    if (DISPATCH(s, migrate_finish_t, s, frame, e_new, n_in_down, (const double*)rows_down, n_in_up, (const double*)rows_up)) return -1;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));          // the caller may reuse its receive buffers
    if (frame != 0) s->next_epoch = e_new + 1;
    s->mig_pending_frame = -1;
    if (new_count) *new_count = s->epochN[e_new];
    return 0;
}
int plmpm_migrate_adjoint_begin(plmpm_handle s, int frame, int32_t* send2, int32_t* recv2, void** rows_down, void** rows_up) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->dist && send2 && recv2 && rows_down && rows_up, "migrate_adjoint_begin: slab engines only; null argument");
    REQUIRE(s->adj_frame[frame & 1] == frame, "migrate_adjoint_begin: adjoint of frame %d is not resident", frame);
    const int e = s->adj_epoch[frame & 1];
    REQUIRE(e > 0 && e == s->frame_epoch[frame], "migrate_adjoint_begin: frame %d did not migrate into the epoch its adjoint is in", frame);
    if (DISPATCH(s, migrate_adjoint_begin_t, s, frame)) return -1;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    send2[0] = s->mig[e].nin[0]; send2[1] = s->mig[e].nin[1];           // adjoints of the arrivals go back where they came from
    recv2[0] = s->mig[e].nout[0]; recv2[1] = s->mig[e].nout[1];         // ... and those of the rows that left come home
    *rows_down = s->mig_send[0]; *rows_up = s->mig_send[1];
    return 0;
}
int plmpm_migrate_adjoint_finish(plmpm_handle s, int frame, const void* rows_down, const void* rows_up) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->dist && s->adj_frame[frame & 1] == frame, "migrate_adjoint_finish: bad call");
    if (DISPATCH(s, migrate_adjoint_finish_t, s, frame, (const double*)rows_down, (const double*)rows_up)) return -1;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}
int plmpm_pose_grad_region(plmpm_handle s, int first_frame, int n_frames, void** pos_adj, size_t* pos_count, void** rot_adj,
                           size_t* rot_count, void** gap_adj, size_t* gap_count) {
    NEED_BOUND(s);
    REQUIRE(first_frame >= 0 && n_frames > 0 && first_frame + n_frames <= s->F + 1, "pose_grad_region: bad frame range");
    double* pa = s->dist ? s->ppos_l : s->ppos_a;
    double* ra = s->dist ? s->prot_l : s->prot_a;
    *pos_adj = pa + (size_t)first_frame * s->P * 3; *pos_count = (size_t)n_frames * s->P * 3;
    *rot_adj = ra + (size_t)first_frame * s->P * 4; *rot_count = (size_t)n_frames * s->P * 4;
    if (gap_adj && gap_count) {
        *gap_adj = (s->dist ? s->pgap_l : s->pgap_a) + (size_t)first_frame * s->P; *gap_count = (size_t)n_frames * s->P;
    }
    return 0;
}
int plmpm_action_grad_region(plmpm_handle s, void** dev_ptr, size_t* count) {
    NEED_BOUND(s);
    *dev_ptr = s->act_a; *count = (size_t)(s->F + 1) * std::max(s->P, 1) * PLMPM_MAX_ACTION_DIM;
    return 0;
}
int plmpm_debug_counters(plmpm_handle s, int* out4) {
    NEED_BOUND(s);
    HIPCHK(hipMemcpyAsync(out4, s->err_d, 16, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemsetAsync(s->err_d + 1, 0, 12, s->stream));
    return 0;
}
#ifdef PLB_PHASE_TIMING
extern "C" int plmpm_debug_trace(plmpm_handle s, unsigned long long* out, size_t n) {       // profiling builds only
    NEED_BOUND(s);
    HIPCHK(hipMemcpyAsync(out, s->staging, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}
#endif
int plmpm_check_error(plmpm_handle s, int* flags) {
    NEED_BOUND(s);
    REQUIRE(flags, "null argument");
    HIPCHK(hipMemcpyAsync(flags, s->err_d, 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (*flags) HIPCHK(hipMemsetAsync(s->err_d, 0, 4, s->stream));
    return 0;
}

int plmpm_profile_enable(plmpm_handle s, int on) {
    REQUIRE(s, "null handle");
    s->prof = on != 0;
    s->ev_used.clear();
    s->ev_next = 0;
    return 0;
}
int plmpm_profile_kernel_count(void) { return K_COUNT; }
const char* plmpm_profile_kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : ""; }
int plmpm_profile_read(plmpm_handle s, double* total_ms, int64_t* launches) {
    REQUIRE(s && total_ms && launches, "null argument");
    HIPCHK(hipStreamSynchronize(s->stream));
    for (int i = 0; i < K_COUNT; ++i) { total_ms[i] = 0; launches[i] = 0; }
    for (auto& u : s->ev_used) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, s->ev_pool[u.second], s->ev_pool[u.second + 1]));
        total_ms[u.first] += ms;
        launches[u.first] += 1;
    }
    s->ev_used.clear();
    s->ev_next = 0;
    return 0;
}

int plmpm_get_order(plmpm_handle s, int32_t* perm) {
    REQUIRE(s && perm, "null argument");
    memcpy(perm, s->perm.data(), (size_t)s->N * 4);
    return 0;
}

}  // extern "C"
