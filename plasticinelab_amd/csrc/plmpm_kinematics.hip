// Actions, the serial primitive kinematics chain (forward_kinematics / set_velocity and their adjoints,
// primive_base.py:117-121,184-192) and per-primitive queries of the C ABI.
#include "plmpm_internal.h"

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
    PLB_DYN_LDS(double, sm);
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
    PLB_DYN_LDS(double, sm);
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
template <class T> static int build_prims_t(plmpm_sim* s, int first, int n) {
    Dev<T> D = make_dev<T>(s);
    hipLaunchKernelGGL((k_build_prims<T>), dim3((n * s->P + 63) / 64), dim3(64), 0, s->stream, D, first, n, (PrimT<T>*)s->ptab);
    return 0;
}

extern "C" {
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

void plmpm_launch_fk_grad(plmpm_sim* s, int first, int n, int step) {
    const size_t lds = (size_t)n * kChainBwdWords * 8;
    if (lds <= kChainMaxLds) hipLaunchKernelGGL(k_fk_chain_grad<true>, dim3(s->P), dim3(kChainThreads), lds, s->stream, chain_args(s), first, n, step, chain_bufs(s));
    else hipLaunchKernelGGL(k_fk_chain_grad<false>, dim3(s->P), dim3(kChainThreads), 0, s->stream, chain_args(s), first, n, step, chain_bufs(s));
}
int plmpm_launch_fk(plmpm_sim* s, int first, int n) {
    if (s->P > 0) {
        const size_t lds = (size_t)n * kChainFwdWords * 8;
        if (lds <= kChainMaxLds) hipLaunchKernelGGL(k_fk_chain<true>, dim3(s->P), dim3(kChainThreads), lds, s->stream, chain_args(s), first, n, chain_bufs(s));
        else hipLaunchKernelGGL(k_fk_chain<false>, dim3(s->P), dim3(kChainThreads), 0, s->stream, chain_args(s), first, n, chain_bufs(s));
        // the primitives of substeps first .. first+n-1 as the grid kernels / fused-grid fills read them
        if (s->cfg.dtype == PLMPM_F64) build_prims_t<double>(s, first, n); else build_prims_t<float>(s, first, n);
    }
    return 0;
}

int plmpm_fk(plmpm_handle s, int first_frame, int n_substeps) {
    NEED_BOUND(s);
    REQUIRE(first_frame >= 0 && n_substeps > 0 && first_frame + n_substeps <= s->F, "fk: bad frame range");
    return plmpm_launch_fk(s, first_frame, n_substeps);
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
    if (s->P > 0) plmpm_launch_fk_grad(s, first_frame, n_substeps, step);
    HIPCHK(hipGetLastError());
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
}  // extern "C"
