// Device-side halo exchange for z-slab ranks: a rank WRITES its copy of the exchanged block planes straight into the
// neighbour's receive area (fine-grained device memory the neighbour allocated and this rank mapped through an IPC
// handle -- across xGMI on a multi-GPU node, inside one GPU when test ranks share it), publishes an arrival counter
// behind the data, and waits for the neighbour's counter -- all inside ONE kernel on the engine's stream.  The host
// only enqueues: no torch.distributed call, no request object and no stream hand-over per substep, so the slab substep
// loops themselves (plmpm_slab_step / plmpm_slab_step_grad) are native and the Python driver is out of them.
//
//   receive area of a face:  [256 B: arrival counter][half 0][half 1],  half = [one validity word per 4^3 block, padded to
//                            256 B][ncomp x count scalars]
//
// Only the blocks that carry something are sent: a wave per block of the exchange planes looks at the block's activity flag
// (the flag the scatter kernels set), writes the validity word, and copies the block's 64 nodes per component if it is set.
//
// Visibility between DIFFERENT GPUs (round 4; every run before it had all ranks on one GPU, i.e. behind the same L2s): both
// sides of the hand-off are at system scope.  The writer stores through its caches (sc0 sc1) and drains vmcnt before the
// counter moves; the receive areas are UNCACHED device memory where the runtime offers it (hipDeviceMallocUncached,
// fine-grained otherwise), and the readers -- the polling lane here, halo_sent / halo_value in the grid kernels -- use
// system-scope loads (sc0 sc1: served by memory, never by a stale L1 / L2 line of a half that was read two exchanges ago).
// MI355X_MICROARCH.md lists "sc0 sc1 stores and loads on both sides" as a valid form of the hand-off on its own; "the
// next launch is the acquire" is no longer relied upon.
// The body's cross-section covers about a quarter of the xy window's blocks at config 3, so about a quarter of the plane
// crosses the link (a 24 x 18-block window: 442 KB per block plane and face dense, ~110 KB as sent).
//
// Exchange k of a field writes half k & 1.  Why two halves are enough: rank A starts exchange k + 1 only behind its grid
// kernel of exchange k, which ran behind A's wait for the neighbour's arrival k, which the neighbour published behind
// ITS grid kernel of exchange k - 1 -- the last reader of half (k + 1) & 1 there.
#include "plmpm_internal.h"

namespace {

constexpr size_t kPeerHeader = 256;

// (PeerXchg, store_through and the publish / wait halves live in plmpm_kernels.h: the grid kernels use them too)

template <class T>
__global__ __launch_bounds__(256) void k_halo_xchg(PeerXchg X) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
    for (int i = 0; i < X.n; ++i) {
        int* valid = (int*)X.dst[i];
        T* data = (T*)(X.dst[i] + X.data_ofs[i]);
        const size_t count = (size_t)X.nblk[i] << 6;
        // a wave owns blocks wave, wave + nwave, ... of the face's planes (at most 64: the launch is sized for that).  Lane k
        // fetches the activity flag of the wave's k-th block and writes its validity word; the set bits are the blocks to copy
        const int bl = wave + lane * nwave;
        int on = 0;
        if (bl < X.nblk[i]) {
            const int blk = X.blk0[i] + bl;
            on = X.flags ? (X.flags[(blk & ((1 << X.fgl) - 1)) * X.fs + (blk >> X.fgl)] != 0) : 1;      // flag_slot
            store_through(valid + bl, on);
        }
        unsigned long long todo = __ballot(on);
        while (todo) {                                   // two blocks per trip: their loads fly together
            const int k0 = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int k1 = todo ? __ffsll((long long)todo) - 1 : -1;
            if (k1 >= 0) todo &= todo - 1;
            const int b0 = wave + k0 * nwave, b1 = k1 >= 0 ? wave + k1 * nwave : b0;
            T v0[4], v1[4];
            for (int c = 0; c < X.ncomp; ++c) {
                v0[c] = ((const T*)X.src[c])[((size_t)(X.blk0[i] + b0) << 6) + lane];
                v1[c] = ((const T*)X.src[c])[((size_t)(X.blk0[i] + b1) << 6) + lane];
                if (i == 0 && X.spoil != 1.0f) { v0[c] *= (T)X.spoil; v1[c] *= (T)X.spoil; }
            }
            for (int c = 0; c < X.ncomp; ++c) {
                store_through(data + (size_t)c * count + ((size_t)b0 << 6) + lane, v0[c]);
                if (k1 >= 0) store_through(data + (size_t)c * count + ((size_t)b1 << 6) + lane, v1[c]);
            }
        }
    }
    // every wave waits for the acknowledgements of its own stores; the workgroup then counts itself done
    wait_vmem();
    __syncthreads();
    __shared__ int last;
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(X.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    // the last workgroup: every copy of this launch has landed.  Publish, then wait for the neighbours (one lane per face).
    if (threadIdx.x == 0) {
        __hip_atomic_store(X.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next launch is stream-ordered behind this one
        for (int i = 0; i < X.n; ++i) store_through(X.arrive_remote[i], X.seq);
    }
    xchg_wait_lanes(X);
    // the grid kernel that consumes the received planes is a later launch on this stream and reads them with system-scope
    // loads (halo_sent / halo_value, plmpm_kernels.h)
}

int peer_init(plmpm_sim* s) {
    if (s->peer_done) return 0;
    HIPCHK(hipMalloc((void**)&s->peer_done, 256));
    HIPCHK(hipMemsetAsync(s->peer_done, 0, 256, s->stream));
    HIPCHK(hipHostMalloc((void**)&s->peer_status, 64, hipHostMallocMapped));
    *s->peer_status = 0;
    return 0;
}

// First contact with a neighbour (plmpm_peer_ping): one lane per face stores a token THROUGH the caches into word 16 of the
// neighbour's receive-area header (the arrival counter is word 0; the rest of the 256 B is otherwise unused) and polls word 16 of
// its own header with system-scope loads until the neighbour's token shows up -- the exchange kernels' own hand-off pattern
// (profiles/microbench/peer_xchg.hip) on the real areas, before any halo depends on it.  res[2 i] = last value seen on face i,
// res[2 i + 1] = wait in 10 ns ticks (-1: timed out).
__global__ void k_peer_ping(int n, unsigned* remote0, unsigned* remote1, unsigned* local0, unsigned* local1, unsigned token,
                            long long timeout_ticks, long long* res) {
    const int i = threadIdx.x;
    if (i >= n) return;
    unsigned* const theirs = i == 0 ? remote0 : remote1;
    unsigned* const mine = i == 0 ? local0 : local1;
    store_through(theirs + 16, token);
    wait_vmem();
    const long long t0 = wall_clock64();
    unsigned got = 0;
    long long waited = -1;
    for (;;) {
        got = __hip_atomic_load(mine + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t = wall_clock64() - t0;
        if (got == token) { waited = t; break; }
        if (t > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(2);
    }
    res[2 * i] = (long long)got;
    res[2 * i + 1] = waited;
}

double peer_timeout_seconds() {
    const char* e = getenv("PLMPM_PEER_TIMEOUT");
    const double v = e ? atof(e) : 20.0;
    return v > 0 ? v : 20.0;
}

}  // namespace

extern "C" {

// ---- memory a neighbour can map ---------------------------------------------------------------------------------------
int plmpm_peer_area_bytes(plmpm_handle s, int field, int bz_a, int bz_b, size_t* bytes) {
    NEED_BOUND(s);
    REQUIRE(bytes, "null argument");
    char* base; int nc;
    if (plmpm_halo_field(s, field, 0, &base, &nc)) return -1;
    REQUIRE(bz_a < bz_b, "peer_area_bytes: empty range of block planes");
    const size_t nblk = (size_t)(bz_b - bz_a) * s->nbw[0] * s->nbw[1];
    *bytes = kPeerHeader + 2 * (align_up(nblk * 4, 256) + (size_t)nc * (nblk << 6) * s->tsz);
    return 0;
}
int plmpm_peer_alloc(plmpm_handle s, size_t bytes, void** dev_ptr, void* ipc_handle64) {
    REQUIRE(s && dev_ptr && ipc_handle64 && bytes > 0, "peer_alloc: bad argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handles travel as 64 bytes");
    if (peer_init(s)) return -1;
    void* p = nullptr;
    // uncached device memory: no L2 of this GPU ever holds a line of it, so what a neighbour (another XCD, another process,
    // another GPU over xGMI) wrote is what the next load returns.  Fine-grained memory -- cached, coherent at system scope
    // for accesses that ask for it -- where the runtime refuses the flag; the readers use system-scope loads either way.
    // PLMPM_PEER_MEM=finegrained forces the latter (A/B of the two, profiles/microbench/peer_xchg.hip).
    const char* pm = getenv("PLMPM_PEER_MEM");
    hipError_t me = hipErrorUnknown;
    if (!(pm && !strcmp(pm, "finegrained"))) {
        me = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
        if (me != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    }
    s->peer_uncached = me == hipSuccess;
    if (!p) HIPCHK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
    s->peer_allocs.push_back(p);
    HIPCHK(hipMemsetAsync(p, 0, bytes, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));         // zeroed (counter = 0) before anybody learns the handle
    hipIpcMemHandle_t hnd;
    HIPCHK(hipIpcGetMemHandle(&hnd, p));
    memcpy(ipc_handle64, &hnd, 64);
    *dev_ptr = p;
    return 0;
}
int plmpm_peer_open(plmpm_handle s, const void* ipc_handle64, void** dev_ptr) {
    REQUIRE(s && dev_ptr && ipc_handle64, "peer_open: bad argument");
    hipIpcMemHandle_t hnd;
    memcpy(&hnd, ipc_handle64, 64);
    void* p = nullptr;
    HIPCHK(hipIpcOpenMemHandle(&p, hnd, hipIpcMemLazyEnablePeerAccess));
    s->peer_mapped.push_back(p);
    *dev_ptr = p;
    return 0;
}

// local[i]: this rank's receive area for face i (plmpm_peer_alloc); remote[i]: the area the neighbour on face i allocated
// for ITS face towards this rank, mapped here (plmpm_peer_open) -- or any device pointer of that layout (loop-back runs)
int plmpm_halo_peer_setup(plmpm_handle s, int field, int n_faces, const int* bz_a, const int* bz_b, void* const* local, void* const* remote) {
    NEED_BOUND(s);
    REQUIRE(field >= 0 && field < 3 && n_faces >= 0 && n_faces <= 2, "halo_peer_setup: a z-slab has at most 2 faces");
    if (peer_init(s)) return -1;
    plmpm_sim::PeerField& F = s->peer[field];
    F = plmpm_sim::PeerField();
    for (int i = 0; i < n_faces; ++i) {
        const int ra = bz_a[i] - s->go[2] / 4, rb = bz_b[i] - s->go[2] / 4;
        REQUIRE(local[i] && remote[i] && ra >= 0 && rb <= s->nbw[2] && ra < rb, "halo_peer_setup: block planes [%d,%d) outside the grid window", bz_a[i], bz_b[i]);
        F.ba[i] = ra; F.bb[i] = rb;
        F.count[i] = (size_t)(rb - ra) * s->nbw[0] * s->nbw[1] * 64;
        F.local[i] = (char*)local[i]; F.remote[i] = (char*)remote[i];
    }
    F.n = n_faces;
    return 0;
}

// The arguments of exchange number ++seq of `field` (frame: which frame's grid_m / grid_v_in for PLMPM_HALO_GRID_IN) and the
// halo input of the grid kernels pointed at the half that exchange fills.  X.n = 0: this rank has no neighbour.
static int peer_prepare(plmpm_sim* s, int field, int frame, PeerXchg& X, size_t* blocks_out = nullptr, size_t* most_out = nullptr) {
    plmpm_sim::PeerField& F = s->peer[field];
    HaloIn& H = s->halo_in[field];
    memset(&X, 0, sizeof X);
    memset(&H, 0, sizeof H);
    if (F.n == 0) return 0;
    REQUIRE(*s->peer_status == 0, "halo exchange: an earlier arrival timed out (status 0x%x: field %d, face %d) -- a neighbouring rank has stopped",
            *s->peer_status, *s->peer_status >> 16, (*s->peer_status >> 8) & 255);
    char* base; int nc;
    if (plmpm_halo_field(s, field, frame, &base, &nc)) return -1;
    const unsigned seq = ++F.seq;
    const size_t pblk = (size_t)s->nbw[0] * s->nbw[1];              // blocks per plane
    X.n = F.n; X.ncomp = nc; X.seq = seq; X.done = s->peer_done; X.tag = ++s->peer_tag ? s->peer_tag : ++s->peer_tag; X.status = s->peer_status; X.status_dev = (int*)(s->peer_done + 16); X.code = (field << 16) | 1;
    X.timeout_ticks = (long long)(peer_timeout_seconds() * 1e8);
    X.spoil = s->peer_spoil;
    // the substep fields are sparse in the blocks the frame's scatter flagged; the loss mass grid is sent whole
    X.flags = field == PLMPM_HALO_LOSS_MASS ? nullptr : s->fstore + (size_t)frame * s->nflag;
    X.fgl = s->gwg_log2; X.fs = s->fs;
    size_t blocks = 0, most = 0;
    for (int c = 0; c < nc; ++c) X.src[c] = base + (size_t)c * s->G * s->tsz;
    for (int i = 0; i < F.n; ++i) {
        const size_t nblk = (size_t)(F.bb[i] - F.ba[i]) * pblk, vbytes = align_up(nblk * 4, 256);
        const size_t half = vbytes + (size_t)nc * F.count[i] * s->tsz;
        X.dst[i] = F.remote[i] + kPeerHeader + (seq & 1) * half;
        X.data_ofs[i] = vbytes;
        X.blk0[i] = (int)(F.ba[i] * pblk); X.nblk[i] = (int)nblk;
        X.arrive_remote[i] = (unsigned*)F.remote[i];
        X.arrive_local[i] = (unsigned*)F.local[i];
        blocks += nblk;
        most = std::max(most, nblk);
        H.ba[i] = F.ba[i]; H.bb[i] = F.bb[i];
        H.valid[i] = (const int*)(F.local[i] + kPeerHeader + (seq & 1) * half);
        H.buf[i] = F.local[i] + kPeerHeader + (seq & 1) * half + vbytes;
    }
    H.n = F.n;
    if (blocks_out) *blocks_out = blocks;
    if (most_out) *most_out = most;
    return 0;
}

// One exchange of `field` as a launch of its own: push, publish, wait -- on the engine's stream.
int plmpm_halo_peer_exchange(plmpm_handle s, int field, int frame) {
    NEED_BOUND(s);
    REQUIRE(field >= 0 && field < 3, "halo_peer_exchange: unknown field %d", field);
    PeerXchg X;
    size_t blocks = 0, most = 0;
    if (peer_prepare(s, field, frame, X, &blocks, &most)) return -1;
    if (X.n == 0) return 0;
    // 64 workgroups (the kernel is three memory latencies long whatever its size), more when a face has more than 64 blocks per wave
    const unsigned nwg = (unsigned)std::max<size_t>(std::min<size_t>(64, std::max<size_t>(1, (blocks + 3) / 4)), (most + 255) / 256);
    if (s->cfg.dtype == PLMPM_F64) LAUNCHB(s, K_HALO_XCHG, (k_halo_xchg<double>), dim3(nwg), 256, X);
    else LAUNCHB(s, K_HALO_XCHG, (k_halo_xchg<float>), dim3(nwg), 256, X);
    HIPCHK(hipGetLastError());
    return 0;
}
// Opt-in (PLMPM_PEER_FUSED=1): the exchanges of the two substep fields FOLDED INTO the grid kernels that consume them
// (k_grid_op_x / k_grid_op_grad_x: send the owned blocks of the exchanged planes | the interior blocks | wait | the blocks of
// the exchanged planes): one launch per exchange + grid phase instead of two, and the interior blocks' grid work hides the
// arrival -- the overlap the point-to-point transport gets from SlabEngine(overlap=True), here inside one launch.  Every
// workgroup of such a launch waits for the neighbours, so all of them must be resident at once -- on the neighbours' GPUs too:
// 512 grid workgroups fit a GPU exactly (k_grid_op_grad: 2 per CU x 256 CUs), so distributed.make_slab_env builds fused engines
// with plmpm_config.grid_workgroups = 256 (one rank per GPU) or 256 / ranks-per-GPU (ranks sharing a GPU: the tests).
// Off by default: measured on a middle rank of 8 at config 3 (one block plane thick: no interior to hide anything behind) the
// kernels sum to 86.7 us per fwd+bwd substep against 89.1 but the wall clock is 91.0 against 89.3 (profiles/r05_slab_host_cost.txt).
// The request alone is not enough: an engine whose grid launches were not sized for residency at create time
// (plmpm_config.grid_workgroups in 1 .. kFusedMaxWG; 0 = the library's 512, which fill a GPU with no margin) keeps the exchange
// kernels -- every fused launch of such an engine would run into the bounded wait.  Ranks may end up on different
// forms: both write the same receive areas and arrival counters, so they interoperate.
constexpr int kFusedMaxWG = 256;
static bool peer_fused(const plmpm_sim* s) {
    const char* e = getenv("PLMPM_PEER_FUSED");
    const bool asked = e ? e[0] != '0' : (PLB_PEER_FUSED_DEFAULT != 0);
    return asked && s->cfg.grid_workgroups > 0 && s->cfg.grid_workgroups <= kFusedMaxWG && s->gwg <= s->cfg.grid_workgroups;
}
int plmpm_peer_fused(plmpm_handle s, int* fused) {
    REQUIRE(s && fused, "null argument");
    *fused = peer_fused(s) ? 1 : 0;
    return 0;
}
// Collective re-synchronisation in two phases, with a host barrier over the ranks behind EACH (SlabEngine.reset_exchange):
//   phase 0 (drain): wait until every exchange kernel this rank has enqueued is finished -- after the barrier no rank has a
//                    kernel left that could still publish an (old, large) sequence number into a neighbour's arrival counter;
//   phase 1 (clear): this rank's arrival counters, its sequence numbers, the workgroup counter and the status word go back to
//                    zero -- after the second barrier every rank starts again at sequence number 1.
// (One phase -- clear, then a barrier -- is not enough: a neighbour whose exchange kernels were still draining could store its
// old sequence number into the counter just cleared, and the first exchange after the reset would pass its arrival test
// `(int)(got - seq) >= 0` on that stale value and read a half the neighbour had not written yet.)
// After a timeout -- or an exception between two exchanges on one rank -- the sequence numbers of the two sides of a face no
// longer agree; without this the engine could never exchange again.
int plmpm_halo_peer_reset(plmpm_handle s, int phase) {
    NEED_BOUND(s);
    REQUIRE(phase == 0 || phase == 1, "halo_peer_reset: phase must be 0 (drain) or 1 (clear)");
    if (!s->peer_done) return 0;
    HIPCHK(hipStreamSynchronize(s->stream));
    if (phase == 0) return 0;
    for (int f = 0; f < 3; ++f) {
        plmpm_sim::PeerField& F = s->peer[f];
        F.seq = 0;
        for (int i = 0; i < F.n; ++i) HIPCHK(hipMemsetAsync(F.local[i], 0, kPeerHeader, s->stream));
    }
    HIPCHK(hipMemsetAsync(s->peer_done, 0, 256, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    *s->peer_status = 0;
    return 0;
}
// Collective over the ranks of a slab run (every rank calls it with the same non-zero token, then meets at a host barrier):
// token into both neighbours' headers of `field`, wait for theirs.  ok[i] = 1 when face i's token arrived within timeout_s,
// wait_us[i] = how long the poll took.  A pre-flight for the first run on real neighbours (bench.py: transport_check.preflight);
// the exchange counters are not touched.
int plmpm_peer_ping(plmpm_handle s, int field, unsigned token, double timeout_s, int* ok2, double* wait_us2) {
    NEED_BOUND(s);
    REQUIRE(field >= 0 && field < 3 && token != 0 && ok2 && wait_us2 && timeout_s > 0, "peer_ping: bad argument");
    const plmpm_sim::PeerField& F = s->peer[field];
    ok2[0] = ok2[1] = 0; wait_us2[0] = wait_us2[1] = 0.0;
    if (F.n == 0) return 0;
    if (peer_init(s)) return -1;
    long long* res = (long long*)(s->peer_done + 32);              // bytes 128 .. 159 of the 256-byte control block
    HIPCHK(hipMemsetAsync(res, 0, 32, s->stream));
    hipLaunchKernelGGL(k_peer_ping, dim3(1), dim3(64), 0, s->stream, F.n, (unsigned*)F.remote[0], (unsigned*)F.remote[1],
                       (unsigned*)F.local[0], (unsigned*)F.local[1], token, (long long)(timeout_s * 1e8), res);
    HIPCHK(hipGetLastError());
    long long h[4];
    HIPCHK(hipMemcpyAsync(h, res, 32, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (int i = 0; i < F.n; ++i) { ok2[i] = h[2 * i + 1] >= 0 ? 1 : 0; wait_us2[i] = h[2 * i + 1] >= 0 ? h[2 * i + 1] * 0.01 : -1.0; }
    return 0;
}
// test hook: scale what this rank sends through face 0 (1 = off).  A run with a spoiled halo must be NOTICED by whoever
// compares transports (bench.py's self-check at N > 1, tests/test_gpu_z_bench_contract.py).
int plmpm_debug_peer_spoil(plmpm_handle s, double factor) {
    REQUIRE(s, "null handle");
    s->peer_spoil = (float)factor;
    return 0;
}
// 1: the receive areas are uncached device memory, 0: fine-grained (or none allocated yet)
int plmpm_peer_memory_kind(plmpm_handle s, int* uncached) {
    REQUIRE(s && uncached, "null argument");
    *uncached = s->peer_uncached ? 1 : 0;
    return 0;
}
int plmpm_peer_status(plmpm_handle s, int* status) {
    REQUIRE(s && status, "null argument");
    *status = s->peer_status ? *s->peer_status : 0;
    return 0;
}

// ---- the slab substep loops, host side: enqueue only ---------------------------------------------------------------
// plmpm_fk(first, n), then per substep  p2g | exchange of grid_m, grid_v_in | grid_op + g2p  (g2p deferred into the next
// substep's particle kernel, as on one GPU).  mpm_simulator.py:245-257, 365-376 for one slab.
int plmpm_slab_step(plmpm_handle s, int first, int n) {
    NEED_BOUND(s);
    REQUIRE(n >= 1 && first >= 0 && first + n < s->F + 1, "slab_step: frames [%d, %d] out of range", first, first + n);
    if (plmpm_fk(s, first, n)) return -1;
    const bool fused = peer_fused(s) && s->peer[PLMPM_HALO_GRID_IN].n > 0;
    int pending = 0;
    for (int f = first; f < first + n; ++f) {
        if (plmpm_p2g(s, f, pending)) return -1;
        pending = f + 1 < first + n;
        if (fused) {
            PeerXchg X;
            if (peer_prepare(s, PLMPM_HALO_GRID_IN, f, X)) return -1;
            // (a refused call has published nothing: take the sequence number back, or this rank would stay one exchange ahead of
            // its neighbours until the next collective reset)
            if (plmpm_grid_g2p_xchg(s, f, pending, &X)) { if (X.n) --s->peer[PLMPM_HALO_GRID_IN].seq; return -1; }
        } else {
            if (plmpm_halo_peer_exchange(s, PLMPM_HALO_GRID_IN, f)) return -1;
            if (plmpm_grid_g2p(s, f, pending)) return -1;
        }
    }
    return 0;
}
// reverse: per substep  g2p.grad | exchange of grid_v_out.grad | grid_op.grad + p2g.grad  (mpm_simulator.py:260-278); the
// caller then sums the pose adjoints over the ranks (plmpm_pose_grad_region) and runs plmpm_chain_grad
int plmpm_slab_step_grad(plmpm_handle s, int first, int n) {
    NEED_BOUND(s);
    REQUIRE(n >= 1 && first >= 0 && first + n < s->F + 1, "slab_step_grad: frames [%d, %d] out of range", first, first + n);
    const bool fused = peer_fused(s) && s->peer[PLMPM_HALO_GRID_OUT_ADJ].n > 0;
    for (int f = first + n - 1; f >= first; --f) {
        if (plmpm_grad_scatter(s, f)) return -1;
        if (fused) {
            PeerXchg X;
            if (peer_prepare(s, PLMPM_HALO_GRID_OUT_ADJ, f, X)) return -1;
            if (plmpm_grad_gather_xchg(s, f, &X)) { if (X.n) --s->peer[PLMPM_HALO_GRID_OUT_ADJ].seq; return -1; }
        } else {
            if (plmpm_halo_peer_exchange(s, PLMPM_HALO_GRID_OUT_ADJ, f)) return -1;
            if (plmpm_grad_gather(s, f)) return -1;
        }
    }
    return 0;
}

}  // extern "C"
