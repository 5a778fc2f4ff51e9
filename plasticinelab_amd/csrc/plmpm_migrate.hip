// Particle migration between z-slabs and the per-epoch row bookkeeping of slab engines.
#include "plmpm_internal.h"

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
}  // extern "C"
