// TEST INFRASTRUCTURE ONLY: the scheduler of the CPU interpreter described in hip/hip_runtime.h -- fibers, workgroup barrier,
// wave-level gather, DPP lane maps, "device" memory -- plus host versions of the two rocPRIM entry points of plmpm_sort.hip.
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#include <numeric>
#include <vector>

// AddressSanitizer has to be told about every stack switch (it keeps per-stack bookkeeping for use-after-return / stack redzones):
// `make SAN=1` compiles the device source and this file with -fsanitize=address,undefined -- out-of-bounds LDS tiles, particle
// rows and grid nodes, misaligned or overflowing index arithmetic in the kernels are then REPORTED instead of silently corrupting
// a neighbour (the GPU has no such tool on this pool).
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#define HIPEMU_ASAN 1
#else
#define HIPEMU_ASAN 0
#endif
// ThreadSanitizer (`make SAN=thread`): every fiber is a thread of its own to TSan (its fiber API), the barrier and the wave-level
// operations are the happens-before edges between them -- so two threads of a workgroup that touch the same LDS or global word
// without a barrier or wave operation in between, one of them writing, are REPORTED: the missing-__syncthreads detector the GPU pool
// has no tool for.  (Workgroups on different OS threads, PLMPM_EMUL_THREADS, add the conflicts between workgroups.)
// (this file itself is compiled WITHOUT -fsanitize=thread in such a build -- the scheduler's own bookkeeping is not what is being
// checked -- and only calls TSan's interface: -DHIPEMU_TSAN=1, tests/host_emul/Makefile)
#ifndef HIPEMU_TSAN
#define HIPEMU_TSAN 0
#endif
#if HIPEMU_TSAN
#include <sanitizer/tsan_interface.h>
static char g_tsan_done;                       // happens-before: every thread of a launch -> the host behind the launch
static thread_local char g_tsan_chain;         // ... and workgroup -> next workgroup on the same OS thread (they reuse its LDS storage)
#endif

extern "C" void hipemu_switch(void** save_sp, void* next_sp);
// callee-saved registers on the old stack, swap stack pointers, restore from the new one (System V x86-64)
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

thread_local Fiber* cur = nullptr;
thread_local Idx g_block = {0, 0, 0};
Idx g_bdim = {1, 1, 1}, g_gdim = {1, 1, 1};

static constexpr int kMaxThreads = 1024;
#if HIPEMU_TSAN || defined(__SANITIZE_ADDRESS__)
static constexpr size_t kStack = 1024u << 10;          // instrumented frames are several times larger
#else
static constexpr size_t kStack = 256u << 10;
#endif
// scheduler state of the workgroup an OS thread is running
static thread_local char* g_stacks = nullptr;
static thread_local Fiber g_fib[kMaxThreads];
static thread_local void* g_main_sp = nullptr;
static thread_local unsigned long g_progress = 0;
static void (*g_thunk)(void*) = nullptr;
static void* g_ctx = nullptr;

struct Wave {
    int live, arrived;
    unsigned gen;
    uint64_t live_mask, snap[2];
    uint64_t buf[2][64];
    unsigned op[64];
};
static thread_local Wave g_wave[kMaxThreads / 64];
static thread_local int g_blk_live, g_blk_arrived;
static thread_local unsigned g_blk_gen;

// ---- memory another process can map (hipExtMallocWithFlags + hipIpc*): POSIX shared memory, the handle = its name and size
struct Shared { std::string name; size_t bytes; bool mine; };
static std::map<void*, Shared> g_shared;
static void shared_cleanup() {
    for (auto& kv : g_shared) if (kv.second.mine) shm_unlink(kv.second.name.c_str());
}
// segments of processes that were killed (a test's timeout) never reach their atexit: every process that allocates first removes
// what dead processes left behind -- /dev/shm is memory
static void shared_gc() {
    DIR* d = opendir("/dev/shm");
    if (!d) return;
    while (dirent* e = readdir(d)) {
        int pid = 0, n = 0;
        if (sscanf(e->d_name, "hipemu.%d.%d", &pid, &n) == 2 && pid > 0 && kill(pid, 0) != 0 && errno == ESRCH)
            shm_unlink((std::string("/") + e->d_name).c_str());
    }
    closedir(d);
}
void* shared_alloc(size_t bytes) {
    static int counter = 0;
    static bool hooked = false;
    if (!hooked) { hooked = true; atexit(shared_cleanup); shared_gc(); }
    char name[48];
    snprintf(name, sizeof name, "/hipemu.%d.%d", (int)getpid(), counter++);
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return nullptr;
    void* p = ftruncate(fd, (off_t)bytes) == 0 ? mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(name); return nullptr; }
    memset(p, 0xff, bytes);
    g_shared[p] = Shared{name, bytes, true};
    return p;
}
bool shared_handle(void* p, char* handle64) {
    auto it = g_shared.find(p);
    if (it == g_shared.end() || !it->second.mine) return false;
    memset(handle64, 0, 64);
    const unsigned long long n = it->second.bytes;
    memcpy(handle64, &n, 8);
    snprintf(handle64 + 8, 56, "%s", it->second.name.c_str());
    return true;
}
void* shared_open(const char* handle64) {
    unsigned long long n = 0;
    memcpy(&n, handle64, 8);
    const std::string name(handle64 + 8, strnlen(handle64 + 8, 55));
    for (auto& kv : g_shared) if (kv.second.mine && kv.second.name == name) return nullptr;      // (as HIP: a process cannot open its own handle)
    const int fd = shm_open(name.c_str(), O_RDWR, 0600);
    if (fd < 0) return nullptr;
    void* p = mmap(nullptr, (size_t)n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return nullptr;
    g_shared[p] = Shared{name, (size_t)n, false};
    return p;
}
void shared_close(void* p) {
    auto it = g_shared.find(p);
    if (it == g_shared.end() || it->second.mine) return;
    munmap(p, it->second.bytes);
    g_shared.erase(it);
}

void* dyn_lds_arena() {
    static thread_local double arena[8192];           // 64 KiB
    return arena;
}
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void* device_alloc(size_t bytes) {
    void* p = nullptr;
    if (posix_memalign(&p, 256, bytes ? bytes : 256)) return nullptr;
    memset(p, 0xff, bytes);              // hipMalloc returns garbage: nothing may rely on zeros (NaN as a float, -1 as an int)
    return p;
}
void device_free(void* p) {
    auto it = g_shared.find(p);
    if (it != g_shared.end()) {
        if (it->second.mine) { munmap(p, it->second.bytes); shm_unlink(it->second.name.c_str()); g_shared.erase(it); }
        return;
    }
    free(p);
}
// Zeroing hundreds of megabytes (an engine clears its per-frame grid stores when its workspaces are bound) by touching every page is
// most of a small test's time: whole pages of a large zero-fill are handed back to the kernel instead (MADV_DONTNEED on private
// anonymous memory: they read as zero again, and only the pages a test really uses are ever faulted in).
void device_memset(void* p, int v, size_t n) {
    if (!n) return;
    char* b = (char*)p;
    if (v == 0 && n >= (4u << 20)) {
        char* lo = (char*)(((uintptr_t)b + 4095) & ~(uintptr_t)4095);
        char* hi = (char*)(((uintptr_t)b + n) & ~(uintptr_t)4095);
        if (hi > lo && madvise(lo, (size_t)(hi - lo), MADV_DONTNEED) == 0) {
            memset(b, 0, (size_t)(lo - b));
            memset(hi, 0, (size_t)(b + n - hi));
            return;
        }
    }
    memset(p, v, n);
}

[[noreturn]] static void die(const char* what) {
    fprintf(stderr, "hipemu: %s (workgroup %u of %u, thread %u, wave %d lane %d)\n", what, g_block.x, g_gdim.x, cur ? cur->tid.x : 0, cur ? cur->wave : -1,
            cur ? cur->lane : -1);
    fflush(stderr);
    abort();
}
#if HIPEMU_ASAN
static thread_local void* g_main_fake = nullptr;
static thread_local const void* g_main_lo = nullptr;
static thread_local size_t g_main_size = 0;
#endif
// fiber -> scheduler
#if HIPEMU_TSAN
static thread_local void* g_main_tsan = nullptr;
#endif
static inline void yield() {
#if HIPEMU_ASAN
    __sanitizer_start_switch_fiber(&cur->asan_fake, g_main_lo, g_main_size);
#endif
#if HIPEMU_TSAN
    __tsan_switch_to_fiber(g_main_tsan, 1);              // (1 = no synchronisation implied by the switch itself)
#endif
    hipemu_switch(&cur->sp, g_main_sp);
#if HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(cur->asan_fake, nullptr, nullptr);
#endif
}
// scheduler -> fiber t
static inline void resume(Fiber& f, int t) {
    cur = &f;
#if HIPEMU_ASAN
    __sanitizer_start_switch_fiber(&g_main_fake, g_stacks + (size_t)t * kStack, kStack);
#endif
#if HIPEMU_TSAN
    __tsan_switch_to_fiber(f.tsan_fiber, 1);
#endif
    hipemu_switch(&g_main_sp, f.sp);
#if HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(g_main_fake, nullptr, nullptr);
#endif
}

// between two polls of a spin-wait on another agent (s_sleep): not a deadlock, and the other fibers / processes get the core
void external_wait() {
    ++g_progress;
    sched_yield();
    yield();
}

static void wave_release(Wave& w) {
    unsigned op = 0;
    bool first = true;
    for (int l = 0; l < 64; ++l) {
        if (!((w.live_mask >> l) & 1)) continue;
        if (first) { op = w.op[l]; first = false; }
        else if (w.op[l] != op) {
            fprintf(stderr, "hipemu: lanes of one wave reached different wave-level operations (lane %d: 0x%x, others 0x%x)\n", l, w.op[l], op);
            die("divergent collective");
        }
    }
    w.snap[w.gen & 1] = w.live_mask;
    w.arrived = 0;
    ++w.gen;
}
Gathered wave_gather(uint64_t mine, unsigned op) {
    Wave& w = g_wave[cur->wave];
    const unsigned g = w.gen;
    w.buf[g & 1][cur->lane] = mine;
    w.op[cur->lane] = op;
    ++w.arrived;
    ++g_progress;
#if HIPEMU_TSAN
    __tsan_release(&w);                                  // what this lane did before the operation ...
#endif
    if (w.arrived == w.live) wave_release(w);
    while (w.gen == g) yield();
#if HIPEMU_TSAN
    __tsan_acquire(&w);                                  // ... is visible to every lane behind it
#endif
    return Gathered{w.buf[g & 1], w.snap[g & 1]};
}
void block_barrier() {
    const unsigned g = g_blk_gen;
    ++g_blk_arrived;
    ++g_progress;
#if HIPEMU_TSAN
    __tsan_release(&g_blk_gen);
#endif
    if (g_blk_arrived == g_blk_live) { g_blk_arrived = 0; ++g_blk_gen; }
    while (g_blk_gen == g) yield();
#if HIPEMU_TSAN
    __tsan_acquire(&g_blk_gen);
#endif
}

int dpp_source_lane(int lane, int ctrl) {
    const int row = lane & ~15, r = lane & 15;
    if (ctrl >= 0 && ctrl <= 0xff) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);                       // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10f) { const int s = r + (ctrl & 15); return s <= 15 ? row | s : -1; }       // row_shl
    if (ctrl >= 0x111 && ctrl <= 0x11f) { const int s = r - (ctrl & 15); return s >= 0 ? row | s : -1; }        // row_shr
    if (ctrl >= 0x121 && ctrl <= 0x12f) return row | ((r - (ctrl & 15)) & 15);                                  // row_ror
    if (ctrl == 0x140) return row | (15 - r);                                                                   // row_mirror
    if (ctrl == 0x141) return (lane & ~7) | (7 - (lane & 7));                                                   // row_half_mirror
    if (ctrl == 0x142) return lane >= 16 ? row - 1 : -1;                                                        // row_bcast:15 (lane 15 of the row before)
    if (ctrl == 0x143) return lane >= 32 ? 31 : -1;                                                             // row_bcast:31
    fprintf(stderr, "hipemu: DPP control 0x%x\n", ctrl);
    die("unsupported DPP control");
}

static void fiber_exit() {
    Fiber* f = cur;
    f->done = true;
    ++g_progress;
    Wave& w = g_wave[f->wave];
    --w.live;
    w.live_mask &= ~(1ULL << f->lane);
    if (w.live > 0 && w.arrived == w.live) wave_release(w);
    --g_blk_live;
    if (g_blk_live > 0 && g_blk_arrived == g_blk_live) { g_blk_arrived = 0; ++g_blk_gen; }
#if HIPEMU_ASAN
    __sanitizer_start_switch_fiber(nullptr, g_main_lo, g_main_size);        // (nullptr: this stack is never used again)
#endif
#if HIPEMU_TSAN
    __tsan_release(&g_tsan_done);
    __tsan_release(&g_tsan_chain);
    __tsan_switch_to_fiber(g_main_tsan, 1);
#endif
    hipemu_switch(&f->sp, g_main_sp);
    die("a finished fiber was resumed");
}
static void fiber_entry() {
#if HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &g_main_lo, &g_main_size);
#endif
    // (ThreadSanitizer: a fiber inherits the happens-before state of the context that CREATES it -- the workgroup's OS thread, behind the
    // host's launch and behind the previous workgroup it ran -- and nothing from its sibling threads)
    g_thunk(g_ctx);
    fiber_exit();
}

// one workgroup, start to finish, on the calling OS thread
static void run_workgroup(unsigned lin, dim3 grid, dim3 block, int nt, long shuffle, unsigned long long& rng) {
    if (!g_stacks) {
        g_stacks = (char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (char*)MAP_FAILED) die("mmap of the fiber stacks");
    }
    const int nw = (nt + 63) / 64;
    g_block = {lin % grid.x, (lin / grid.x) % grid.y, lin / (grid.x * grid.y)};
    for (int w = 0; w < nw; ++w) { g_wave[w].live = 0; g_wave[w].arrived = 0; g_wave[w].gen = 0; g_wave[w].live_mask = 0; }
    g_blk_live = nt; g_blk_arrived = 0; g_blk_gen = 0;
    for (int t = 0; t < nt; ++t) {
        Fiber& f = g_fib[t];
        f.tid = {(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
        f.wave = t >> 6; f.lane = t & 63; f.done = false; f.asan_fake = nullptr;
#if HIPEMU_TSAN
        f.tsan_fiber = __tsan_create_fiber(0);
#endif
        Wave& w = g_wave[f.wave];
        ++w.live; w.live_mask |= 1ULL << f.lane;
        // initial frame: six callee-saved registers, the entry point as the return address, a null return address above it
        // (16-byte alignment of a freshly called function: rsp = 16 k + 8 at its first instruction)
        // (the tops are staggered: 256 stacks at a power-of-two stride would put every fiber's hot frames into the same cache sets)
        uintptr_t top = ((uintptr_t)g_stacks + (size_t)(t + 1) * kStack - (size_t)((t * 2368) & 0xffff)) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;
        *--sp = (void*)&fiber_entry;
        for (int k = 0; k < 6; ++k) *--sp = nullptr;
        f.sp = sp;
    }
#if HIPEMU_TSAN
    g_main_tsan = __tsan_get_current_fiber();
#endif
    int alive = nt;
    while (alive > 0) {
        const unsigned long before = g_progress;
        alive = 0;
        int t0 = 0;
        if (shuffle) { rng = rng * 6364136223846793005ULL + 1442695040888963407ULL; t0 = (int)((rng >> 33) % (unsigned)nt); }
        for (int k = 0; k < nt; ++k) {
            const int t = shuffle ? (t0 + ((rng >> 20) & 1 ? k : nt - 1 - k) + nt) % nt : k;
            Fiber& f = g_fib[t];
            if (f.done) continue;
            resume(f, t);
            if (!f.done) ++alive;
        }
        if (alive > 0 && g_progress == before) {
            for (int t = 0; t < nt; ++t) if (!g_fib[t].done) { cur = &g_fib[t]; break; }
            die("deadlock: no thread of the workgroup can make progress (a barrier or wave operation not reached by all live threads, or a "
                "spin-wait without s_sleep on another workgroup)");
        }
    }
    cur = nullptr;
#if HIPEMU_TSAN
    __tsan_acquire(&g_tsan_chain);                       // the next workgroup on this OS thread (same LDS storage) comes after this one
    for (int t = 0; t < nt; ++t) __tsan_destroy_fiber(g_fib[t].tsan_fiber);
#endif
}

// PLMPM_EMUL_THREADS=n: a pool of OS threads runs the workgroups of a launch, each thread one workgroup at a time (its own fibers,
// its own LDS).  Launches of at most kResident workgroups get one thread PER workgroup: all of them are resident, as the kernels that
// wait for other workgroups of their launch require.
static constexpr size_t kResident = 16;
struct Pool {
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    std::vector<std::thread> threads;
    unsigned long generation = 0;
    size_t want = 0, busy = 0;                 // threads that should take part in the current launch / that are still in it
    std::atomic<size_t> next{0};
    // the current launch
    dim3 grid, block;
    int nt = 0;
    long shuffle = 0;
    const unsigned* order = nullptr;
    size_t nwg = 0;
    bool stop = false;
};
static Pool* g_pool = nullptr;
static void pool_worker(Pool* P, size_t index) {
    unsigned long seen = 0;
    unsigned long long rng = 0x9e3779b97f4a7c15ULL * (index + 2);
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(P->m);
            P->cv_go.wait(lk, [&] { return P->stop || (P->generation != seen && index < P->want); });
            if (P->stop) return;
            seen = P->generation;
        }
        for (;;) {
            const size_t i = P->next.fetch_add(1);
            if (i >= P->nwg) break;
            run_workgroup(P->order[i], P->grid, P->block, P->nt, P->shuffle, rng);
        }
        std::lock_guard<std::mutex> lk(P->m);
        if (--P->busy == 0) P->cv_done.notify_all();
    }
}

void run_grid(dim3 grid, dim3 block, size_t lds, void (*thunk)(void*), void* ctx) {
    const int nt = (int)(block.x * block.y * block.z);
    if (nt <= 0 || nt > kMaxThreads) die("workgroup size");
    if (lds > 8192 * sizeof(double)) die("dynamic LDS larger than the arena");
    if (cur) die("nested launch");
    g_thunk = thunk; g_ctx = ctx;
    g_bdim = {block.x, block.y, block.z};
    g_gdim = {grid.x, grid.y, grid.z};
    // PLMPM_EMUL_SHUFFLE=<seed>: the workgroups of every launch in a pseudo-random order, and the threads of a workgroup resumed in a
    // pseudo-random rotation that changes with every scheduling round -- nothing in the kernels may depend on which workgroup's atomics
    // arrive first or on which lane of a wave runs ahead (beyond the round-off of the floating-point sums)
    static const long shuffle = getenv("PLMPM_EMUL_SHUFFLE") ? atol(getenv("PLMPM_EMUL_SHUFFLE")) + 1 : 0;
    static const long nthreads = getenv("PLMPM_EMUL_THREADS") ? atol(getenv("PLMPM_EMUL_THREADS")) : 1;
    static unsigned long long rng = 0x9e3779b97f4a7c15ULL;
    const size_t nwg = (size_t)grid.x * grid.y * grid.z;
    std::vector<unsigned> order(nwg);
    std::iota(order.begin(), order.end(), 0u);
    if (shuffle) {
        rng ^= (unsigned long long)shuffle * 0xbf58476d1ce4e5b9ULL;
        for (size_t i = nwg; i > 1; --i) {
            rng = rng * 6364136223846793005ULL + 1442695040888963407ULL;
            std::swap(order[i - 1], order[(size_t)((rng >> 33) % i)]);
        }
    }
    if (nthreads <= 1 || nwg <= 1) {
        for (size_t i = 0; i < nwg; ++i) run_workgroup(order[i], grid, block, nt, shuffle, rng);
#if HIPEMU_TSAN
        __tsan_acquire(&g_tsan_done);
#endif
        return;
    }
    if (!g_pool) g_pool = new Pool;             // (never destroyed: its threads sleep until the process ends)
    Pool* P = g_pool;
    const size_t want = nwg <= kResident ? nwg : std::min<size_t>((size_t)nthreads, nwg);
    std::unique_lock<std::mutex> lk(P->m);
    while (P->threads.size() < want) { const size_t idx = P->threads.size(); P->threads.emplace_back(pool_worker, P, idx); }
    P->grid = grid; P->block = block; P->nt = nt; P->shuffle = shuffle; P->order = order.data(); P->nwg = nwg;
    P->next.store(0);
    P->want = want; P->busy = want;
    ++P->generation;
    P->cv_go.notify_all();
    P->cv_done.wait(lk, [&] { return P->busy == 0; });
#if HIPEMU_TSAN
    __tsan_acquire(&g_tsan_done);
#endif
}
}  // namespace hipemu

// ---- plmpm_sort.hip on the host (the device build calls rocPRIM; same contracts: stable pair sort on the low key bits, exclusive scan)
extern "C" size_t plmpm_sort_temp_bytes(int) { return 256; }
extern "C" size_t plmpm_scan_temp_bytes(size_t) { return 256; }
extern "C" int plmpm_exclusive_scan(void*, size_t, const unsigned* in, unsigned* out, size_t n, void*) {
    unsigned acc = 0;
    for (size_t i = 0; i < n; ++i) { const unsigned v = in[i]; out[i] = acc; acc += v; }
    return 0;
}
extern "C" int plmpm_sort_pairs(void*, size_t, const unsigned* kin, unsigned* kout, const int* vin, int* vout, int n, int key_bits, void*) {
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    const unsigned mask = key_bits >= 32 ? 0xffffffffu : ((1u << key_bits) - 1u);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return (kin[a] & mask) < (kin[b] & mask); });
    std::vector<unsigned> k(n);
    std::vector<int> v(n);
    for (int i = 0; i < n; ++i) { k[i] = kin[idx[i]]; v[i] = vin[idx[i]]; }
    memcpy(kout, k.data(), sizeof(unsigned) * n);
    memcpy(vout, v.data(), sizeof(int) * n);
    return 0;
}
