// TEST INFRASTRUCTURE ONLY -- a tiny CPU emulation of the slice of the HIP
// runtime + device language that lfpsqp.jl_amd/csrc uses, so the *same* kernel
// and host sources can be compiled with g++ and exercised (logic, indexing,
// reduction order, host orchestration, sharding) in the GPU-less build
// container and under CPU sanitizers.  It is reached only by putting
// tests/emu/include in front of the include path (tests/emu/Makefile); the
// product build (hipcc, __graft_entry__.build) never sees it, and the product
// Python package refuses to run without the real HIP library.
//
// Model: one fiber per GPU thread (hand-written x86-64 context switch: ucontext's swapcontext makes a
// sigprocmask syscall per switch, ~50x slower), a block's fibers are scheduled
// round-robin on one OS thread, __syncthreads()/wave shuffles are cooperative
// yields; blocks are distributed over a small pool of OS threads.  Wave = 64.
#pragma once

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#define LFPSQP_HIP_EMULATED 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
typedef struct emuStream_st* hipStream_t;
struct emuEvent_st { std::chrono::steady_clock::time_point t; };
typedef emuEvent_st* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0, hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 { double x, y; };
struct int2 { int x, y; };
static inline int2 make_int2(int x, int y) { int2 v; v.x = x; v.y = y; return v; }
static inline double2 make_double2(double a, double b) { return double2{a, b}; }

struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated HIP error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* nb, K, int, size_t) { *nb = 3; return hipSuccess; }
struct hipUUID { char bytes[16]; };
typedef int hipDevice_t;
static inline hipError_t hipDeviceGet(hipDevice_t* d, int i) { *d = i; return hipSuccess; }
static inline hipError_t hipDeviceGetUuid(hipUUID* u, hipDevice_t) { std::memcpy(u->bytes, "00000000000e0000", 16); return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    std::memset(p, 0, sizeof(*p)); std::strcpy(p->name, "cpu-emulator"); std::strcpy(p->gcnArchName, "emu");
    p->multiProcessorCount = 4; p->totalGlobalMem = size_t(8) << 30; return hipSuccess;
}
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::aligned_alloc(256, (n + 255) / 256 * 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
// ---- inter-process "device" memory for the P2P transport test: hipExtMallocWithFlags(hipDeviceMallocUncached) is backed by POSIX shared
// memory, its IPC handle is the object's name (two emulator processes then really share the mailbox)
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <map>
#include <string>
#define hipDeviceMallocUncached 0x3
#define hipIpcMemLazyEnablePeerAccess 0x1
struct hipIpcMemHandle_t { char reserved[64]; };
struct HipemuShm { std::string name; size_t bytes; bool owner; };
static inline std::map<void*, HipemuShm>& hipemu_shm() { static std::map<void*, HipemuShm> m; return m; }
static inline hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) {
    static int counter = 0;
    char name[64];
    std::snprintf(name, sizeof name, "/lfpsqp_emu_%d_%d", (int)getpid(), counter++);
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)n) != 0) return hipErrorInvalidValue;
    void* q = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (q == MAP_FAILED) return hipErrorInvalidValue;
    hipemu_shm()[q] = HipemuShm{name, n, true};
    *p = q;
    return hipSuccess;
}
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) {
    auto it = hipemu_shm().find(p);
    if (it == hipemu_shm().end()) return hipErrorInvalidValue;
    std::memset(h, 0, sizeof *h);
    std::snprintf(h->reserved, 48, "%s", it->second.name.c_str());
    std::memcpy(h->reserved + 48, &it->second.bytes, sizeof(size_t));
    return hipSuccess;
}
static inline hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) {
    size_t n = 0;
    std::memcpy(&n, h.reserved + 48, sizeof(size_t));
    const int fd = shm_open(h.reserved, O_RDWR, 0600);
    if (fd < 0) return hipErrorInvalidValue;
    void* q = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (q == MAP_FAILED) return hipErrorInvalidValue;
    hipemu_shm()[q] = HipemuShm{h.reserved, n, false};
    *p = q;
    return hipSuccess;
}
static inline hipError_t hipIpcCloseMemHandle(void* p) {
    auto it = hipemu_shm().find(p);
    if (it == hipemu_shm().end()) return hipErrorInvalidValue;
    munmap(p, it->second.bytes);
    hipemu_shm().erase(it);
    return hipSuccess;
}
static inline hipError_t hipFree(void* p) {
    auto it = hipemu_shm().find(p);
    if (it != hipemu_shm().end()) {
        munmap(p, it->second.bytes);
        if (it->second.owner) shm_unlink(it->second.name.c_str());
        hipemu_shm().erase(it);
        return hipSuccess;
    }
    std::free(p);
    return hipSuccess;
}
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
    for (size_t r = 0; r < h; ++r) std::memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
    return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)8 << 30; *t = (size_t)16 << 30; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent_st(); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess;
}

// ------------------------------------------------------------------ device side
namespace hipemu {
struct Idx { unsigned x, y, z; };
struct BlockState {
    unsigned nthreads = 0, live = 0, arrived = 0, gen = 0;
    unsigned w_arrived[16] = {0}, w_gen[16] = {0}, w_live[16] = {0};
    double scratch[1024];
    double scratch2[1024];
    void* main_sp = nullptr;
    std::vector<void*> fibers;   // saved stack pointers
    std::vector<char*> stacks;
    std::vector<char> done;
    unsigned cur = 0;
    const std::function<void()>* body = nullptr;
};
inline thread_local BlockState* g_blk = nullptr;
inline thread_local Idx g_tid{0, 0, 0}, g_bid{0, 0, 0}, g_bdim{1, 1, 1}, g_gdim{1, 1, 1};
constexpr size_t kStack = 256 * 1024;

// save callee-saved registers + stack pointer of the current context, load another one
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
__asm__(".text\n.weak hipemu_switch\n.type hipemu_switch,@function\nhipemu_switch:\n"
        "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
        "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
        "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
        ".size hipemu_switch,.-hipemu_switch\n");

inline void yield_() { BlockState* b = g_blk; hipemu_switch(&b->fibers[b->cur], b->main_sp); }

inline void syncthreads() {
    BlockState* b = g_blk;
    unsigned my = b->gen;
    if (++b->arrived >= b->live) { b->arrived = 0; b->gen++; return; }
    while (b->gen == my) yield_();
}
inline void wave_sync() {
    BlockState* b = g_blk;
    unsigned w = g_tid.x >> 6;
    unsigned my = b->w_gen[w];
    if (++b->w_arrived[w] >= b->w_live[w]) { b->w_arrived[w] = 0; b->w_gen[w]++; return; }
    while (b->w_gen[w] == my) yield_();
}
inline void trampoline() {
    BlockState* b = g_blk;
    (*b->body)();
    unsigned t = b->cur;
    b->done[t] = 1;
    b->live--;
    b->w_live[t >> 6]--;
    // a thread that exits releases barriers the remaining threads are waiting on
    if (b->live > 0 && b->arrived >= b->live) { b->arrived = 0; b->gen++; }
    unsigned w = t >> 6;
    if (b->w_live[w] > 0 && b->w_arrived[w] >= b->w_live[w]) { b->w_arrived[w] = 0; b->w_gen[w]++; }
    hipemu_switch(&b->fibers[t], b->main_sp);
    __builtin_trap();   // a finished fiber is never resumed
}
inline void run_block(BlockState& b, unsigned nthreads, const std::function<void()>& body) {
    if (b.fibers.size() < nthreads) {
        size_t old = b.fibers.size();
        b.fibers.resize(nthreads);
        b.stacks.resize(nthreads, nullptr);
        for (size_t i = old; i < nthreads; ++i) b.stacks[i] = (char*)std::malloc(kStack);
    }
    b.done.assign(nthreads, 0);
    b.nthreads = b.live = nthreads;
    b.arrived = 0;
    for (unsigned w = 0; w < 16; ++w) {
        b.w_arrived[w] = 0;
        unsigned lo = w * 64;
        b.w_live[w] = nthreads > lo ? (nthreads - lo > 64 ? 64 : nthreads - lo) : 0;
    }
    b.body = &body;
    g_blk = &b;
    for (unsigned t = 0; t < nthreads; ++t) {
        // initial frame: six zeroed callee-saved registers, then `ret` into trampoline with the
        // stack aligned as at a normal function entry (rsp == 8 mod 16)
        uintptr_t top = ((uintptr_t)b.stacks[t] + kStack) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;                      // fake return address of trampoline
        *--sp = (void*)&trampoline;           // ret target
        for (int k = 0; k < 6; ++k) *--sp = nullptr;
        b.fibers[t] = (void*)sp;
    }
    while (b.live > 0) {
        for (unsigned t = 0; t < nthreads; ++t) {
            if (b.done[t]) continue;
            b.cur = t;
            g_tid = Idx{t % g_bdim.x, (t / g_bdim.x) % g_bdim.y, t / (g_bdim.x * g_bdim.y)};
            hipemu_switch(&b.main_sp, b.fibers[t]);
        }
    }
}
inline std::mutex& pool_mutex() { static std::mutex m; return m; }
inline std::vector<BlockState*>& pool() { static std::vector<BlockState*> p; return p; }
inline BlockState* acquire_block_state() {
    std::lock_guard<std::mutex> g(pool_mutex());
    if (!pool().empty()) { BlockState* b = pool().back(); pool().pop_back(); return b; }
    return new BlockState();
}
inline void release_block_state(BlockState* b) {
    std::lock_guard<std::mutex> g(pool_mutex());
    pool().push_back(b);
}
inline int num_workers() {
    const char* e = std::getenv("HIPEMU_THREADS");
    int n = e ? std::atoi(e) : 4;
    return n < 1 ? 1 : n;
}
// Helper threads of a launch, kept for the life of the process (a launch used to start and join fresh OS threads: tens of microseconds each,
// several per launch, millions of launches in a test run).  One launch at a time (the library launches from one host thread); the pool is
// leaked on purpose -- its threads sleep on a condition variable until the process exits.
struct HelperPool {
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> th;
    const std::function<void()>* job = nullptr;
    unsigned long epoch = 0;
    int want = 0, pending = 0;
    void helper(int idx) {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void()>* j = nullptr;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_job.wait(lk, [&] { return epoch != seen; });
                seen = epoch;
                if (idx < want) j = job;
            }
            if (j) {
                (*j)();
                std::lock_guard<std::mutex> lk(m);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    void run(int n, const std::function<void()>& work) {
        {
            std::lock_guard<std::mutex> lk(m);
            while ((int)th.size() < n) {
                const int idx = (int)th.size();
                th.emplace_back([this, idx] { helper(idx); });
                th.back().detach();
            }
            job = &work;
            want = pending = n;
            ++epoch;
        }
        cv_job.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
        job = nullptr;
    }
};
inline HelperPool& helpers() { static HelperPool* p = new HelperPool(); return *p; }

template <class K, class... Args>
void launch(K kernel, dim3 grid, dim3 block, Args... args) {
    unsigned nblocks = grid.x * grid.y * grid.z;
    unsigned nthreads = block.x * block.y * block.z;
    if (nblocks == 0 || nthreads == 0) return;
    std::function<void()> body = [=]() { kernel(args...); };
    std::atomic<unsigned> next{0};
    auto worker = [&]() {
        // fiber stacks are expensive to set up (256 KB each, up to 1024 per block) and every launch starts fresh OS threads:
        // block states are pooled across launches instead of living (and leaking) in thread-local storage
        struct Lease {
            BlockState* b;
            Lease() : b(acquire_block_state()) {}
            ~Lease() { release_block_state(b); }
        } lease;
        BlockState& bs = *lease.b;
        g_bdim = Idx{block.x, block.y, block.z};
        g_gdim = Idx{grid.x, grid.y, grid.z};
        for (;;) {
            unsigned bidx = next.fetch_add(1);
            if (bidx >= nblocks) break;
            g_bid = Idx{bidx % grid.x, (bidx / grid.x) % grid.y, bidx / (grid.x * grid.y)};
            run_block(bs, nthreads, body);
        }
    };
    int nw = num_workers();
    if ((unsigned)nw > nblocks) nw = (int)nblocks;
    if (nw <= 1) { worker(); return; }
    helpers().run(nw - 1, worker);          // nw - 1 pooled helper threads + this one
}
template <class T> inline T shfl_generic(T v, unsigned src_lane_in_wave) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    BlockState* b = g_blk;
    unsigned t = g_tid.x;  // 1-D blocks only
    std::memcpy(&b->scratch[t], &v, sizeof(T));
    wave_sync();
    T r;
    std::memcpy(&r, &b->scratch[(t & ~63u) | (src_lane_in_wave & 63u)], sizeof(T));
    wave_sync();
    return r;
}
// v_mfma_f64_16x16x4_f64: lane l holds A[i=l&15][k=l>>4] and B[k=l>>4][j=l&15]; D[row=(l>>4)+4*reg][col=l&15]
typedef double f64x4_t __attribute__((vector_size(32)));
inline f64x4_t mfma_f64_16x16x4(double a, double b, f64x4_t c) {
    BlockState* blk = g_blk;
    const unsigned t = g_tid.x, lane = t & 63u, base = t & ~63u;
    blk->scratch[t] = a;
    blk->scratch2[t] = b;
    wave_sync();
    for (int r = 0; r < 4; ++r) {
        const unsigned row = (lane >> 4) + 4u * r, col = lane & 15u;
        double s = 0.0;
        for (unsigned k = 0; k < 4; ++k) s += blk->scratch[base + row + 16u * k] * blk->scratch2[base + col + 16u * k];
        c[r] += s;
    }
    wave_sync();
    return c;
}
}  // namespace hipemu

#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) hipemu::mfma_f64_16x16x4((a), (b), (c))
#define threadIdx (hipemu::g_tid)
#define blockIdx (hipemu::g_bid)
#define blockDim (hipemu::g_bdim)
#define gridDim (hipemu::g_gdim)
#define warpSize 64
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipemu::launch(kernel, dim3(grid), dim3(block), ##__VA_ARGS__)

static inline void __syncthreads() { hipemu::syncthreads(); }
template <class T> static inline T __shfl_xor(T v, int mask, int = 64) { return hipemu::shfl_generic(v, (threadIdx.x & 63) ^ (unsigned)mask); }
template <class T> static inline T __shfl_down(T v, unsigned d, int = 64) { unsigned l = threadIdx.x & 63; return hipemu::shfl_generic(v, l + d < 64 ? l + d : l); }
template <class T> static inline T __shfl(T v, int src, int = 64) { return hipemu::shfl_generic(v, (unsigned)src); }
// v_permlane32_swap / v_permlane16_swap (semantics probed on MI355X, tools/micro/permlane_probe.hip): lanes with the
// bit clear get {own a, partner's a}, lanes with it set get {partner's b, own b}
struct hipemu_u32x2 { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
static inline hipemu_u32x2 hipemu_permlane_swap(unsigned a, unsigned b, unsigned bit) {
    const unsigned pa = __shfl_xor(a, (int)bit), pb = __shfl_xor(b, (int)bit);
    return (threadIdx.x & bit) ? hipemu_u32x2{{pb, b}} : hipemu_u32x2{{a, pa}};
}
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipemu_permlane_swap((a), (b), 32u)
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) hipemu_permlane_swap((a), (b), 16u)
// buffer-descriptor loads (scalar base + 32-bit lane offset)
struct __amdgpu_buffer_rsrc_t { const char* base; };
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int, int) { return {static_cast<const char*>(p)}; }
static inline unsigned long long __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int) {
    unsigned long long v;
    memcpy(&v, rs.base + voff + soff, 8);
    return v;
}
static inline double __builtin_nontemporal_load(const double* p) { return *p; }
static inline void __builtin_nontemporal_store(double v, double* p) { *p = v; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long expect, unsigned long long v) {
    __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return expect;
}
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
    unsigned long long cur = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (cur < v && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return cur;
}
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
template <class T> static inline T __hip_atomic_load(const T* p, int, int) { T v; __atomic_load(const_cast<T*>(p), &v, __ATOMIC_SEQ_CST); return v; }   // (really atomic: two emulator processes may share the word)
template <class T, class V> static inline void __hip_atomic_store(T* p, V v, int, int) { T w = (T)v; __atomic_store(p, &w, __ATOMIC_SEQ_CST); }
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) { return v; }   // only used on wave-uniform values
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#include <sched.h>
#include <time.h>
static inline long long wall_clock64() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (long long)ts.tv_sec * 100000000LL + ts.tv_nsec / 10; }   // 100 MHz
#define __builtin_amdgcn_s_sleep(x) sched_yield()
