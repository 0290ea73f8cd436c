// Context, buffers, transfers, communicator (RCCL via dlopen or host callback).
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <ctype.h>
#include <string.h>

#include <algorithm>

#include "internal.h"

namespace lfpsqp {

int set_err(lfpsqp_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else fprintf(stderr, "lfpsqp: %s\n", buf);
    return code;
}

int ensure_part(lfpsqp_ctx* ctx, size_t doubles) {
    if (doubles <= ctx->part_cap) return 0;
    // growing the workspace is a rare, synchronising event (first call at a new size)
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->part) LF_HIP(ctx, hipFree(ctx->part));
    ctx->part = nullptr;
    ctx->part_cap = 0;
    size_t cap = doubles + doubles / 4 + 4096;
    LF_HIP(ctx, hipMalloc((void**)&ctx->part, cap * sizeof(double)));
    ctx->part_cap = cap;
    return 0;
}

int ensure_small(lfpsqp_ctx* ctx, size_t doubles) {
    if (doubles <= ctx->small_cap) return 0;
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->small) LF_HIP(ctx, hipFree(ctx->small));
    ctx->small = nullptr;
    ctx->small_cap = 0;
    LF_HIP(ctx, hipMalloc((void**)&ctx->small, doubles * sizeof(double)));
    ctx->small_cap = doubles;
    return 0;
}

int ensure_nvec(lfpsqp_ctx* ctx, size_t doubles) {
    if (doubles <= ctx->nvec_cap) return 0;
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_nvec) LF_HIP(ctx, hipFree(ctx->d_nvec));
    ctx->d_nvec = nullptr;
    ctx->nvec_cap = 0;
    const size_t cap = (size_t)round_up((int64_t)doubles + 1, kPadRows);
    LF_HIP(ctx, hipMalloc((void**)&ctx->d_nvec, cap * sizeof(double)));
    LF_HIP(ctx, hipMemsetAsync(ctx->d_nvec, 0, cap * sizeof(double), ctx->stream));
    ctx->nvec_cap = cap;
    return 0;
}

int ensure_view(lfpsqp_ctx* ctx) {
    if (ctx->d_view) return 0;
    LF_HIP(ctx, hipMalloc((void**)&ctx->d_view, sizeof(double) * kViewScratch));
    LF_HIP(ctx, hipMemsetAsync(ctx->d_view, 0, sizeof(double) * kViewScratch, ctx->stream));
    return 0;
}

namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("LFPSQP_ROCTX");
        if (e && atoi(e) == 0) return;
        void* h = nullptr;
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"})   // (rocprofv3 --marker-trace reads the SDK's)
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr;
    }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
TraceRange::TraceRange(const char* name) : on(false) {
    Roctx& r = roctx();
    if (r.push) { r.push(name); on = true; }
}
TraceRange::~TraceRange() {
    if (on) roctx().pop();
}

int ensure_mvec(lfpsqp_ctx* ctx, size_t doubles) {
    ctx->pcg_resume.valid = false;        // whoever asks for the m-vector staging area is about to overwrite it
    if (doubles <= ctx->m_cap) return 0;
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_m) LF_HIP(ctx, hipFree(ctx->d_m));
    if (ctx->h_m) LF_HIP(ctx, hipHostFree(ctx->h_m));
    ctx->d_m = nullptr; ctx->h_m = nullptr; ctx->m_cap = 0;
    const size_t cap = (size_t)round_up((int64_t)doubles + 64, kPadRows);
    LF_HIP(ctx, hipMalloc((void**)&ctx->d_m, cap * sizeof(double)));
    LF_HIP(ctx, hipMemsetAsync(ctx->d_m, 0, cap * sizeof(double), ctx->stream));
    LF_HIP(ctx, hipHostMalloc((void**)&ctx->h_m, cap * sizeof(double)));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->m_cap = cap;
    return 0;
}

// ---- P2P: a one-shot all-reduce for small payloads over peer-mapped memory (SURVEY 5 "distributed communication backend") -------------
// Every collective of the hot loops is latency-bound (2m + 5 doubles per projected-CG iteration, m + 1 per Newton step): a ring costs
// 2 (N - 1) hops, a one-shot exchange one.  Each rank owns a MAILBOX in its device memory -- two slots (sequence parity) of kP2PSlot
// doubles + a sequence flag each -- that every other rank maps through hipIpc.  One workgroup per collective: copy the payload into my
// slot, publish the sequence number with a system-scope release store; then, rank by rank in FIXED order (so every rank adds in the same
// order: bit-identical results everywhere), wait for that rank's flag and accumulate its payload with system-scope loads.  Stream-ordered,
// no host involvement.  A slot is rewritten two collectives later: a rank posts sequence s + 1 only after it has finished reading every
// slot of sequence s, and I start s + 2 only after I have seen everybody's s + 1.
constexpr int kP2PSlot = 4096;                 // doubles per slot (32 KB); larger payloads go in pieces
constexpr int kP2PMaxRanks = 16;
struct P2PSlot { double data[kP2PSlot]; unsigned long long flag; unsigned long long pad[15]; };
struct P2PBox { P2PSlot slot[2]; };
struct P2PArgs {
    P2PBox* box[kP2PMaxRanks];
    int rank, nranks;
    double* buf;
    int count, op;
    unsigned long long seq;                    // >= 1
    int* err;
    long long timeout_ticks;                   // wall_clock64 ticks (100 MHz)
};
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(P2PArgs a) {
    P2PSlot* mine = &a.box[a.rank]->slot[a.seq & 1];
    for (int i = threadIdx.x; i < a.count; i += 256)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine->data) + i, __builtin_bit_cast(unsigned long long, a.buf[i]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __shared__ volatile int ok;             // cleared by any lane whose peer did not arrive (the only writers after the barrier all write 0)
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&mine->flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // every peer's flag is awaited by a lane of its own, all at once: the wait costs the slowest peer's arrival plus ONE round trip over the
    // links, not one round trip per peer
    if ((int)threadIdx.x < a.nranks && (int)threadIdx.x != a.rank) {
        const unsigned long long* f = &a.box[threadIdx.x]->slot[a.seq & 1].flag;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
            if (wall_clock64() - t0 > a.timeout_ticks) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __threadfence_system();
    __syncthreads();
    if (!ok) {   // a peer never arrived: poison the result, tell the host, do not hang the device
        if (threadIdx.x == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int i = threadIdx.x; i < a.count; i += 256) a.buf[i] = NAN;
        return;
    }
    // the payloads of all ranks are requested together (independent loads, one round trip) and added in FIXED rank order 0, 1, 2, ...:
    // the same bits on every rank, and the same bits as any other transport that adds in that order
    for (int i = threadIdx.x; i < a.count; i += 256) {
        double v[kP2PMaxRanks];
#pragma unroll
        for (int r = 0; r < kP2PMaxRanks; ++r)
            if (r < a.nranks)
                v[r] = __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<unsigned long long*>(a.box[r]->slot[a.seq & 1].data) + i,
                                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        double acc = v[0];
#pragma unroll
        for (int r = 1; r < kP2PMaxRanks; ++r)
            if (r < a.nranks) acc = (a.op == 1) ? nanmax(acc, v[r]) : acc + v[r];
        a.buf[i] = acc;
    }
}

static int p2p_allreduce(lfpsqp_ctx* ctx, double* buf, int64_t count, int op) {
    Comm& c = ctx->comm;
    for (int64_t off = 0; off < count; off += kP2PSlot) {
        P2PArgs a;
        for (int r = 0; r < kP2PMaxRanks; ++r) a.box[r] = static_cast<P2PBox*>(c.p2p_peer[r]);
        a.rank = c.rank; a.nranks = c.nranks;
        a.buf = buf + off;
        a.count = (int)std::min<int64_t>(kP2PSlot, count - off);
        a.op = op;
        a.seq = ++c.p2p_seq;
        a.err = c.p2p_err;
        a.timeout_ticks = 20LL * 100000000LL;   // 20 s
        hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(1), dim3(256), 0, ctx->stream, a);
        LF_LAUNCH_CHECK(ctx);
    }
    return 0;
}

int allreduce_dev(lfpsqp_ctx* ctx, double* buf, int64_t count, int op) {
    Comm& c = ctx->comm;
    if (c.kind == Comm::NONE || count == 0) return 0;
    if (c.kind == Comm::P2P) {
        if (c.p2p_err && *(volatile int*)c.p2p_err) return set_err(ctx, LFPSQP_ERR_COMM, "P2P all-reduce: a peer did not arrive within the time limit");
        return p2p_allreduce(ctx, buf, count, op);
    }
    if (c.kind == Comm::RCCL) {
        const int ncclFloat64 = 8, ncclSum = 0, ncclMax = 2;  // rccl.h enums
        int rc = c.ncclAllReduce(buf, buf, (size_t)count, ncclFloat64, op == 1 ? ncclMax : ncclSum, c.nccl_comm, ctx->stream);
        if (rc != 0)
            return set_err(ctx, LFPSQP_ERR_COMM, "ncclAllReduce failed: %s", c.ncclGetErrorString ? c.ncclGetErrorString(rc) : "?");
        return 0;
    }
    if (c.kind == Comm::CALLBACK) {
        int rc = c.cb(c.cb_user, buf, count, op, (void*)ctx->stream);
        if (rc != 0) return set_err(ctx, LFPSQP_ERR_COMM, "all-reduce callback returned %d", rc);
        return 0;
    }
    return set_err(ctx, LFPSQP_ERR_COMM, "nranks = %d but no communicator is initialised", c.nranks);
}

static int prof_lazy_init(lfpsqp_ctx* ctx) {
    if (ctx->prof_init) return 0;
    for (int s = 0; s < kProfSlots; ++s)
        for (int e = 0; e < kProfEvents; ++e)
            for (int k = 0; k < 2; ++k)
                if (hipEventCreate(&ctx->prof_ev[s][e][k]) != hipSuccess) return -1;
    ctx->prof_init = true;
    return 0;
}

// Every kProfStride-th launch of a slot is bracketed (an event record between two kernels costs a few microseconds of
// device time at the boundary -- per launch that was ~1 % of an iteration at n = 1e7 and ~6 % at the 8-GPU shard size).
constexpr int kProfStride = 4;
void prof_begin(lfpsqp_ctx* ctx, int s) {
    if (!ctx->profiling) return;
    ctx->prof_live[s] = (ctx->prof_seq[s]++ % kProfStride == 0) && ctx->prof_used[s] < kProfEvents;
    if (ctx->prof_live[s]) (void)hipEventRecord(ctx->prof_ev[s][ctx->prof_used[s]][0], ctx->stream);
}
void prof_end(lfpsqp_ctx* ctx, int s) {
    if (!ctx->profiling || !ctx->prof_live[s]) return;
    (void)hipEventRecord(ctx->prof_ev[s][ctx->prof_used[s]][1], ctx->stream);
    ctx->prof_used[s]++;
    ctx->prof_live[s] = false;
}
void prof_collect(lfpsqp_ctx* ctx) {
    if (!ctx->profiling) return;
    for (int s = 0; s < kProfSlots; ++s) {
        for (int e = 0; e < ctx->prof_used[s]; ++e) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->prof_ev[s][e][0], ctx->prof_ev[s][e][1]) == hipSuccess) {
                ctx->prof_ms[s] += ms;
                ctx->prof_count[s] += 1;
            }
        }
        ctx->prof_used[s] = 0;
    }
}

int read_back(lfpsqp_ctx* ctx, const double* dev, double* host, int64_t count) {
    if (count > 64) return set_err(ctx, LFPSQP_ERR_ARG, "read_back: count too large");
    LF_HIP(ctx, hipMemcpyAsync(ctx->h_scal, dev, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(host, ctx->h_scal, sizeof(double) * count);
    return 0;
}

}  // namespace lfpsqp

using namespace lfpsqp;

// (device buffers come from hipMalloc; the virtual-memory-API experiment of round 3 -- FINDINGS.md 6, tools/archive/gpu_vmm_probe.sh -- was a
// variant build and is no longer in the sources)
namespace {
inline hipError_t dev_alloc(void** out, size_t bytes) { return hipMalloc(out, bytes); }
inline void dev_free(void* p) { (void)hipFree(p); }
}  // namespace

// dst[:, j] = rs .* src[:, j] + u w_j: the matrix a view stands for (lfpsqp_mat_copy of a view; 16 columns per block row)
__global__ __launch_bounds__(lfpsqp::kThreads) void view_copy_kernel(const double* __restrict__ A, const double* __restrict__ rs, const double* __restrict__ u,
                                                                     const double* __restrict__ w, double* __restrict__ D, int64_t ld, int64_t n, int m) {
    const int64_t i = ((int64_t)blockIdx.x * lfpsqp::kThreads + threadIdx.x) * 2;
    if (i >= n) return;
    const double2 s = rs ? lfpsqp::ld2(rs + i) : make_double2(1.0, 1.0);
    const double2 uu = u ? lfpsqp::ld2(u + i) : make_double2(0.0, 0.0);
    const bool v1 = i + 1 < n;
    const int j0 = blockIdx.y * 16, j1 = (j0 + 16 < m) ? j0 + 16 : m;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
        const double2 c = lfpsqp::ld2(A + (int64_t)j * ld + i);
        const double wj = u ? w[j] : 0.0;
        const double2 o = make_double2(fma(uu.x, wj, s.x * c.x), fma(uu.y, wj, s.y * c.y));
        if (v1) lfpsqp::st2(D + (int64_t)j * ld + i, o);
        else D[(int64_t)j * ld + i] = o.x;
    }
}

extern "C" {

int lfpsqp_ctx_create(int device, lfpsqp_ctx** out) {
    if (!out) return LFPSQP_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(nullptr, LFPSQP_ERR_HIP, "no HIP device visible");
    if (device < 0 || device >= ndev) return set_err(nullptr, LFPSQP_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
    lfpsqp_ctx* ctx = new lfpsqp_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return set_err(nullptr, LFPSQP_ERR_HIP, "hipSetDevice(%d) failed", device); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->devname = std::string(prop.name) + " (" + prop.gcnArchName + ")";
        ctx->num_cu = prop.multiProcessorCount;
        ctx->real_gpu = strncmp(prop.gcnArchName, "gfx", 3) == 0;
    }
    bool ok = hipStreamCreate(&ctx->stream) == hipSuccess;
    ok = ok && hipMalloc((void**)&ctx->scal, 64 * sizeof(double)) == hipSuccess;
    ok = ok && hipMalloc((void**)&ctx->istat, 128 * sizeof(int64_t)) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&ctx->h_scal, 4 * 64 * sizeof(double)) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&ctx->h_istat, 4 * 16 * sizeof(int64_t)) == hipSuccess;
    for (int i = 0; ok && i < 4; ++i) ok = hipEventCreate(&ctx->ev_slot[i]) == hipSuccess;
    ok = ok && hipEventCreate(&ctx->ev_t0) == hipSuccess && hipEventCreate(&ctx->ev_t1) == hipSuccess;
    if (ok) ok = hipMemset(ctx->scal, 0, 64 * sizeof(double)) == hipSuccess && hipMemset(ctx->istat, 0, 128 * sizeof(int64_t)) == hipSuccess;
    if (!ok) { delete ctx; return set_err(nullptr, LFPSQP_ERR_HIP, "context resource allocation failed"); }
    if (const char* e = getenv("LFPSQP_ONEPASS")) ctx->tune_onepass = atoi(e);
    if (const char* e = getenv("LFPSQP_SPGRAM")) ctx->tune_spgram = atoi(e) < 0 ? -1 : 0;
    if (const char* e = getenv("LFPSQP_VEC_BLOCKS")) ctx->tune_vec_blocks = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("LFPSQP_NRB_MFMA")) ctx->tune_nrb_mfma = atoi(e) > 0 ? 1 : (atoi(e) < 0 ? -1 : 0);
    if (const char* e = getenv("LFPSQP_GPING")) ctx->tune_gping = atoi(e) == 1 ? 1 : 0;
    if (const char* e = getenv("LFPSQP_NRB_WG_CAP")) ctx->batch_wg_cap = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("LFPSQP_STAGE_ROUNDS")) ctx->stage_cap = atoi(e) > 0 ? atoi(e) : 0;
    *out = ctx;
    return 0;
}

int lfpsqp_ctx_destroy(lfpsqp_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm.kind == Comm::RCCL && ctx->comm.nccl_comm && ctx->comm.ncclCommDestroy) ctx->comm.ncclCommDestroy(ctx->comm.nccl_comm);
    for (int r = 0; r < kP2PMaxRanks; ++r)                     // the P2P transport's mappings and this rank's mailbox
        if (ctx->comm.p2p_peer[r] && ctx->comm.p2p_peer[r] != ctx->comm.p2p_mine) (void)hipIpcCloseMemHandle(ctx->comm.p2p_peer[r]);
    if (ctx->comm.p2p_mine) (void)hipFree(ctx->comm.p2p_mine);
    if (ctx->comm.p2p_err) (void)hipHostFree(ctx->comm.p2p_err);
    if (ctx->prof_init)
        for (int s = 0; s < kProfSlots; ++s)
            for (int e = 0; e < kProfEvents; ++e)
                for (int k = 0; k < 2; ++k) (void)hipEventDestroy(ctx->prof_ev[s][e][k]);
    for (int i = 0; i < 4; ++i) if (ctx->ev_slot[i]) (void)hipEventDestroy(ctx->ev_slot[i]);
    if (ctx->ev_t0) (void)hipEventDestroy(ctx->ev_t0);
    if (ctx->ev_t1) (void)hipEventDestroy(ctx->ev_t1);
    if (ctx->part) (void)hipFree(ctx->part);
    if (ctx->small) (void)hipFree(ctx->small);
    if (ctx->d_m) (void)hipFree(ctx->d_m);
    if (ctx->d_qw) (void)hipFree(ctx->d_qw);
    if (ctx->d_zeros) (void)hipFree(ctx->d_zeros);
    if (ctx->d_nvec) (void)hipFree(ctx->d_nvec);
    if (ctx->d_tri) (void)hipFree(ctx->d_tri);
    if (ctx->d_view) (void)hipFree(ctx->d_view);
    if (ctx->h_m) (void)hipHostFree(ctx->h_m);
    if (ctx->scal) (void)hipFree(ctx->scal);
    if (ctx->istat) (void)hipFree(ctx->istat);
    if (ctx->h_scal) (void)hipHostFree(ctx->h_scal);
    if (ctx->h_istat) (void)hipHostFree(ctx->h_istat);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int lfpsqp_ctx_sync(lfpsqp_ctx* ctx) {
    LF_ARG(ctx, ctx != nullptr);
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

const char* lfpsqp_last_error(const lfpsqp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int lfpsqp_device_name(const lfpsqp_ctx* ctx, char* buf, int64_t buflen) {
    if (!ctx || !buf || buflen <= 0) return LFPSQP_ERR_ARG;
    snprintf(buf, (size_t)buflen, "%s", ctx->devname.c_str());
    return 0;
}

// "GPU-<16 hex digits>" as rocminfo / rocm-smi print it, of the device THIS context computes on (hipDeviceGetUuid: the 16 bytes are the
// hex digits themselves on AMD devices; anything else is hex-encoded)
int lfpsqp_device_uuid(const lfpsqp_ctx* ctx, char* buf, int64_t buflen) {
    if (!ctx || !buf || buflen < 40) return LFPSQP_ERR_ARG;
    hipUUID id;
    memset(&id, 0, sizeof(id));
    hipDevice_t d;
    if (hipDeviceGet(&d, ctx->device) != hipSuccess || hipDeviceGetUuid(&id, d) != hipSuccess) {
        (void)hipGetLastError();
        snprintf(buf, (size_t)buflen, "unknown");
        return 0;
    }
    bool hexchars = true;
    for (int i = 0; i < 16; ++i) hexchars = hexchars && isxdigit((unsigned char)id.bytes[i]);
    int o = snprintf(buf, (size_t)buflen, "GPU-");
    for (int i = 0; i < 16; ++i)
        o += hexchars ? snprintf(buf + o, (size_t)buflen - o, "%c", id.bytes[i]) : snprintf(buf + o, (size_t)buflen - o, "%02x", (unsigned char)id.bytes[i]);
    return 0;
}

int lfpsqp_timer_begin(lfpsqp_ctx* ctx) {
    LF_ARG(ctx, ctx != nullptr);
    LF_HIP(ctx, hipEventRecord(ctx->ev_t0, ctx->stream));
    return 0;
}
int lfpsqp_timer_end(lfpsqp_ctx* ctx, double* ms) {
    LF_ARG(ctx, ctx != nullptr && ms != nullptr);
    LF_HIP(ctx, hipEventRecord(ctx->ev_t1, ctx->stream));
    LF_HIP(ctx, hipEventSynchronize(ctx->ev_t1));
    float f = 0.f;
    LF_HIP(ctx, hipEventElapsedTime(&f, ctx->ev_t0, ctx->ev_t1));
    *ms = f;
    return 0;
}

int lfpsqp_ctx_set_tuning(lfpsqp_ctx* ctx, int ks, int nt) {
    LF_ARG(ctx, ctx != nullptr && (ks == 0 || ks == 2 || ks == 4));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tune_ks = ks;
    ctx->tune_nt = nt != 0;
    return 0;
}

int lfpsqp_ctx_set_onepass(lfpsqp_ctx* ctx, int mode) {
    LF_ARG(ctx, ctx != nullptr && (mode == 0 || mode == -1));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tune_onepass = mode;
    return 0;
}

int lfpsqp_ctx_set_residual_buffers(lfpsqp_ctx* ctx, int mode) {
    LF_ARG(ctx, ctx != nullptr && (mode == 0 || mode == 1));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tune_gping = mode;
    ctx->pcg_resume.valid = false;          // a resumable solve is tied to the scheme it ran with
    return 0;
}

int lfpsqp_ctx_set_profiling(lfpsqp_ctx* ctx, int on) {
    LF_ARG(ctx, ctx != nullptr);
    if (on && prof_lazy_init(ctx) != 0) return set_err(ctx, LFPSQP_ERR_HIP, "profiling event creation failed");
    ctx->profiling = on != 0;
    for (int s = 0; s < kProfSlots; ++s) { ctx->prof_used[s] = 0; ctx->prof_count[s] = 0; ctx->prof_ms[s] = 0.0; ctx->prof_seq[s] = 0; ctx->prof_live[s] = false; }
    return 0;
}
int lfpsqp_profile_read(lfpsqp_ctx* ctx, double ms[8], int64_t counts[8]) {
    LF_ARG(ctx, ctx != nullptr && ms && counts);
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    for (int s = 0; s < kProfSlots; ++s) { ms[s] = ctx->prof_ms[s]; counts[s] = ctx->prof_count[s]; }
    return 0;
}

/* ---- sharding / communicator -------------------------------------------- */

int lfpsqp_shard_range(int64_t n, int rank, int nranks, int64_t* row0, int64_t* row1) {
    if (n < 0 || nranks <= 0 || rank < 0 || rank >= nranks || !row0 || !row1) return LFPSQP_ERR_ARG;
    // contiguous blocks; boundaries on whole tiles so no tile straddles two ranks
    const int64_t tiles = (n + kPadRows - 1) / kPadRows;
    const int64_t base = tiles / nranks, extra = tiles % nranks;
    const int64_t t0 = (int64_t)rank * base + (rank < extra ? rank : extra);
    const int64_t t1 = t0 + base + (rank < extra ? 1 : 0);
    int64_t r0 = t0 * kPadRows, r1 = t1 * kPadRows;
    if (r0 > n) r0 = n;
    if (r1 > n) r1 = n;
    *row0 = r0;
    *row1 = r1;
    return 0;
}

static int load_rccl(lfpsqp_ctx* ctx) {
    Comm& c = ctx->comm;
    if (c.rccl_lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        c.rccl_lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (c.rccl_lib) break;
    }
    if (!c.rccl_lib) return set_err(ctx, LFPSQP_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
    return 0;
}

int lfpsqp_comm_unique_id(lfpsqp_ctx* ctx, void* id128) {
    LF_ARG(ctx, ctx != nullptr && id128 != nullptr);
    LF_TRY(load_rccl(ctx));
    typedef int (*get_id_fn)(void*);
    get_id_fn f = (get_id_fn)dlsym(ctx->comm.rccl_lib, "ncclGetUniqueId");
    if (!f) return set_err(ctx, LFPSQP_ERR_COMM, "ncclGetUniqueId not found");
    int rc = f(id128);
    if (rc != 0) return set_err(ctx, LFPSQP_ERR_COMM, "ncclGetUniqueId failed (%d)", rc);
    return 0;
}

int lfpsqp_comm_init_rccl(lfpsqp_ctx* ctx, int rank, int nranks, const void* id128) {
    LF_ARG(ctx, ctx != nullptr && id128 != nullptr && nranks >= 1 && rank >= 0 && rank < nranks);
    LF_TRY(load_rccl(ctx));
    Comm& c = ctx->comm;
    // ncclUniqueId is a 128-byte struct passed BY VALUE
    struct id_t { char b[128]; } id;
    memcpy(&id, id128, 128);
    typedef int (*init_fn)(void**, int, id_t, int);
    init_fn init = (init_fn)dlsym(c.rccl_lib, "ncclCommInitRank");
    c.ncclAllReduce = (decltype(c.ncclAllReduce))dlsym(c.rccl_lib, "ncclAllReduce");
    c.ncclCommDestroy = (decltype(c.ncclCommDestroy))dlsym(c.rccl_lib, "ncclCommDestroy");
    c.ncclGetErrorString = (decltype(c.ncclGetErrorString))dlsym(c.rccl_lib, "ncclGetErrorString");
    if (!init || !c.ncclAllReduce) return set_err(ctx, LFPSQP_ERR_COMM, "RCCL symbols not found");
    LF_HIP(ctx, hipSetDevice(ctx->device));
    int rc = init(&c.nccl_comm, nranks, id, rank);
    if (rc != 0) return set_err(ctx, LFPSQP_ERR_COMM, "ncclCommInitRank failed: %s", c.ncclGetErrorString ? c.ncclGetErrorString(rc) : "?");
    c.kind = Comm::RCCL;
    c.rank = rank;
    c.nranks = nranks;
    return 0;
}

int lfpsqp_comm_init_callback(lfpsqp_ctx* ctx, int rank, int nranks, lfpsqp_allreduce_fn fn, void* user) {
    LF_ARG(ctx, ctx != nullptr && fn != nullptr && nranks >= 1 && rank >= 0 && rank < nranks);
    ctx->comm.kind = Comm::CALLBACK;
    ctx->comm.cb = fn;
    ctx->comm.cb_user = user;
    ctx->comm.rank = rank;
    ctx->comm.nranks = nranks;
    return 0;
}

// P2P transport.  lfpsqp_comm_p2p_export: allocate this rank's mailbox, return its 64-byte IPC handle; ship the handles of all ranks to
// every rank by any control plane; lfpsqp_comm_init_p2p maps them.  Works across the GPUs of one node (xGMI / PCIe peer access) and for
// several ranks sharing one GPU (the functional test on a 1-GPU box).
int lfpsqp_comm_p2p_export(lfpsqp_ctx* ctx, void* handle64) {
    LF_ARG(ctx, ctx && handle64);
    Comm& c = ctx->comm;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the header promises a 64-byte handle");
    hipIpcMemHandle_t h;
    if (!c.p2p_mine) {
        // Fine-grained (uncached) device memory: no cache between a peer and the mailbox.  If the runtime cannot allocate or export that
        // kind, the call FAILS -- unless ordinary (coarse-grained) device memory was explicitly allowed (lfpsqp_comm_p2p_allow_coarse /
        // LFPSQP_P2P_ALLOW_COARSE=1): every access of the protocol is a system-scope atomic either way, and ranks sharing ONE GPU work with
        // it, but across xGMI that kind is untested, so it is never chosen silently.  lfpsqp_comm_p2p_info reports what was allocated.
        hipError_t e = hipExtMallocWithFlags(&c.p2p_mine, sizeof(P2PBox), hipDeviceMallocUncached);
        if (e == hipSuccess) e = hipIpcGetMemHandle(&h, c.p2p_mine);
        c.p2p_memkind = LFPSQP_P2P_MEM_FINE;
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (c.p2p_mine) (void)hipFree(c.p2p_mine);
            c.p2p_mine = nullptr;
            c.p2p_memkind = LFPSQP_P2P_MEM_NONE;
            const char* env = getenv("LFPSQP_P2P_ALLOW_COARSE");
            if (!(c.p2p_allow_coarse || (env && atoi(env) > 0)))
                return set_err(ctx, LFPSQP_ERR_COMM, "P2P mailbox: fine-grained (uncached) device memory cannot be allocated / exported through hipIpc (%s); "
                               "coarse-grained memory was not allowed (lfpsqp_comm_p2p_allow_coarse, LFPSQP_P2P_ALLOW_COARSE=1)", hipGetErrorString(e));
            LF_HIP(ctx, hipMalloc(&c.p2p_mine, sizeof(P2PBox)));
            c.p2p_memkind = LFPSQP_P2P_MEM_COARSE;
        }
        LF_HIP(ctx, hipMemset(c.p2p_mine, 0, sizeof(P2PBox)));
        LF_HIP(ctx, hipDeviceSynchronize());      // the zeros are in memory before the handle leaves this process
    }
    LF_HIP(ctx, hipIpcGetMemHandle(&h, c.p2p_mine));
    memcpy(handle64, &h, 64);
    return 0;
}
int lfpsqp_comm_p2p_allow_coarse(lfpsqp_ctx* ctx, int allow) {
    if (!ctx) return LFPSQP_ERR_ARG;
    ctx->comm.p2p_allow_coarse = allow != 0;
    return 0;
}
int lfpsqp_comm_p2p_info(const lfpsqp_ctx* ctx, int* mem_kind, unsigned long long* collectives) {
    if (!ctx) return LFPSQP_ERR_ARG;
    if (mem_kind) *mem_kind = ctx->comm.p2p_memkind;
    if (collectives) *collectives = ctx->comm.p2p_seq;
    return 0;
}
// Ordering the transport relies on (documented in the header): export zero-fills the mailbox and completes before the handle is returned; a
// peer can touch my mailbox only after it received that handle; its first collective WRITES only its own mailbox and READS mine.  So no
// barrier is needed between init and the first collective.  A context is initialised ONCE: a second init (a re-rendezvous after a timeout)
// would restart the sequence numbers over mailboxes that still hold flags of the first life, and waits would pass on stale slots --
// rejected; create a new context instead.
int lfpsqp_comm_init_p2p(lfpsqp_ctx* ctx, int rank, int nranks, const void* handles) {
    LF_ARG(ctx, ctx && handles && nranks >= 1 && nranks <= kP2PMaxRanks && rank >= 0 && rank < nranks && ctx->comm.p2p_mine);
    Comm& c = ctx->comm;
    if (c.kind != Comm::NONE)
        return set_err(ctx, LFPSQP_ERR_ARG, "lfpsqp_comm_init_p2p: this context already has a communicator (a P2P communicator cannot be "
                       "re-initialised: its mailboxes keep the sequence flags of their first life); create a new context");
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) { c.p2p_peer[r] = c.p2p_mine; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char*>(handles) + 64 * (size_t)r, 64);
        LF_HIP(ctx, hipIpcOpenMemHandle(&c.p2p_peer[r], h, hipIpcMemLazyEnablePeerAccess));
    }
    if (!c.p2p_err) {
        LF_HIP(ctx, hipHostMalloc((void**)&c.p2p_err, sizeof(int)));
        *c.p2p_err = 0;
    }
    c.rank = rank; c.nranks = nranks; c.kind = Comm::P2P; c.p2p_seq = 0;
    return 0;
}

int lfpsqp_comm_info(const lfpsqp_ctx* ctx, int* rank, int* nranks) {
    if (!ctx) return LFPSQP_ERR_ARG;
    if (rank) *rank = ctx->comm.rank;
    if (nranks) *nranks = ctx->comm.nranks;
    return 0;
}

/* ---- buffers -------------------------------------------------------------- */


int lfpsqp_vec_alloc(lfpsqp_ctx* ctx, int64_t n, lfpsqp_vec** out) {
    LF_ARG(ctx, ctx != nullptr && out != nullptr && n >= 0);
    lfpsqp_vec* v = new lfpsqp_vec();
    v->n = n;
    v->cap = round_up(n > 0 ? n : 1, kPadRows);
    hipError_t e = dev_alloc((void**)&v->p, sizeof(double) * v->cap);
    if (e != hipSuccess) {
        (void)hipGetLastError();      // clear HIP's sticky last error: a caller that carries on (a linesearch batching as many trials as fit) must not see it at its next launch check
        delete v;
        return set_err(ctx, LFPSQP_ERR_HIP, "hipMalloc(%lld doubles) failed: %s", (long long)n, hipGetErrorString(e));
    }
    e = hipMemsetAsync(v->p, 0, sizeof(double) * v->cap, ctx->stream);
    if (e != hipSuccess) { dev_free(v->p); delete v; return set_err(ctx, LFPSQP_ERR_HIP, "hipMemsetAsync failed"); }
    *out = v;
    return 0;
}

int lfpsqp_vec_free(lfpsqp_ctx* ctx, lfpsqp_vec* v) {
    if (!v) return 0;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (v->slab) {                                   // a member of a placement-tuned set: the set's allocation goes with its last member
        if (--v->slab->refs == 0) { dev_free(v->slab->p); delete v->slab; }
    } else if (v->p) dev_free(v->p);
    delete v;
    return 0;
}

// ---- placement policy (FINDINGS.md 6) ------------------------------------------------------------------------------------------------
// On MI355X the kernels that run a small store stream inside a matrix read stream (the fused projected-CG iteration, the Newton step, pcg!)
// run at one of two speeds -- 10-15 % apart -- depending on WHERE the matrix and the n-vectors they touch were allocated: a property of the
// pair of allocations, reproducible within a process (profiles/r03b_*: same traffic and cache hits, fewer requests in flight and +57 %
// latency-FIFO stall cycles in the vector caches on a slow pair).  The library therefore allocates such buffers by trial: a few candidate
// allocations, a few launches of the fused kernel itself on each (on zeros), the fastest kept, the others freed.
int lfpsqp_ctx_set_placement(lfpsqp_ctx* ctx, int tries) {
    LF_ARG(ctx, ctx && tries >= 1 && tries <= 8);
    ctx->place_tries = tries;
    return 0;
}

int lfpsqp_ctx_set_nr_batch_mode(lfpsqp_ctx* ctx, int mode) {
    LF_ARG(ctx, ctx && (mode == LFPSQP_NR_BATCH_EXACT || mode == LFPSQP_NR_BATCH_MATRIX_CORES));
    ctx->tune_nrb_mfma = mode == LFPSQP_NR_BATCH_MATRIX_CORES ? 0 : -1;
    return 0;
}

int lfpsqp_placement_info(const lfpsqp_ctx* ctx, int* tries, int* picked, double* ms, int ms_cap) {
    if (!ctx) return LFPSQP_ERR_ARG;
    if (tries) *tries = ctx->place_last_n;
    if (picked) *picked = ctx->place_last_pick;
    if (ms) for (int k = 0; k < ms_cap; ++k) ms[k] = k < ctx->place_last_n ? ctx->place_last_ms[k] : 0.0;
    return 0;
}

// the trial on its own (diagnostics, tools/placement_probe_check.py): g, d, a must be zero-filled n-vectors
int lfpsqp_placement_probe(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols, lfpsqp_vec* g, lfpsqp_vec* d, lfpsqp_vec* a, int reps, double* ms) {
    LF_ARG(ctx, ctx && M && g && d && a && ms && reps >= 1 && g->n >= M->n && d->n >= M->n && a->n >= M->n);
    return lfpsqp::placement_probe(ctx, M, (int)ncols, g->p, d->p, a->p, reps, ms);
}

static int64_t mat_ld_skew();
static bool worth_placing(const lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols) {
    return ctx->place_tries > 1 && ctx->real_gpu && M && M->n > 0 && ncols >= 4 && ncols <= M->m &&
           (int64_t)sizeof(double) * M->ld * ncols >= ctx->place_min_bytes && getenv("LFPSQP_PLACEMENT_OFF") == nullptr;
}

// `count` vectors of n doubles each as views of one slab
static void slab_views(void* base, int64_t n, int64_t cap, int count, lfpsqp_vec** out) {
    lfpsqp_slab* slab = new lfpsqp_slab();
    slab->p = base;
    slab->refs = count;
    for (int k = 0; k < count; ++k) {
        lfpsqp_vec* v = new lfpsqp_vec();
        v->p = (double*)base + (size_t)k * cap;
        v->n = n;
        v->cap = cap;
        v->slab = slab;
        out[k] = v;
    }
}

// The trial itself: every (matrix candidate, vector-slab candidate) pair, two rounds over all pairs (a GPU coming out of idle runs its first
// launches several per cent slower: the first round doubles as the warm-up, a pair's time is the smaller of its two), the fastest pair kept.
// mats[0..nm), slabs[0..nv): live candidates.  Returns the chosen indices; ms_out (nm * nv, row = matrix) for lfpsqp_placement_info.
static int place_grid(lfpsqp_ctx* ctx, lfpsqp_mat* const* mats, int nm, int ncols, void* const* slabs, int nv, int64_t cap, int* bm, int* bv,
                      double* ms_out) {
    *bm = 0; *bv = 0;
    double best = 1e300;
    for (int k = 0; k < nm * nv; ++k) ms_out[k] = 1e300;
    if (nm * nv <= 1) { ms_out[0] = 0.0; return 0; }
    for (int round = 0; round < 2; ++round)
        for (int i = 0; i < nm; ++i)
            for (int j = 0; j < nv; ++j) {
                double* base = (double*)slabs[j];
                double t = -1.0;
                LF_TRY(lfpsqp::placement_probe(ctx, mats[i], ncols, base, base + cap, base + 2 * cap, round == 0 ? 2 : 3, &t));
                if (t < 0.0) { ms_out[0] = 0.0; return 0; }            // no one-pass kernel for this shape: nothing to choose
                if (t < ms_out[i * nv + j]) ms_out[i * nv + j] = t;
            }
    for (int i = 0; i < nm; ++i)
        for (int j = 0; j < nv; ++j)
            if (ms_out[i * nv + j] < best) { best = ms_out[i * nv + j]; *bm = i; *bv = j; }
    return 0;
}

// Did no candidate stand out (all trials within 1.5 % of the fastest)?  Pairs come in two kinds about 3.5 % apart (a matrix and a vector
// slab "collide" or they do not, tools/placement_pairs_probe.py: consecutive slabs tend to be of one kind against a given matrix), so a
// uniform table may be all collisions -- the callers then try as many vector slabs again, once: a slab is small next to the matrix.
static bool place_uniform(const double* ms, int n) {
    if (const char* e = getenv("LFPSQP_PLACEMENT_EXTEND")) return atoi(e) != 0 && n > 1;      // (tests: force / forbid the second batch)
    double lo = 1e300, hi = 0.0;
    for (int k = 0; k < n; ++k) { lo = ms[k] < lo ? ms[k] : lo; hi = ms[k] > hi ? ms[k] : hi; }
    return n > 1 && lo > 0.0 && hi <= 1.015 * lo;
}

// one zero-filled candidate slab; a candidate that could not be allocated OR zeroed does not exist (nothing is left behind)
static bool slab_alloc_zeroed(lfpsqp_ctx* ctx, void** p, size_t bytes) {
    *p = nullptr;
    if (dev_alloc(p, bytes) != hipSuccess) { *p = nullptr; (void)hipGetLastError(); return false; }
    if (hipMemsetAsync(*p, 0, bytes, ctx->stream) != hipSuccess) { (void)hipGetLastError(); dev_free(*p); *p = nullptr; return false; }
    return true;
}
// A trial that FAILED (a launch or an event error inside place_grid) is "no preference": the first candidates are kept and the allocation
// succeeds -- the policy is an optimisation, its failure must neither fail nor leak the allocation it serves.
static void place_no_preference(lfpsqp_ctx* ctx, int* bm, int* bv) {
    (void)hipGetLastError();
    *bm = 0; *bv = 0;
    ctx->place_last_n = 0;
}

static void place_record(lfpsqp_ctx* ctx, int n, int pick, const double* ms) {
    ctx->place_last_n = n > 64 ? 64 : n;
    ctx->place_last_pick = pick;
    for (int k = 0; k < 64; ++k) ctx->place_last_ms[k] = k < n ? ms[k] : 0.0;
}

int lfpsqp_vecs_alloc_placed(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols, int64_t n, int count, lfpsqp_vec** out) {
    LF_ARG(ctx, ctx && out && n >= 0 && count >= 1 && count <= 64);
    for (int k = 0; k < count; ++k) out[k] = nullptr;
    const int64_t cap = round_up(n > 0 ? n : 1, kPadRows);
    const size_t bytes = sizeof(double) * (size_t)cap * count;
    const int tries = (worth_placing(ctx, M, ncols) && count >= 3 && n >= M->n) ? ctx->place_tries : 1;
    void* cand[8] = {nullptr};
    int got = 0;
    for (int k = 0; k < tries; ++k) {
        if (!slab_alloc_zeroed(ctx, &cand[got], bytes)) break;
        ++got;
    }
    if (got == 0) return set_err(ctx, LFPSQP_ERR_HIP, "hipMalloc(%d vectors of %lld doubles) failed", count, (long long)n);
    int bm = 0, bv = 0;
    double ms[8] = {0};
    lfpsqp_mat* mats[1] = {const_cast<lfpsqp_mat*>(M)};
    bool trial_ok = got <= 1 || place_grid(ctx, mats, 1, (int)ncols, cand, got, cap, &bm, &bv, ms) == 0;
    if (trial_ok && got == tries && tries > 1 && 2 * tries <= 8 && place_uniform(ms, got)) {      // nobody stood out: as many slabs again
        int more = 0;
        for (int k = got; k < 2 * tries; ++k) {
            if (!slab_alloc_zeroed(ctx, &cand[got + more], bytes)) break;
            ++more;
        }
        if (more > 0) {
            int bm2 = 0, bv2 = 0;
            trial_ok = place_grid(ctx, mats, 1, (int)ncols, cand + got, more, cap, &bm2, &bv2, ms + got) == 0;
            if (more == 1) ms[got] = ms[bv];                      // (a single extra candidate is not timed by place_grid: leave the choice alone)
            else if (trial_ok && ms[got + bv2] < ms[bv]) bv = got + bv2;
            got += more;
        }
    }
    if (!trial_ok) place_no_preference(ctx, &bm, &bv);
    const hipError_t es = hipStreamSynchronize(ctx->stream);      // the zero fills (and the trials) are done before anything is freed or handed out
    for (int k = 0; k < got; ++k)
        if (k != bv || es != hipSuccess) dev_free(cand[k]);
    if (es != hipSuccess) return set_err(ctx, LFPSQP_ERR_HIP, "hipStreamSynchronize failed while allocating %d vectors: %s", count, hipGetErrorString(es));
    if (trial_ok) place_record(ctx, got > 1 ? got : 0, bv, ms);
    slab_views(cand[bv], n, cap, count, out);
    return 0;
}

// candidate matrices (as many of `tries` as fit side by side in three quarters of the free memory)
static int mat_candidates(lfpsqp_ctx* ctx, int64_t n, int64_t m, int tries, lfpsqp_mat** cand) {
    lfpsqp_mat shape;
    shape.n = n; shape.m = m;
    shape.ld = round_up(n > 0 ? n : 1, kPadRows) + mat_ld_skew();
    const size_t bytes = sizeof(double) * (size_t)shape.ld * (size_t)(m > 0 ? m : 1);
    if (!worth_placing(ctx, &shape, m)) tries = 1;
    size_t free_b = 0, total_b = 0;
    if (tries > 1 && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t room = free_b - free_b / 4;
        while (tries > 1 && (size_t)tries * bytes > room) --tries;
    }
    int got = 0;
    for (int k = 0; k < tries; ++k) {
        if (lfpsqp_mat_alloc(ctx, n, m, &cand[k]) != 0) { cand[k] = nullptr; (void)hipGetLastError(); break; }
        ++got;
    }
    return got;
}

int lfpsqp_mat_alloc_placed(lfpsqp_ctx* ctx, int64_t n, int64_t m, lfpsqp_mat** out) {
    LF_ARG(ctx, ctx != nullptr && out != nullptr && n >= 0 && m >= 0);
    lfpsqp_mat* cand[8] = {nullptr};
    const int got = mat_candidates(ctx, n, m, ctx->place_tries, cand);
    if (got == 0) return set_err(ctx, LFPSQP_ERR_HIP, "hipMalloc(%lld x %lld matrix) failed", (long long)n, (long long)m);
    int bm = 0, bv = 0;
    bool trial_ok = true;
    double ms[8] = {0};
    if (got > 1) {
        const int64_t cap = round_up(n > 0 ? n : 1, kPadRows);
        void* scratch = nullptr;
        if (slab_alloc_zeroed(ctx, &scratch, sizeof(double) * (size_t)cap * 3))
            trial_ok = place_grid(ctx, cand, got, (int)m, &scratch, 1, cap, &bm, &bv, ms) == 0;
        else
            trial_ok = false;
        (void)hipStreamSynchronize(ctx->stream);
        if (scratch) dev_free(scratch);
    }
    if (!trial_ok) place_no_preference(ctx, &bm, &bv);
    for (int k = 0; k < got; ++k)
        if (k != bm) lfpsqp_mat_free(ctx, cand[k]);
    if (trial_ok) place_record(ctx, got > 1 ? got : 0, bm, ms);
    *out = cand[bm];
    return 0;
}

// The basis and the n-vectors streamed with it, allocated TOGETHER: which allocation of the one is fast depends on the other (a property of
// the pair), so every (matrix candidate, vector-set candidate) pair is tried and the fastest pair kept.
int lfpsqp_basis_work_alloc_placed(lfpsqp_ctx* ctx, int64_t n, int64_t m, int64_t nvec, int count, lfpsqp_mat** M_out, lfpsqp_vec** out) {
    LF_ARG(ctx, ctx && M_out && out && n >= 0 && m >= 0 && nvec >= n && count >= 3 && count <= 64);
    *M_out = nullptr;
    for (int k = 0; k < count; ++k) out[k] = nullptr;
    lfpsqp_mat* cand[8] = {nullptr};
    const int gm = mat_candidates(ctx, n, m, ctx->place_tries, cand);
    if (gm == 0) return set_err(ctx, LFPSQP_ERR_HIP, "hipMalloc(%lld x %lld matrix) failed", (long long)n, (long long)m);
    const int64_t cap = round_up(nvec > 0 ? nvec : 1, kPadRows);
    const size_t bytes = sizeof(double) * (size_t)cap * count;
    const int tv = worth_placing(ctx, cand[0], m) ? ctx->place_tries : 1;
    void* slabs[8] = {nullptr};
    int gv = 0;
    for (int k = 0; k < tv; ++k) {
        if (!slab_alloc_zeroed(ctx, &slabs[gv], bytes)) break;
        ++gv;
    }
    if (gv == 0) {
        for (int k = 0; k < gm; ++k) lfpsqp_mat_free(ctx, cand[k]);
        return set_err(ctx, LFPSQP_ERR_HIP, "hipMalloc(%d vectors of %lld doubles) failed", count, (long long)nvec);
    }
    int bm = 0, bv = 0;
    double ms[64] = {0};
    bool trial_ok = gm * gv <= 1 || place_grid(ctx, cand, gm, (int)m, slabs, gv, cap, &bm, &bv, ms) == 0;
    if (trial_ok && gm * gv > 1 && gv == tv && tv > 1 && 2 * tv <= 8 && place_uniform(ms, gm * gv)) {   // nobody stood out: as many slabs again
        int more = 0;
        for (int k = gv; k < 2 * tv; ++k) {
            if (!slab_alloc_zeroed(ctx, &slabs[gv + more], bytes)) break;
            ++more;
        }
        if (more > 0 && gm * more > 1) {
            double ms2[64];
            int bm2 = 0, bv2 = 0;
            trial_ok = place_grid(ctx, cand, gm, (int)m, slabs + gv, more, cap, &bm2, &bv2, ms2) == 0;
            if (trial_ok) {
                double all[64];
                const int nv2 = gv + more;
                for (int i = 0; i < gm; ++i) {
                    for (int j = 0; j < gv; ++j) all[i * nv2 + j] = ms[i * gv + j];
                    for (int j = 0; j < more; ++j) all[i * nv2 + gv + j] = ms2[i * more + j];
                }
                if (ms2[bm2 * more + bv2] < ms[bm * gv + bv]) { bm = bm2; bv = gv + bv2; }
                for (int k = 0; k < gm * nv2; ++k) ms[k] = all[k];
            }
            gv += more;
        } else {
            (void)hipStreamSynchronize(ctx->stream);
            for (int k = gv; k < gv + more; ++k) { dev_free(slabs[k]); slabs[k] = nullptr; }
        }
    }
    if (!trial_ok) place_no_preference(ctx, &bm, &bv);
    const hipError_t es = hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < gm; ++k)
        if (k != bm || es != hipSuccess) lfpsqp_mat_free(ctx, cand[k]);
    for (int k = 0; k < gv; ++k)
        if (k != bv || es != hipSuccess) dev_free(slabs[k]);
    if (es != hipSuccess) return set_err(ctx, LFPSQP_ERR_HIP, "hipStreamSynchronize failed while allocating the basis and its vectors: %s", hipGetErrorString(es));
    if (trial_ok) place_record(ctx, gm * gv > 1 ? gm * gv : 0, bm * gv + bv, ms);
    *M_out = cand[bm];
    slab_views(slabs[bv], nvec, cap, count, out);
    return 0;
}

int64_t lfpsqp_vec_len(const lfpsqp_vec* v) { return v ? v->n : -1; }

int lfpsqp_vec_upload(lfpsqp_ctx* ctx, lfpsqp_vec* v, int64_t offset, const double* host, int64_t count) {
    LF_ARG(ctx, ctx && v && host && offset >= 0 && count >= 0 && offset + count <= v->n);
    if (count == 0) return 0;
    LF_HIP(ctx, hipMemcpyAsync(v->p + offset, host, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host buffer is caller-owned and pageable
    return 0;
}

int lfpsqp_vec_download(lfpsqp_ctx* ctx, const lfpsqp_vec* v, int64_t offset, double* host, int64_t count) {
    LF_ARG(ctx, ctx && v && host && offset >= 0 && count >= 0 && offset + count <= v->n);
    if (count == 0) return 0;
    LF_HIP(ctx, hipMemcpyAsync(host, v->p + offset, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// Rows added to the leading dimension of every matrix (env LFPSQP_LD_SKEW, even, 0 .. 2046).  Rounded to whole 2048-row tiles alone, every column
// would start on the same 16 KB phase, and the one-pass kernels -- whose wave instructions read 16 rows of FOUR neighbouring columns -- would
// send the four 128-byte pieces of an instruction to addresses that agree in their low 14 bits.
static int64_t mat_ld_skew() {
    static const int64_t v = [] {
        const char* e = getenv("LFPSQP_LD_SKEW");
        int64_t s = e ? atoll(e) : kLdSkewDefault;
        if (s < 0 || s > 2046) s = kLdSkewDefault;
        return s & ~(int64_t)1;
    }();
    return v;
}

int lfpsqp_mat_alloc(lfpsqp_ctx* ctx, int64_t n, int64_t m, lfpsqp_mat** out) {
    LF_ARG(ctx, ctx != nullptr && out != nullptr && n >= 0 && m >= 0);
    lfpsqp_mat* M = new lfpsqp_mat();
    M->n = n;
    M->m = m;
    M->ld = round_up(n > 0 ? n : 1, kPadRows);
    M->ld += mat_ld_skew();
    const size_t bytes = sizeof(double) * (size_t)M->ld * (size_t)(m > 0 ? m : 1);
    hipError_t e = dev_alloc((void**)&M->p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();      // (as in lfpsqp_vec_alloc)
        delete M;
        return set_err(ctx, LFPSQP_ERR_HIP, "hipMalloc(%lld x %lld matrix) failed: %s", (long long)n, (long long)m, hipGetErrorString(e));
    }
    e = hipMemsetAsync(M->p, 0, bytes, ctx->stream);
    if (e != hipSuccess) { dev_free(M->p); delete M; return set_err(ctx, LFPSQP_ERR_HIP, "hipMemsetAsync failed"); }
    *out = M;
    return 0;
}

int lfpsqp_mat_free(lfpsqp_ctx* ctx, lfpsqp_mat* M) {
    if (!M) return 0;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (M->p && !M->view) dev_free(M->p);      // (a view borrows its storage)
    delete M;
    return 0;
}

int lfpsqp_mat_view(lfpsqp_ctx* ctx, const lfpsqp_mat* A, const lfpsqp_vec* rs, const lfpsqp_vec* u, const lfpsqp_vec* w, lfpsqp_mat** out) {
    LF_ARG(ctx, ctx && A && out && plain_mat(A) && (rs || u) && (!rs || rs->n >= A->n) && ((u == nullptr) == (w == nullptr)) &&
                    (!u || (u->n >= A->n && w->n >= A->m)));
    lfpsqp_mat* V = new (std::nothrow) lfpsqp_mat(*A);
    if (!V) return set_err(ctx, LFPSQP_ERR_HIP, "out of host memory");
    V->rs = rs ? rs->p : nullptr;
    V->ru = u ? u->p : nullptr;
    V->rw = w ? w->p : nullptr;
    V->view = true;
    *out = V;
    return 0;
}

int lfpsqp_mat_rowscaled_view(lfpsqp_ctx* ctx, const lfpsqp_mat* A, const lfpsqp_vec* rs, lfpsqp_mat** out) {
    LF_ARG(ctx, ctx && rs);
    return lfpsqp_mat_view(ctx, A, rs, nullptr, nullptr, out);
}

int lfpsqp_mat_shape(const lfpsqp_mat* M, int64_t* n, int64_t* m) {
    if (!M) return LFPSQP_ERR_ARG;
    if (n) *n = M->n;
    if (m) *m = M->m;
    return 0;
}

int lfpsqp_mat_upload(lfpsqp_ctx* ctx, lfpsqp_mat* M, int64_t col0, int64_t ncols, const double* host, int64_t ldh) {
    LF_ARG(ctx, ctx && plain_mat(M) && host && col0 >= 0 && ncols >= 0 && col0 + ncols <= M->m && ldh >= M->n);
    if (ncols == 0 || M->n == 0) return 0;
    LF_HIP(ctx, hipMemcpy2DAsync(M->p + col0 * M->ld, sizeof(double) * M->ld, host, sizeof(double) * ldh, sizeof(double) * M->n,
                                 (size_t)ncols, hipMemcpyHostToDevice, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int lfpsqp_mat_download(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t col0, int64_t ncols, double* host, int64_t ldh) {
    LF_ARG(ctx, ctx && plain_mat(M) && host && col0 >= 0 && ncols >= 0 && col0 + ncols <= M->m && ldh >= M->n);
    if (ncols == 0 || M->n == 0) return 0;
    LF_HIP(ctx, hipMemcpy2DAsync(host, sizeof(double) * ldh, M->p + col0 * M->ld, sizeof(double) * M->ld, sizeof(double) * M->n,
                                 (size_t)ncols, hipMemcpyDeviceToHost, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int lfpsqp_mat_copy(lfpsqp_ctx* ctx, lfpsqp_mat* dst, const lfpsqp_mat* src) {
    LF_ARG(ctx, ctx && plain_mat(dst) && src && dst->n == src->n && dst->m == src->m && dst->ld == src->ld);
    if (src->view) {                               // the copy of a view is the matrix it stands for: dst = diag(rs) * src + u w'
        LF_ARG(ctx, dst->p != src->p);
        if (src->n == 0 || src->m == 0) return 0;
        hipLaunchKernelGGL(view_copy_kernel, dim3((unsigned)((src->n + 2 * kThreads - 1) / (2 * kThreads)), (unsigned)((src->m + 15) / 16)), dim3(kThreads), 0,
                           ctx->stream, src->p, src->rs, src->ru, src->rw, dst->p, src->ld, src->n, (int)src->m);
        LF_LAUNCH_CHECK(ctx);
        return 0;
    }
    LF_HIP(ctx, hipMemcpyAsync(dst->p, src->p, sizeof(double) * (size_t)src->ld * (size_t)src->m, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

}  // extern "C"
