// Context, buffers, transfers, communicator (RCCL via dlopen or host callback).
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

int allreduce_dev(lfpsqp_ctx* ctx, double* buf, int64_t count, int op) {
    Comm& c = ctx->comm;
    if (c.kind == Comm::NONE || count == 0) return 0;
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

#ifdef LFPSQP_VMM_EXPERIMENT
// EXPERIMENT (tools/gpu_vmm_probe.sh, not in the product build): n-vectors and matrices from the virtual-memory API instead of
// hipMalloc.  LFPSQP_VMM=1: one physical handle per buffer; LFPSQP_VMM=<c> > 1: physical chunks of c MB each, created
// one by one and mapped back to back (LFPSQP_VMM_REV=1: mapped in reverse order of creation).
#include <mutex>
#include <unordered_map>
#include <vector>
namespace {
struct VmmRec { std::vector<hipMemGenericAllocationHandle_t> h; size_t size; };
std::mutex vmm_mu;
std::unordered_map<void*, VmmRec> vmm_map;
int vmm_mode() { static int m = [] { const char* e = getenv("LFPSQP_VMM"); return e ? atoi(e) : 0; }(); return m; }
hipError_t dev_alloc(void** out, size_t bytes) {
    const int mode = vmm_mode();
    if (mode <= 0 || (getenv("LFPSQP_VMM_VEC_ONLY") && bytes > ((size_t)1 << 30))) return hipMalloc(out, bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess) return e;
    static bool said = false;
    if (!said) { said = true; fprintf(stderr, "lfpsqp: VMM allocations, granularity %zu bytes, mode %d\n", gran, mode); }
    const size_t chunk = mode == 1 ? (bytes + gran - 1) / gran * gran : ((size_t)mode << 20);
    const size_t size = (bytes + chunk - 1) / chunk * chunk;
    void* va = nullptr;
    e = hipMemAddressReserve(&va, size, 0, nullptr, 0);
    if (e != hipSuccess) return e;
    VmmRec rec;
    rec.size = size;
    const size_t nch = size / chunk;
    for (size_t c = 0; c < nch; ++c) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, chunk, &prop, 0);
        if (e != hipSuccess) return e;
        rec.h.push_back(h);
    }
    const bool rev = getenv("LFPSQP_VMM_REV") != nullptr;
    for (size_t c = 0; c < nch; ++c) {
        e = hipMemMap((char*)va + c * chunk, chunk, 0, rec.h[rev ? nch - 1 - c : c], 0);
        if (e != hipSuccess) return e;
    }
    hipMemAccessDesc ad{};
    ad.location = prop.location;
    ad.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(va, size, &ad, 1);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(vmm_mu);
    vmm_map[va] = std::move(rec);
    *out = va;
    return hipSuccess;
}
void dev_free(void* p) {
    VmmRec rec;
    {
        std::lock_guard<std::mutex> lk(vmm_mu);
        auto it = vmm_map.find(p);
        if (it == vmm_map.end()) { (void)hipFree(p); return; }
        rec = std::move(it->second);
        vmm_map.erase(it);
    }
    (void)hipMemUnmap(p, rec.size);
    for (auto h : rec.h) (void)hipMemRelease(h);
    (void)hipMemAddressFree(p, rec.size);
}
std::unordered_map<lfpsqp_vec*, double*> skew_base;
}  // namespace
// EXPERIMENT (tools/skew_probe.py): move a vector's start inside its (over-sized) allocation; such a vector is never freed
extern "C" int lfpsqp_x_vec_skew(lfpsqp_vec* v, int64_t skew_doubles, int64_t n) {
    std::lock_guard<std::mutex> lk(vmm_mu);
    auto it = skew_base.find(v);
    if (it == skew_base.end()) it = skew_base.emplace(v, v->p).first;
    v->p = it->second + skew_doubles;
    v->n = n;
    v->cap = round_up(n > 0 ? n : 1, kPadRows);
    return 0;
}
#else
namespace {
inline hipError_t dev_alloc(void** out, size_t bytes) { return hipMalloc(out, bytes); }
inline void dev_free(void* p) { (void)hipFree(p); }
}  // namespace
#endif

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
    ok = ok && hipMalloc((void**)&ctx->istat, 64 * sizeof(int64_t)) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&ctx->h_scal, 4 * 64 * sizeof(double)) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&ctx->h_istat, 4 * 16 * sizeof(int64_t)) == hipSuccess;
    for (int i = 0; ok && i < 4; ++i) ok = hipEventCreate(&ctx->ev_slot[i]) == hipSuccess;
    ok = ok && hipEventCreate(&ctx->ev_t0) == hipSuccess && hipEventCreate(&ctx->ev_t1) == hipSuccess;
    if (ok) ok = hipMemset(ctx->scal, 0, 64 * sizeof(double)) == hipSuccess && hipMemset(ctx->istat, 0, 64 * sizeof(int64_t)) == hipSuccess;
    if (!ok) { delete ctx; return set_err(nullptr, LFPSQP_ERR_HIP, "context resource allocation failed"); }
    if (const char* e = getenv("LFPSQP_ONEPASS")) ctx->tune_onepass = atoi(e);
    if (const char* e = getenv("LFPSQP_SPGRAM")) ctx->tune_spgram = atoi(e) < 0 ? -1 : 0;
    if (const char* e = getenv("LFPSQP_GPING")) ctx->tune_gping = atoi(e) == 1 ? 1 : 0;
    *out = ctx;
    return 0;
}

int lfpsqp_ctx_destroy(lfpsqp_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm.kind == Comm::RCCL && ctx->comm.nccl_comm && ctx->comm.ncclCommDestroy) ctx->comm.ncclCommDestroy(ctx->comm.nccl_comm);
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
    if (e != hipSuccess) { delete v; return set_err(ctx, LFPSQP_ERR_HIP, "hipMalloc(%lld doubles) failed: %s", (long long)n, hipGetErrorString(e)); }
    e = hipMemsetAsync(v->p, 0, sizeof(double) * v->cap, ctx->stream);
    if (e != hipSuccess) { dev_free(v->p); delete v; return set_err(ctx, LFPSQP_ERR_HIP, "hipMemsetAsync failed"); }
    *out = v;
    return 0;
}

int lfpsqp_vec_free(lfpsqp_ctx* ctx, lfpsqp_vec* v) {
    if (!v) return 0;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (v->p) dev_free(v->p);
    delete v;
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
    if (e != hipSuccess) { delete M; return set_err(ctx, LFPSQP_ERR_HIP, "hipMalloc(%lld x %lld matrix) failed: %s", (long long)n, (long long)m, hipGetErrorString(e)); }
    e = hipMemsetAsync(M->p, 0, bytes, ctx->stream);
    if (e != hipSuccess) { dev_free(M->p); delete M; return set_err(ctx, LFPSQP_ERR_HIP, "hipMemsetAsync failed"); }
    *out = M;
    return 0;
}

int lfpsqp_mat_free(lfpsqp_ctx* ctx, lfpsqp_mat* M) {
    if (!M) return 0;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (M->p) dev_free(M->p);
    delete M;
    return 0;
}

int lfpsqp_mat_shape(const lfpsqp_mat* M, int64_t* n, int64_t* m) {
    if (!M) return LFPSQP_ERR_ARG;
    if (n) *n = M->n;
    if (m) *m = M->m;
    return 0;
}

int lfpsqp_mat_upload(lfpsqp_ctx* ctx, lfpsqp_mat* M, int64_t col0, int64_t ncols, const double* host, int64_t ldh) {
    LF_ARG(ctx, ctx && M && host && col0 >= 0 && ncols >= 0 && col0 + ncols <= M->m && ldh >= M->n);
    if (ncols == 0 || M->n == 0) return 0;
    LF_HIP(ctx, hipMemcpy2DAsync(M->p + col0 * M->ld, sizeof(double) * M->ld, host, sizeof(double) * ldh, sizeof(double) * M->n,
                                 (size_t)ncols, hipMemcpyHostToDevice, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int lfpsqp_mat_download(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t col0, int64_t ncols, double* host, int64_t ldh) {
    LF_ARG(ctx, ctx && M && host && col0 >= 0 && ncols >= 0 && col0 + ncols <= M->m && ldh >= M->n);
    if (ncols == 0 || M->n == 0) return 0;
    LF_HIP(ctx, hipMemcpy2DAsync(host, sizeof(double) * ldh, M->p + col0 * M->ld, sizeof(double) * M->ld, sizeof(double) * M->n,
                                 (size_t)ncols, hipMemcpyDeviceToHost, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int lfpsqp_mat_copy(lfpsqp_ctx* ctx, lfpsqp_mat* dst, const lfpsqp_mat* src) {
    LF_ARG(ctx, ctx && dst && src && dst->n == src->n && dst->m == src->m && dst->ld == src->ld);
    LF_HIP(ctx, hipMemcpyAsync(dst->p, src->p, sizeof(double) * (size_t)src->ld * (size_t)src->m, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

}  // extern "C"
