// Host-side internals shared by the translation units of liblfpsqp_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/lfpsqp_hip.h"
#include "kernels.h"

struct lfpsqp_slab {   // one device allocation shared by the vectors of a placement-tuned set (lfpsqp_vecs_alloc_placed)
    void* p = nullptr;
    int refs = 0;
};
struct lfpsqp_vec {
    double* p = nullptr;
    int64_t n = 0;    // logical length (local part of a sharded vector)
    int64_t cap = 0;  // allocated doubles: n rounded up to whole tiles, padding kept finite
    lfpsqp_slab* slab = nullptr;   // non-null: p points into this shared allocation
};

constexpr int64_t kLdSkewDefault = 16;  // rows (128 bytes); see mat_ld_skew (context.hip)

struct lfpsqp_mat {
    double* p = nullptr;
    int64_t n = 0, m = 0;
    int64_t ld = 0;  // rows rounded up to whole tiles; padding rows are zero and never written
    // VIEW (lfpsqp_mat_view): the matrix is diag(rs) * [p] + ru rw'; p belongs to another lfpsqp_mat, rs / ru (n) and rw (m) to lfpsqp_vecs;
    // rs == nullptr: no scaling, ru == rw == nullptr: no rank-one term.  Honoured by the launch helpers run_gemv_t / run_gemv_n / run_gemv_nt /
    // run_onepass and by the Gram / basis-forming products; every other consumer of a matrix refuses a view (plain_mat()).
    const double* rs = nullptr;
    const double* ru = nullptr;
    const double* rw = nullptr;
    bool view = false;
    lfpsqp_mat plain() const { lfpsqp_mat q = *this; q.rs = q.ru = q.rw = nullptr; q.view = false; return q; }
};

namespace lfpsqp {

constexpr int kProfSlots = 8;
constexpr int kProfEvents = 128;  // event pairs kept per slot

struct Comm {
    int rank = 0, nranks = 1;
    enum Kind { NONE, RCCL, CALLBACK, P2P } kind = NONE;
    // P2P: one-shot all-reduce over peer-mapped mailboxes (hipIpc), for the library's latency-bound payloads (context.hip)
    void* p2p_mine = nullptr;            // this rank's mailbox (device memory, exported with hipIpcGetMemHandle)
    void* p2p_peer[16] = {nullptr};      // every rank's mailbox as mapped here (p2p_peer[rank] == p2p_mine)
    uint64_t p2p_seq = 0;                // collectives issued so far (the same on every rank by construction)
    int* p2p_err = nullptr;              // pinned host word: set by a kernel whose wait for a peer timed out
    int p2p_memkind = 0;                 // LFPSQP_P2P_MEM_*: what kind of memory the mailbox is (lfpsqp_comm_p2p_info)
    bool p2p_allow_coarse = false;       // lfpsqp_comm_p2p_allow_coarse
    void* rccl_lib = nullptr;
    void* nccl_comm = nullptr;
    int (*ncclAllReduce)(const void*, void*, size_t, int /*dtype*/, int /*op*/, void*, hipStream_t) = nullptr;
    int (*ncclCommDestroy)(void*) = nullptr;
    const char* (*ncclGetErrorString)(int) = nullptr;
    lfpsqp_allreduce_fn cb = nullptr;
    void* cb_user = nullptr;
};

}  // namespace lfpsqp

struct lfpsqp_ctx {
    int device = 0;
    int num_cu = 0;   // compute units (sizes the persistent grid of onepass_kernel)
    hipStream_t stream = nullptr;
    std::string err;
    std::string devname;
    lfpsqp::Comm comm;
    // an initialised communicator (even with one rank) routes every reduction through it, so the
    // multi-GPU launch sequence can be exercised on a 1-GPU box
    bool comm_active() const { return comm.kind != lfpsqp::Comm::NONE; }

    // reduction workspace: partial sums [rows][ld] (device)
    double* part = nullptr;
    size_t part_cap = 0;  // doubles
    // small replicated device scratch (Gram matrices, m x m factors)
    double* small = nullptr;
    size_t small_cap = 0;
    // replicated m-vector staging: device + pinned host
    double* d_m = nullptr;
    double* h_m = nullptr;
    size_t m_cap = 0;
    // the weights of lfpsqp_elementwise's quadratic term (device, m_lin)
    std::vector<double> warm_V;      // lfpsqp_factorize_hint: eigenvectors of a nearby Gram matrix (m x m, column j = vector j), consumed by the next factorisation
    int warm_m = 0;
    double* d_view = nullptr;        // matrix views: [tau (8) | the raw sums of a second product before the rank-one term is folded in]
    double* d_nvec = nullptr;        // an n-vector of scratch (combined weights of the Gram matrix of a row-scaled view)
    size_t nvec_cap = 0;
    double* d_tri = nullptr;         // lfpsqp_projcg_tridiag: the four n-vectors of its set-up (|off|, sign(off), the two parts of the diagonal weights)
    size_t tri_cap = 0;
    double* d_zeros = nullptr;       // kOnepassMaxCols zeros (the first-product coefficients of a one-pass launch that only evaluates)
    double* d_qw = nullptr;
    size_t qw_cap = 0;
    // small device blocks: solver scalars / status, and their pinned host mirrors
    double* scal = nullptr;    // 64 doubles
    int64_t* istat = nullptr;  // 64 int64
    double* h_scal = nullptr;  // pinned: 4 slots x 64
    int64_t* h_istat = nullptr;  // pinned: 4 slots x 16
    hipEvent_t ev_slot[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;

    // streaming-kernel tuning (lfpsqp_ctx_set_tuning): row pairs per lane (2 or 4), non-temporal matrix loads
    int tune_ks = 0;   // 0 = auto: 4 for >= 4M local rows (measured best at n = 1e7), else 2 (better at the 8-GPU shard size 1.25e6)
    bool tune_nt = true;
    int ks_for(int64_t n) const { return tune_ks ? tune_ks : (n >= 4000000 ? 4 : 2); }
    // one-stream N->T kernels (Newton-retraction step, fused projected-CG iteration): 0 = on, -1 = always the two-pass kernels.
    // Development override: environment variable LFPSQP_ONEPASS, read once at lfpsqp_ctx_create.
    int tune_onepass = 0;
    int tune_spgram = 0;    // lfpsqp_factorize_sp: 0 = Gram matrix from the nonzeros (sp_gram), -1 = on a dense copy (env LFPSQP_SPGRAM=-1; A/B timing)
    int tune_vec_blocks = 0;   // vec_kernel: 0 = one tile per block (4096 blocks at most with reductions); > 0 = at most this many blocks (env LFPSQP_VEC_BLOCKS)
    int tune_nrb_mfma = -1;    // batched Newton step: -1 = the EXACT batch only (default; lfpsqp_ctx_set_nr_batch_mode), 0 = matrix cores for more than
                               // 2 trials, 1 = for every batch (env LFPSQP_NRB_MFMA overrides at context creation)
    int tune_gping = 0;     // fused projected-CG iteration: 0 = the residual updated in place, 1 = two buffers alternating (lfpsqp_ctx_set_residual_buffers)

    bool real_gpu = false;   // gcnArchName "gfx..." (false only in the CPU emulator build of the tests)
    // placement policy (lfpsqp_ctx_set_placement): candidate allocations tried by lfpsqp_mat_alloc_placed / lfpsqp_vecs_alloc_placed
    int place_tries = 3;
    int64_t place_min_bytes = (int64_t)1 << 30;     // matrices below this size are not worth a probe
    double place_last_ms[64] = {0};                  // probe times of the last placed allocation (diagnostics)
    int place_last_n = 0, place_last_pick = 0;

    // state a projcg call that stopped at its iteration limit leaves behind for LFPSQP_PROJCG_RESUME (scalars, t3 and the last sums
    // stay in scal / d_m; anything else that uses d_m invalidates it -- ensure_mvec)
    struct ProjcgResume {
        bool valid = false;
        const double *x = nullptr, *g = nullptr, *d = nullptr, *Z = nullptr;
        int m = 0;
        int64_t nv = 0, iters = 0;
        bool gcur_is_rp = false;     // which of the two alternating residual buffers holds g
        const double *dg = nullptr, *b = nullptr;    // the operator diagonal and right-hand side the solve was started with
        double a0 = 0.0;
        int64_t n_global = 0;
        uint64_t epoch = 0;          // launch_epoch when the call returned: ANY kernel the library queued since then voids the state
    } pcg_resume;
    int batch_wg_cap = 0;            // test hook (env LFPSQP_NRB_WG_CAP at context creation): workgroups per CU of the exact batch's launch (then several virtual spans each)
    int stage_cap = 0;               // test hook (env LFPSQP_STAGE_ROUNDS at context creation): cap on the rounds per burst of staged stores
    uint64_t launch_epoch = 0;       // bumped by every launch helper (run_vec / run_gemv_* / run_onepass / launch_reduce)

    // optional per-kernel-family profiling with HIP events on `stream`
    bool profiling = false;
    hipEvent_t prof_ev[lfpsqp::kProfSlots][lfpsqp::kProfEvents][2];
    int prof_used[lfpsqp::kProfSlots] = {0};
    int64_t prof_seq[lfpsqp::kProfSlots] = {0};     // launches seen per slot (every 4th is timed)
    bool prof_live[lfpsqp::kProfSlots] = {false};
    int64_t prof_count[lfpsqp::kProfSlots] = {0};
    double prof_ms[lfpsqp::kProfSlots] = {0};
    bool prof_init = false;
};

namespace lfpsqp {

int set_err(lfpsqp_ctx* ctx, int code, const char* fmt, ...);
int ensure_part(lfpsqp_ctx* ctx, size_t doubles);
int ensure_small(lfpsqp_ctx* ctx, size_t doubles);
int ensure_mvec(lfpsqp_ctx* ctx, size_t doubles);   // d_m / h_m, each `doubles` long
int ensure_nvec(lfpsqp_ctx* ctx, size_t doubles);   // d_nvec (padded to whole tiles)
int ensure_view(lfpsqp_ctx* ctx);                   // d_view (kViewScratch doubles)
constexpr int kViewTau = 8;                          // doubles reserved for tau at the head of d_view
constexpr int kViewScratch = kViewTau + 4 * 1024 + 64;
int allreduce_dev(lfpsqp_ctx* ctx, double* buf, int64_t count, int op = 0);  // in place, stream ordered; op 0 sum / 1 max; no-op for 1 rank

// profiling helpers: bracket one launch of slot `s`
void prof_begin(lfpsqp_ctx* ctx, int s);
void prof_end(lfpsqp_ctx* ctx, int s);
void prof_collect(lfpsqp_ctx* ctx);  // after a stream sync: fold event pairs into prof_ms

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
inline int64_t ntiles_of(int64_t n, int ks) { return (n + (int64_t)kSlabRows * ks - 1) / ((int64_t)kSlabRows * ks); }

#define LF_HIP(ctx, expr)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return lfpsqp::set_err((ctx), LFPSQP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
                                   hipGetErrorString(e__), __FILE__, __LINE__);                        \
    } while (0)

#define LF_TRY(expr)               \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != 0) return rc__; \
    } while (0)

#define LF_ARG(ctx, cond)                                                                               \
    do {                                                                                                \
        if (!(cond)) return lfpsqp::set_err((ctx), LFPSQP_ERR_ARG, "invalid argument: %s (%s:%d)", #cond, \
                                            __FILE__, __LINE__);                                        \
    } while (0)

#define LF_LAUNCH_CHECK(ctx) LF_HIP(ctx, hipGetLastError())

// ---- generic launchers (templates, so they live in the header) -------------

// second stage of a reduction: out[0:ncols] = reduce over `nrows` partial rows of ctx->part.
// Many rows x many columns (the m-vector of a GEMV-T) go through two launches so that more than
// ceil(ncols/32) workgroups share the work; the caller reserves kReduceScratchRows extra rows of
// `part_ld` doubles behind the partials for the intermediate.
constexpr int kReduceRowBlocks = 16;
inline size_t reduce_scratch(int part_ld) { return (size_t)kReduceRowBlocks * part_ld; }
template <class POST>
int launch_reduce(lfpsqp_ctx* ctx, int64_t nrows, int ncols, int part_ld, unsigned ismax, double* out, POST post, int ncols_like = 0) {
    ++ctx->launch_epoch;
    // (ncols_like > 0: the shape of the reduction -- columns per block, hence row groups per column, and one stage or two -- is chosen as for
    // that many columns, so that every column is summed in the order a launch over ncols_like columns uses: the exact batch of retract.hip)
    const int shape_cols = ncols_like > 0 ? ncols_like : ncols;
    int cw_log2 = 0;
    while ((1 << cw_log2) < shape_cols && cw_log2 < 5) ++cw_log2;
    const int cw = 1 << cw_log2;
    const int gx = (ncols + cw - 1) / cw;
    if (shape_cols >= 32 && nrows >= 2048) {
        double* mid = ctx->part + (size_t)nrows * part_ld;
        const int64_t chunk = (nrows + kReduceRowBlocks - 1) / kReduceRowBlocks;
        hipLaunchKernelGGL((reduce_rows_kernel<NoPost>), dim3(gx, kReduceRowBlocks), dim3(1024), 0, ctx->stream, ctx->part, nrows, ncols,
                           part_ld, ismax, mid, part_ld, chunk, cw_log2, NoPost());
        LF_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL((reduce_rows_kernel<POST>), dim3(gx, 1), dim3(1024), 0, ctx->stream, mid, (int64_t)kReduceRowBlocks, ncols,
                           part_ld, ismax, out, 0, (int64_t)kReduceRowBlocks, cw_log2, post);
        LF_LAUNCH_CHECK(ctx);
        return 0;
    }
    hipLaunchKernelGGL((reduce_rows_kernel<POST>), dim3(gx, 1), dim3(1024), 0, ctx->stream, ctx->part, nrows, ncols, part_ld, ismax, out,
                       0, nrows, cw_log2, post);
    LF_LAUNCH_CHECK(ctx);
    return 0;
}

// t_out[0:ncols] = M[:, :ncols]' * vp  (global: all-reduced over ranks).  The producer runs
// (with its fused stores) even when ncols == 0, so solvers need no special case for an
// empty basis (reference: projcg! with an n x 0 U).
template <class VP>
int run_gemv_t(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int ncols, int64_t n, VP vp, double* t_out, int prof_slot = -1) {
    if constexpr (!is_rowscaled<VP>::value) {
        if (M && M->view) {                        // a view: the same kernel over the plain storage, producer wrapped (kernels.h)
            const lfpsqp_mat plain = M->plain();
            const RsLoadV<VP> wv{vp, ViewD{M->rs, M->ru, nullptr}};
            if (!M->ru) return run_gemv_t(ctx, &plain, ncols, n, wv, t_out, prof_slot);
            if (ncols + 1 > kViewScratch - kViewTau) return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "matrix view with a rank-one term: too many columns");
            LF_TRY(ensure_view(ctx));
            double* raw = ctx->d_view + kViewTau;  // [M'(rs .* v) (ncols) ; u'v]
            LF_TRY(run_gemv_t(ctx, &plain, ncols, n, wv, raw, prof_slot));
            hipLaunchKernelGGL((view_fold_kernel<0>), dim3(1), dim3(256), 0, ctx->stream, raw, t_out, M->rw, 1, ncols, 0);
            LF_LAUNCH_CHECK(ctx);
            return 0;
        }
    }
    ++ctx->launch_epoch;
    const int ks = ctx->ks_for(n);
    const int64_t tiles = ntiles_of(n, ks);
    int xcol = 0;                                  // a view's rank-one term: one more (virtual) column, sum_i u_i v_i
    if constexpr (is_rowscaled<VP>::value) xcol = vp.vw.u ? 1 : 0;
    const int part_ld = (int)round_up(ncols + xcol > 0 ? ncols + xcol : 1, 32);
    if (tiles > 0) {
        LF_TRY(ensure_part(ctx, (size_t)tiles * part_ld + reduce_scratch(part_ld)));
        if (prof_slot >= 0) prof_begin(ctx, prof_slot);
        const double* Mp = M ? M->p : nullptr;
        const int64_t ld = M ? M->ld : 0;
#define LF_GT(KS, NT)                                                                                                        \
    hipLaunchKernelGGL((gemv_t_kernel<VP, KS, NT>), dim3((unsigned)tiles), dim3(kThreads), 0, ctx->stream, Mp, ld, ncols, n, vp, \
                       ctx->part, part_ld)
        if (ks == 4) { if (ctx->tune_nt) LF_GT(4, true); else LF_GT(4, false); }
        else         { if (ctx->tune_nt) LF_GT(2, true); else LF_GT(2, false); }
#undef LF_GT
        if (prof_slot >= 0) prof_end(ctx, prof_slot);
        LF_LAUNCH_CHECK(ctx);
    }
    if (ncols + xcol == 0) return 0;
    if (tiles > 0) LF_TRY(launch_reduce(ctx, tiles, ncols + xcol, part_ld, 0u, t_out, NoPost()));
    else LF_HIP(ctx, hipMemsetAsync(t_out, 0, sizeof(double) * (ncols + xcol), ctx->stream));
    return allreduce_dev(ctx, t_out, ncols + xcol);
}

// fused y-side consumer of M[:, :ncols] * t; reductions (NRED sums) land in red_out (global)
// when POST is given it runs after the (all-reduced) sums are final.
template <class EP, int NRED, class POST>
int run_gemv_n(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int ncols, int64_t n, const double* t, EP ep, double* red_out, POST post,
               int prof_slot = -1) {
    if constexpr (!is_rowscaled<EP>::value) {
        if (M && M->view) {
            const lfpsqp_mat plain = M->plain();
            if (M->ru) {                           // tau = w't ahead of the launch
                LF_TRY(ensure_view(ctx));
                hipLaunchKernelGGL((view_tau_kernel<0>), dim3(1), dim3(256), 0, ctx->stream, M->rw, t, ncols, ctx->d_view);
                LF_LAUNCH_CHECK(ctx);
            }
            return run_gemv_n<RsApplyE<EP>, NRED, POST>(ctx, &plain, ncols, n, t, RsApplyE<EP>{ep, ViewD{M->rs, M->ru, ctx->d_view}}, red_out, post, prof_slot);
        }
    }
    ++ctx->launch_epoch;
    const int ks = ctx->ks_for(n);
    const int64_t tiles = ntiles_of(n, ks);
    if (tiles > 0) {
        LF_TRY(ensure_part(ctx, (size_t)tiles * kMaxRed));
        if (prof_slot >= 0) prof_begin(ctx, prof_slot);
        const double* Mp = M ? M->p : nullptr;
        const int64_t ld = M ? M->ld : 0;
#define LF_GN(KS, NT)                                                                                                          \
    hipLaunchKernelGGL((gemv_n_kernel<EP, NRED, KS, NT>), dim3((unsigned)tiles), dim3(kThreads), 0, ctx->stream, Mp, ld, ncols, n, \
                       t, ep, ctx->part)
        if (ks == 4) { if (ctx->tune_nt) LF_GN(4, true); else LF_GN(4, false); }
        else         { if (ctx->tune_nt) LF_GN(2, true); else LF_GN(2, false); }
#undef LF_GN
        if (prof_slot >= 0) prof_end(ctx, prof_slot);
        LF_LAUNCH_CHECK(ctx);
    }
    if (NRED > 0) {
        if (tiles == 0) LF_HIP(ctx, hipMemsetAsync(red_out, 0, sizeof(double) * NRED, ctx->stream));
        if (!ctx->comm_active()) {
            if (tiles > 0) LF_TRY(launch_reduce(ctx, tiles, NRED, kMaxRed, 0u, red_out, post));
            else {
                hipLaunchKernelGGL((post_kernel<POST>), dim3(1), dim3(1), 0, ctx->stream, red_out, post);
                LF_LAUNCH_CHECK(ctx);
            }
        } else {
            if (tiles > 0) LF_TRY(launch_reduce(ctx, tiles, NRED, kMaxRed, 0u, red_out, NoPost()));
            LF_TRY(allreduce_dev(ctx, red_out, NRED));
            hipLaunchKernelGGL((post_kernel<POST>), dim3(1), dim3(1), 0, ctx->stream, red_out, post);
            LF_LAUNCH_CHECK(ctx);
        }
    }
    return 0;
}

// out[0 : n2 + NRED) = [ M2[:, :n2]' v ; reduction terms ], v produced by EP from M1[:, :n1] * t (all-reduced)
template <class EP, int NRED>
int run_gemv_nt(lfpsqp_ctx* ctx, const lfpsqp_mat* M1, int n1, const double* t, const lfpsqp_mat* M2, int n2, int64_t n, EP ep,
                double* out) {
    if constexpr (!is_rowscaled<EP>::value) {
        const bool w1 = M1 && M1->view, w2 = M2 && M2->view;
        if (w1 || w2) {
            lfpsqp_mat p1, p2;
            if (M1) p1 = M1->plain();
            if (M2) p2 = M2->plain();
            if (n2 + NRED + 1 > kViewScratch - kViewTau) return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "matrix view: too many columns");
            LF_TRY(ensure_view(ctx));
            if (w1 && M1->ru) {
                hipLaunchKernelGGL((view_tau_kernel<0>), dim3(1), dim3(256), 0, ctx->stream, M1->rw, t, n1, ctx->d_view);
                LF_LAUNCH_CHECK(ctx);
            }
            const RsStepE<EP, NRED> we{ep, w1 ? ViewD{M1->rs, M1->ru, ctx->d_view} : ViewD{nullptr, nullptr, nullptr},
                                       w2 ? ViewD{M2->rs, M2->ru, nullptr} : ViewD{nullptr, nullptr, nullptr}, w1, w2};
            double* raw = ctx->d_view + kViewTau;  // [M2'(rs2 .* v) (n2) ; the functor's sums (NRED) ; u2'v]
            LF_TRY((run_gemv_nt<RsStepE<EP, NRED>, NRED + 1>(ctx, M1 ? &p1 : nullptr, n1, t, M2 ? &p2 : nullptr, n2, n, we, raw)));
            hipLaunchKernelGGL((view_fold_kernel<0>), dim3(1), dim3(256), 0, ctx->stream, raw, out, (w2 && M2->ru) ? M2->rw : nullptr, 1, n2, NRED);
            LF_LAUNCH_CHECK(ctx);
            return 0;
        }
    }
    ++ctx->launch_epoch;
    const int ks = ctx->ks_for(n);
    const int64_t tiles = ntiles_of(n, ks);
    const int nout = n2 + NRED;
    const int part_ld = (int)round_up(nout > 0 ? nout : 1, 32);
    if (tiles > 0) {
        LF_TRY(ensure_part(ctx, (size_t)tiles * part_ld + reduce_scratch(part_ld)));
        const double* p1 = M1 ? M1->p : nullptr;
        const double* p2 = M2 ? M2->p : nullptr;
#define LF_GNT(KS, NT)                                                                                                           \
    hipLaunchKernelGGL((gemv_nt_kernel<EP, NRED, KS, NT>), dim3((unsigned)tiles), dim3(kThreads), 0, ctx->stream, p1, M1 ? M1->ld : 0, \
                       n1, t, p2, M2 ? M2->ld : 0, n2, n, ep, ctx->part, part_ld)
        if (ks == 4) { if (ctx->tune_nt) LF_GNT(4, true); else LF_GNT(4, false); }
        else         { if (ctx->tune_nt) LF_GNT(2, true); else LF_GNT(2, false); }
#undef LF_GNT
        LF_LAUNCH_CHECK(ctx);
    }
    if (nout == 0) return 0;
    if (tiles > 0) LF_TRY(launch_reduce(ctx, tiles, nout, part_ld, 0u, out, NoPost()));
    else LF_HIP(ctx, hipMemsetAsync(out, 0, sizeof(double) * nout, ctx->stream));
    return allreduce_dev(ctx, out, nout);
}

// a matrix argument of a routine that reads or writes the storage directly: not a row-scaled view
inline bool plain_mat(const lfpsqp_mat* M) { return M && !M->view; }

// One-stream N->T product over M (onepass_kernel): usable for this shape?  Returns the lane-group count CW (4) or 0.
inline int onepass_cw(const lfpsqp_ctx* ctx, int ncN, int64_t ld, int64_t n) {
    const int cw = 4;
    if (ctx->tune_onepass < 0 || ncN < cw || ncN > kOnepassMaxCols) return 0;
    // 32-bit lane offsets: column-within-group stride and the row byte offset must fit
    if ((int64_t)(cw - 1) * ld * 8 + (int64_t)kPadRows * 8 >= ((int64_t)1 << 32) || (n + kPadRows) * 8 >= ((int64_t)1 << 32)) return 0;
    return cw;
}

// out[k*ncT + j] = sum_rows M[row, j] * v_k[row] (k < NV, j < ncT), out[NV*ncT + r] = reduction r, with v produced by EP
// from M[row, :ncN] . t -- all-reduced over ranks; one pass over M.  Persistent grid: as many workgroups as the device
// keeps resident for this instantiation (occupancy query, cached), capped by the number of 64-row rounds.
// Staged stores (onepass_kernel STG) of functor EP at CPL column groups per wave, for functors that offer a staged form (EP::kStageStreams):
// rounds per burst and the waves per SIMD the instantiation is compiled for.  The CU's 160 KB of LDS hold 320 rounds of the narrow form
// (512 bytes each) however they are split: three workgroups of 102 rounds (3 waves per SIMD: tiles of up to 24 column groups) or two
// of 153 (2 waves per SIMD, the register budget the larger tiles need with their running sums in registers).  A round of the wide
// form is 16 rows (128 bytes): 256 rounds (160 at three workgroups per CU) fit beside its LDS running sums.
#ifndef LFPSQP_OP_STAGE
#define LFPSQP_OP_STAGE 1
#endif
#ifndef LFPSQP_OP_STAGE_MINCPL
#define LFPSQP_OP_STAGE_MINCPL 8
#endif
#ifndef LFPSQP_OP_STAGE_WIDE
#define LFPSQP_OP_STAGE_WIDE 1
#endif
struct StageCfg { int rounds, waves; };
template <class EP>
constexpr StageCfg onepass_stage(int cpl, bool wide, int na) {
    // (tiles of more than 33 column groups per wave run at one wave per SIMD with part of the tile in accumulation registers: no staging)
    constexpr int ns = stage_streams<EP>::value;      // two staged vectors (stacked forms): half the rounds per burst
    if (!LFPSQP_OP_STAGE || ns == 0 || (na != 1 && !stage_any_na<EP>::value) || cpl < LFPSQP_OP_STAGE_MINCPL || cpl > 33) return StageCfg{0, 1};
    // several first products (the tangent step: 2-3 coefficient vectors in LDS, 3-4 staged vectors): two workgroups per CU, what is left of
    // their 80 KB after the coefficients (na x 1 KB at 33 column groups) in rounds of ns x 512 bytes
    if (na != 1) return wide ? StageCfg{0, 1} : StageCfg{(150 - 3 * na) / ns, 2};
    if (wide) return !LFPSQP_OP_STAGE_WIDE ? StageCfg{0, 1} : (cpl <= 24 ? StageCfg{160 / ns, 3} : StageCfg{256 / ns, 2});
    // (the stacked forms at 17..24 column groups need a few registers more than three waves per SIMD leave them)
    return (cpl <= 16 || (cpl <= 24 && ns == 1)) ? StageCfg{102 / ns, 3} : StageCfg{153 / ns, 2};
}
template <class EP, int NV, int NRED, int CPL, bool EXACT, bool WIDE, int NA, bool LACC, int STG = 0, int SW = 1>
inline int onepass_grid(lfpsqp_ctx* ctx, int64_t rounds, int wg_per_cu_cap = 0) {
    static int per_cu = 0;                        // one per kernel instantiation
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, onepass_kernel<EP, NV, NRED, CPL, EXACT, WIDE, NA, LACC, STG, SW>, kThreads, 0) != hipSuccess ||
            nb < 1)
            nb = 1;
        per_cu = nb;
    }
    // (wg_per_cu_cap: launches of DIFFERENT instantiations that must sum their partials in the same order ask for the same grid)
    const int pc = (wg_per_cu_cap > 0 && wg_per_cu_cap < per_cu) ? wg_per_cu_cap : per_cu;
    int64_t g = (int64_t)pc * (ctx->num_cu > 0 ? ctx->num_cu : 1);
    if (g > rounds) g = rounds;
    return (int)(g < 1 ? 1 : g);
}

// vspans > 0 (batched first products only, NA > 1): cut the rows into that many VIRTUAL spans -- the grid of another instantiation, queried with
// grid_out -- and emit one partial row per span (onepass_kernel, "virtual spans"): the sums are then formed in that instantiation's order.
// grid_out != nullptr: no launch; *grid_out = the grid this call would use (= its number of partial rows).
template <class EP, int NV, int NRED, int NA = 1>
int run_onepass(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int ncN, int ncT, int64_t n, const double* t, const EP& ep, double* out,
                int prof_slot = -1, int t_stride = 0, int wg_per_cu_cap = 0, bool discard_sums = false, int vspans = 0, int* grid_out = nullptr,
                int reduce_like = 0) {
    if constexpr (!is_rowscaled<EP>::value) {
        if (M->view) {                             // a view: the same kernel over the plain storage, row functor wrapped (kernels.h)
            if constexpr (no_rowscale<EP>::value || NA != 1) {
                return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "this kernel does not take a matrix view");
            } else {
                const lfpsqp_mat plain = M->plain();
                const int nraw = NV * ncT + NRED + NV;
                if (nraw > kViewScratch - kViewTau) return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "matrix view: too many columns");
                LF_TRY(ensure_view(ctx));
                if (M->ru) {
                    hipLaunchKernelGGL((view_tau_kernel<0>), dim3(1), dim3(256), 0, ctx->stream, M->rw, t, ncN, ctx->d_view);
                    LF_LAUNCH_CHECK(ctx);
                }
                using WE = RsRowE<EP, NV, NRED>;
                double* raw = ctx->d_view + kViewTau;   // [second products (NV x ncT) ; the functor's sums (NRED) ; u'v_q (NV)]
                if (grid_out)
                    return run_onepass<WE, NV, NRED + NV, 1>(ctx, &plain, ncN, ncT, n, t, WE{ep, ViewD{M->rs, M->ru, ctx->d_view}}, raw, prof_slot, t_stride,
                                                             wg_per_cu_cap, discard_sums, 0, grid_out);
                LF_TRY((run_onepass<WE, NV, NRED + NV, 1>(ctx, &plain, ncN, ncT, n, t, WE{ep, ViewD{M->rs, M->ru, ctx->d_view}}, raw, prof_slot, t_stride,
                                                          wg_per_cu_cap, discard_sums)));
                if (discard_sums) return 0;
                hipLaunchKernelGGL((view_fold_kernel<0>), dim3(1), dim3(256), 0, ctx->stream, raw, out, M->ru ? M->rw : nullptr, NV, ncT, NRED);
                LF_LAUNCH_CHECK(ctx);
                return 0;
            }
        }
    }
    if (!grid_out) ++ctx->launch_epoch;
    const int cpl = (ncN + 3) / 4;                // column groups
    const bool wide = cpl > 64;                   // more than 256 columns: the four waves of a workgroup split the columns
    const int round_rows = wide ? 16 : kOnepassRound;
    const int64_t rounds = (n + round_rows - 1) / round_rows;
    const int nout = NV * ncT + NRED;
    const int part_ld = (int)round_up(nout, 32);
    if (grid_out) *grid_out = 0;
    if (rounds > 0) {
        int grid = 0, nrows = 0, vs = 0;
// LACC (running sums of the second product in LDS): where the registers it frees buy a wave per SIMD and the LDS it takes
// (NV * ceil(CPL/4) * 2 KB per workgroup) still leaves room for those workgroups -- the single-vector / two-vector kernels
// with 17..33 column groups per wave (m = 65..132, and 260..528 in the wide form).
#define LF_OP(CPL, EXACT, WIDE)                                                                                                      \
    do {                                                                                                                             \
        constexpr StageCfg kC = onepass_stage<EP>(CPL, WIDE, NA);                                                                    \
        constexpr int kG = kC.rounds, kW0 = kC.waves;                                                                                \
        /* batched first products (NA > 1, functors that ask for it -- EP::kLaccAnyNA): the NV x ceil(CPL/4) running sums of a lane in LDS */ \
        /* (up to 72 KB: two workgroups per CU) instead of twice as many registers -- two waves per SIMD instead of one */                  \
        constexpr bool kLB = NA > 1 && lacc_any_na<EP>::value && !(WIDE) && kOpLacc && (CPL) > 16 && (CPL) <= 33 &&                        \
                             NV * (((CPL) + 3) / 4) * kThreads * 8 <= 72 * 1024;                                                             \
        constexpr bool kL = kLB || ((kG == 0 || (WIDE)) && kOpLacc && NA == 1 && (CPL) > 16 && (CPL) <= 33);                                \
        constexpr int kW = kLB ? 2 : kW0;                                                                                                    \
        grid = onepass_grid<EP, NV, NRED, CPL, EXACT, WIDE, NA, kL, kG, kW>(ctx, rounds, wg_per_cu_cap);                \
        if (grid_out) { *grid_out = grid; return 0; }                                                                                \
        nrows = grid;                                                                                                                \
        if (NA > 1 && vspans > 0) {                                                                                                  \
            vs = (int)((int64_t)vspans > rounds ? rounds : (int64_t)vspans);                                                         \
            const int per = (vs + grid - 1) / grid;                                                                                  \
            grid = (vs + per - 1) / per;                                                                                             \
            nrows = vs;                                                                                                              \
        }                                                                                                                            \
        LF_TRY(ensure_part(ctx, (size_t)nrows * part_ld + reduce_scratch(part_ld)));                                                 \
        if (prof_slot >= 0) prof_begin(ctx, prof_slot);                                                                              \
        hipLaunchKernelGGL((onepass_kernel<EP, NV, NRED, CPL, EXACT, WIDE, NA, kL, kG, kW>), dim3((unsigned)grid), dim3(kThreads), \
                           0, ctx->stream, M->p, M->ld, ncN, ncT, n, rounds, t, t_stride, ep, ctx->part, part_ld, ctx->stage_cap, vs); \
    } while (0)
        if (wide) {
            const int cplw = (cpl + kWaves - 1) / kWaves;     // column groups per wave
            if (cplw <= 24) LF_OP(24, false, true);
            else if (cplw <= 32) LF_OP(32, false, true);
            else if (cplw <= 48) LF_OP(48, false, true);
            else LF_OP(64, false, true);
        } else if (cpl <= 4) LF_OP(4, false, false);
        else if (cpl <= 8) LF_OP(8, false, false);
        else if (cpl <= 16) LF_OP(16, false, false);
        else if (cpl <= 24) LF_OP(24, false, false);
        else if (cpl == 32) LF_OP(32, true, false);
        else if (cpl == 33) LF_OP(33, true, false);
        else if (cpl <= 33) LF_OP(33, false, false);
        else if (cpl <= 48) LF_OP(48, false, false);
        else LF_OP(64, false, false);
#undef LF_OP
        if (prof_slot >= 0) prof_end(ctx, prof_slot);
        LF_LAUNCH_CHECK(ctx);
        if (discard_sums) return 0;          // (a launch used for its row update alone, e.g. GEMV-N: no second stage, no collective)
        LF_TRY(launch_reduce(ctx, nrows, nout, part_ld, 0u, out, NoPost(), reduce_like));
    } else {
        if (grid_out || discard_sums) return 0;
        LF_HIP(ctx, hipMemsetAsync(out, 0, sizeof(double) * nout, ctx->stream));
    }
    return allreduce_dev(ctx, out, nout);
}

// grid of vec_kernel: one 512-row tile per block without reductions; with reductions at most kVecRedBlocks blocks of consecutive tiles
// (their partial rows are read back by the second stage).  ctx->tune_vec_blocks > 0 caps the grid of BOTH kinds (env LFPSQP_VEC_BLOCKS:
// A/B against the earlier persistent grid of 2048 blocks).
constexpr int kVecRedBlocks = 4096;
inline void vec_grid(const lfpsqp_ctx* ctx, int64_t n, bool reduces, int* grid, int* tpb) {
    int64_t t = (n + kSlabRows - 1) / kSlabRows;
    if (t < 1) t = 1;
    int64_t cap = reduces ? kVecRedBlocks : t;
    if (ctx->tune_vec_blocks > 0 && ctx->tune_vec_blocks < cap) cap = ctx->tune_vec_blocks;
    const int64_t per = (t + cap - 1) / cap;
    *tpb = (int)per;
    *grid = (int)((t + per - 1) / per);
}

// elementwise map with NRED reductions (sum, or max where ismax bit set) -> red_out (global)
template <class F, int NRED, class POST>
int run_vec(lfpsqp_ctx* ctx, int64_t n, F f, unsigned ismax, double* red_out, POST post, int prof_slot = -1) {
    ++ctx->launch_epoch;
    int grid = 1, tpb = 1;
    vec_grid(ctx, n, NRED > 0, &grid, &tpb);
    if (NRED > 0) LF_TRY(ensure_part(ctx, (size_t)grid * kMaxRed));
    if (prof_slot >= 0) prof_begin(ctx, prof_slot);
    hipLaunchKernelGGL((vec_kernel<F, NRED>), dim3(grid), dim3(kThreads), 0, ctx->stream, f, n, ismax, ctx->part, tpb);
    if (prof_slot >= 0) prof_end(ctx, prof_slot);
    LF_LAUNCH_CHECK(ctx);
    if (NRED > 0) {
        if (!ctx->comm_active()) {
            LF_TRY(launch_reduce(ctx, grid, NRED, kMaxRed, ismax, red_out, post));
        } else {
            const unsigned all = (1u << NRED) - 1u;
            if (ismax != 0 && ismax != all) return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "mixed sum/max reduction across ranks");
            LF_TRY(launch_reduce(ctx, grid, NRED, kMaxRed, ismax, red_out, NoPost()));
            LF_TRY(allreduce_dev(ctx, red_out, NRED, ismax ? 1 : 0));
            hipLaunchKernelGGL((post_kernel<POST>), dim3(1), dim3(1), 0, ctx->stream, red_out, post);
            LF_LAUNCH_CHECK(ctx);
        }
    }
    return 0;
}

// G = R' diag(w) R for R_i = M_i + sgn_i M_{i+1} on the matrix cores (factorize.hip, gram_kernel SHIFT); plain M, w >= 0, sgn = +-1 (device)
int gram_shifted(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int ncols, const double* w, const double* sgn, std::vector<double>& G);

// one-sided Jacobi on the columns of a small host matrix, run on the device (jacobi.hip); false = shape not covered
bool device_jacobi(lfpsqp_ctx* ctx, int rows_dot, int rows_all, int cols, std::vector<double>& X, int* sweeps_out = nullptr);

// average time (ms) of `reps` launches of the fused projected-CG kernel over M[:, :ncols] with (g, d, a) as its residual (loaded and stored in
// place), direction and operator-diagonal vectors -- all three must be ZERO-filled (the kernel then leaves them zero); *ms < 0: shape without a
// one-pass kernel.  The placement probe of lfpsqp_mat_alloc_placed / lfpsqp_vecs_alloc_placed (projcg.hip).
int placement_probe(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int ncols, double* g, double* d, double* a, int reps, double* ms);

// roctx range around a solver call (SURVEY 5 "tracing"): rocprofv3 --marker-trace / the kernel trace then shows which lfpsqp_* call a kernel
// belongs to.  libroctx64 is loaded on first use (dlopen, like RCCL); without it, or with LFPSQP_ROCTX=0, the ranges are no-ops.
struct TraceRange {
    explicit TraceRange(const char* name);
    ~TraceRange();
    bool on;
};
#define LF_RANGE(name) lfpsqp::TraceRange lf_range__(name)

// read `count` doubles of device memory back after everything queued so far
int read_back(lfpsqp_ctx* ctx, const double* dev, double* host, int64_t count);

}  // namespace lfpsqp
