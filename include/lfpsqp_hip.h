/*
 * lfpsqp_hip.h -- C ABI of liblfpsqp_hip.so, the MI355X (gfx950) implementation
 * of the LFPSQP inner-loop hot path (projected CG, retractions, the tall-skinny
 * fp64 linear algebra under them).
 *
 * This is the drop-in boundary: every entry point below names
 * the reference call site it replaces (paths into ksil/LFPSQP.jl).  The
 * reference reaches its native code through two Fortran-ABI ccalls
 * (src/la_helper.jl:22 dgesvd_, :37 dgemv_) plus LinearAlgebra.BLAS/mul!; a
 * Julia maintainer binds the functions here with `ccall((:name, liblfpsqp_hip),
 * Cint, (...), ...)` -- see INTEGRATION.md and julia/LFPSQPHip.jl.
 *
 * Conventions
 *   - plain C: opaque handles, raw pointers, sizes as int64_t; no exceptions.
 *   - every function returns an int status: 0 = ok, < 0 = error
 *     (lfpsqp_last_error(ctx) gives the text).  Numerical outcomes (negative
 *     curvature, retraction flags, ...) are OUTPUT values, never errors, exactly
 *     as in the reference (SURVEY §8b "error conventions").
 *   - one context per process and GPU.  Multi-GPU = one process per GPU; each
 *     context owns the rows [row0, row0+n_loc) of every n-vector and of the
 *     n x m matrices (lfpsqp_shard_range), m-sized data is replicated, and the
 *     library all-reduces the m-vector / CG scalars itself over RCCL
 *     (lfpsqp_comm_init_rccl) -- SURVEY §8e.
 *   - host arrays stay caller-owned; device memory is library-owned.
 *   - everything is fp64, matrices are column-major with unit row stride
 *     (Julia's layout), 0-based sizes/counters.
 */
#ifndef LFPSQP_HIP_H
#define LFPSQP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFPSQP_OK 0
#define LFPSQP_ERR_ARG (-1)
#define LFPSQP_ERR_HIP (-2)
#define LFPSQP_ERR_COMM (-3)
#define LFPSQP_ERR_NUMERIC (-4)
#define LFPSQP_ERR_UNSUPPORTED (-5)

typedef struct lfpsqp_ctx lfpsqp_ctx; /* one GPU + stream + workspaces + communicator */
typedef struct lfpsqp_vec lfpsqp_vec; /* device fp64 vector (sharded n-vector or replicated m-vector) */
typedef struct lfpsqp_mat lfpsqp_mat; /* device fp64 column-major n_loc x m matrix */
typedef struct lfpsqp_spmat lfpsqp_spmat; /* device fp64 sparse n_loc x m matrix with a few nonzeros per row */

/* ---- context ------------------------------------------------------------ */
int lfpsqp_ctx_create(int device, lfpsqp_ctx** out);
int lfpsqp_ctx_destroy(lfpsqp_ctx* ctx);
int lfpsqp_ctx_sync(lfpsqp_ctx* ctx); /* wait for the context's stream */
const char* lfpsqp_last_error(const lfpsqp_ctx* ctx);
/* name of the device the context runs on ("cpu-emulator" only in the test build) */
int lfpsqp_device_name(const lfpsqp_ctx* ctx, char* buf, int64_t buflen);
/* "GPU-<16 hex digits>" (as rocminfo prints it) of the device this context computes on; buflen >= 40 */
int lfpsqp_device_uuid(const lfpsqp_ctx* ctx, char* buf, int64_t buflen);
/* streaming-kernel variant: ks = 16-byte row pairs per lane (2 or 4; tile = 512*ks rows; 0 = auto: 4 for
 * >= 4M local rows, else 2), nt != 0 = non-temporal loads of the matrix stream.  Results are bit-identical
 * across nt, and differ only in summation order across ks.  Default (auto, 1), FINDINGS.md §5. */
int lfpsqp_ctx_set_tuning(lfpsqp_ctx* ctx, int ks, int nt);
/* One-pass kernels (lfpsqp_projcg, lfpsqp_pcg, lfpsqp_retract_nr): 0 = on (default), -1 = off (their two-pass forms, which
 * are also the fallback for fewer than 4 or more than 1024 columns). */
int lfpsqp_ctx_set_onepass(lfpsqp_ctx* ctx, int mode);
/* Residual buffers of the fused projected-CG iteration: 0 = updated in place (default), 1 = two buffers alternating between iterations
 * (the kernel then never stores to lines it has just loaded).  Same iterates bit for bit; which is faster depends on where the buffers
 * landed in memory (FINDINGS.md 6): in-place wins by 2-4 % on well-placed buffers, the alternating scheme by ~12 % on badly placed ones. */
int lfpsqp_ctx_set_residual_buffers(lfpsqp_ctx* ctx, int mode);
/* The same switch is read once by lfpsqp_ctx_create from the environment (LFPSQP_ONEPASS=-1); the tests use it to
 * cross-check the two forms. */
/* HIP-event timing on the context's stream: begin .. end -> milliseconds */
int lfpsqp_timer_begin(lfpsqp_ctx* ctx);
int lfpsqp_timer_end(lfpsqp_ctx* ctx, double* ms);

/* ---- multi-GPU ---------------------------------------------- */
/* rows [*row0, *row1) of a global length-n vector owned by `rank` of `nranks` */
int lfpsqp_shard_range(int64_t n, int rank, int nranks, int64_t* row0, int64_t* row1);
/* RCCL: rank 0 calls unique_id (128 bytes), ships it to the other processes by
 * any means (torch.distributed / MPI.jl / a file), then every rank calls init. */
int lfpsqp_comm_unique_id(lfpsqp_ctx* ctx, void* id128);
int lfpsqp_comm_init_rccl(lfpsqp_ctx* ctx, int rank, int nranks, const void* id128);
/* Alternative transport: the host supplies the all-reduce (e.g. torch.distributed).
 * `buf` is a DEVICE pointer to `count` doubles, reduced in place (op 0 = sum, 1 = max),
 * ordered on `stream` (a hipStream_t). */
typedef int (*lfpsqp_allreduce_fn)(void* user, double* buf, int64_t count, int op, void* stream);
int lfpsqp_comm_init_callback(lfpsqp_ctx* ctx, int rank, int nranks, lfpsqp_allreduce_fn fn, void* user);
/* Third transport: a ONE-SHOT all-reduce over peer-mapped memory for the latency-bound payloads of the hot loops (2m + 5 doubles per
 * projected-CG iteration, src/projcg.jl:75,84,96,98,103; m + 1 per Newton step) -- a ring pays 2 (N - 1) hops, this one exchange.  Every
 * rank owns a mailbox in its device memory that the others map through hipIpc; a collective is one single-workgroup kernel on the
 * context's stream: publish my payload + sequence flag (system-scope release), then rank by rank in FIXED order wait for the flag and
 * accumulate (bit-identical results on every rank).  Payloads beyond 4096 doubles go in pieces.  lfpsqp_comm_p2p_export: create this rank's
 * mailbox and return its 64-byte IPC handle; ship all handles to all ranks by any control plane; lfpsqp_comm_init_p2p(handles: nranks x 64
 * bytes, in rank order) maps them.  One node (xGMI / PCIe peers), at most 16 ranks; ranks may also share a GPU (the 1-GPU test). */
int lfpsqp_comm_p2p_export(lfpsqp_ctx* ctx, void* handle64);
int lfpsqp_comm_init_p2p(lfpsqp_ctx* ctx, int rank, int nranks, const void* handles);
/* Required order: export (zero-fills the mailbox; complete before it returns) -> handles to all ranks -> init_p2p -> collectives; no barrier
 * is needed before the first collective (a peer writes only its own mailbox).  A context takes ONE communicator: a second init is an error.
 * The mailbox is FINE-GRAINED (uncached) device memory; if the runtime cannot allocate or export that kind, export fails with
 * LFPSQP_ERR_COMM unless coarse-grained memory was allowed beforehand (allow_coarse(ctx, 1), or LFPSQP_P2P_ALLOW_COARSE=1 in the
 * environment) -- never silently.  p2p_info: *mem_kind = LFPSQP_P2P_MEM_* of this rank's mailbox, *collectives = all-reduce launches so far. */
#define LFPSQP_P2P_MEM_NONE 0
#define LFPSQP_P2P_MEM_FINE 1   /* hipExtMallocWithFlags(hipDeviceMallocUncached) */
#define LFPSQP_P2P_MEM_COARSE 2 /* hipMalloc (explicitly allowed fallback) */
int lfpsqp_comm_p2p_allow_coarse(lfpsqp_ctx* ctx, int allow);
int lfpsqp_comm_p2p_info(const lfpsqp_ctx* ctx, int* mem_kind, unsigned long long* collectives);
int lfpsqp_comm_info(const lfpsqp_ctx* ctx, int* rank, int* nranks);

/* ---- buffers ------------------------------------------------------------- */
int lfpsqp_vec_alloc(lfpsqp_ctx* ctx, int64_t n, lfpsqp_vec** out); /* zero-filled */
int lfpsqp_vec_free(lfpsqp_ctx* ctx, lfpsqp_vec* v);
int64_t lfpsqp_vec_len(const lfpsqp_vec* v);
int lfpsqp_vec_upload(lfpsqp_ctx* ctx, lfpsqp_vec* v, int64_t offset, const double* host, int64_t count);
int lfpsqp_vec_download(lfpsqp_ctx* ctx, const lfpsqp_vec* v, int64_t offset, double* host, int64_t count);
int lfpsqp_vec_fill(lfpsqp_ctx* ctx, lfpsqp_vec* v, double value);
int lfpsqp_vec_copy(lfpsqp_ctx* ctx, lfpsqp_vec* dst, const lfpsqp_vec* src); /* dst .= src */
/* dst[dst_off : dst_off+count) = src[src_off : src_off+count)   (views such as view(x, 1:n)) */
int lfpsqp_vec_copy_range(lfpsqp_ctx* ctx, lfpsqp_vec* dst, int64_t dst_off, const lfpsqp_vec* src, int64_t src_off, int64_t count);
int lfpsqp_mat_alloc(lfpsqp_ctx* ctx, int64_t n, int64_t m, lfpsqp_mat** out); /* zero-filled */
int lfpsqp_mat_free(lfpsqp_ctx* ctx, lfpsqp_mat* M);
int lfpsqp_mat_shape(const lfpsqp_mat* M, int64_t* n, int64_t* m);
/* host is column-major with leading dimension ldh (>= n); columns [col0, col0+ncols) */
int lfpsqp_mat_upload(lfpsqp_ctx* ctx, lfpsqp_mat* M, int64_t col0, int64_t ncols, const double* host, int64_t ldh);
int lfpsqp_mat_download(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t col0, int64_t ncols, double* host, int64_t ldh);
int lfpsqp_mat_copy(lfpsqp_ctx* ctx, lfpsqp_mat* dst, const lfpsqp_mat* src);
/* A VIEW of A: the n x m matrix  diag(rs) * A + u w'  without a copy (rs, u: n-vectors, w: a device m-vector; rs == NULL: no scaling,
 * u == w == NULL: no rank-one term; A's storage and the vectors are borrowed and must outlive the view; lfpsqp_mat_free releases the view
 * only).  What it is for: constraint gradients of the form Jct(x) = diag(phi'(x)) A + 2 x qw' -- the reference's jac!
 * (src/autodiff_generators.jl:60-66, entry src/optimize.jl:284) rewrites the whole n x m matrix at every outer iteration and at every
 * iteration of the ProjPenalty retraction; with a view it rewrites one or two n-vectors and every solver streams the constant A
 * (lfpsqp_elementwise with Jct = a view of its own A: "streamed gradients").  A view is accepted wherever a matrix is only READ through the
 * library's product kernels: lfpsqp_gemv_t / _n, lfpsqp_gram, lfpsqp_rmul (input), lfpsqp_factorize (Jct), lfpsqp_basis.Z / .A
 * (lfpsqp_q_gemv_*, lfpsqp_projcg, lfpsqp_pcg, lfpsqp_pcg_pre), lfpsqp_constraints.Jct (lfpsqp_constraints_*, lfpsqp_retract_nr,
 * lfpsqp_retract_pp; no ball column, no sparse twin), lfpsqp_calculate_lambda_y, and as the source of lfpsqp_mat_copy (which materialises
 * it).  Everything that writes a matrix or reads its storage directly (upload / download / hash_fill, outputs, sparse twins, the batched
 * Newton retraction -- width 0) refuses a view (LFPSQP_ERR_ARG).  Cost per pass: 8 or 16 bytes per row next to 8 m, and (rank-one term)
 * two one-workgroup launches around the kernel: w't ahead of a first product, the fold of w (u'v) into the column sums behind a second. */
int lfpsqp_mat_view(lfpsqp_ctx* ctx, const lfpsqp_mat* A, const lfpsqp_vec* rs, const lfpsqp_vec* u, const lfpsqp_vec* w, lfpsqp_mat** out);
/* = lfpsqp_mat_view(ctx, A, rs, NULL, NULL, out) */
int lfpsqp_mat_rowscaled_view(lfpsqp_ctx* ctx, const lfpsqp_mat* A, const lfpsqp_vec* rs, lfpsqp_mat** out);
/* ---- placement-tuned allocation (FINDINGS.md 6) ---------------------------------------------------------------------------------------
 * On MI355X the kernels that run a small store stream inside a matrix read stream -- the fused projected-CG iteration (src/projcg.jl:93-97),
 * the Newton step, pcg! -- run 10-15 % faster or slower depending on WHERE the matrix and the n-vectors they touch were allocated: a
 * property of the pair of allocations, reproducible within a process.  These two calls allocate by trial: `tries` candidate allocations
 * (lfpsqp_ctx_set_placement, default 3, 1 = off), a few launches of the fused kernel itself on each (on zeros), the fastest kept, the rest
 * freed.  Matrices below 1 GiB, shapes without a one-pass kernel (fewer than 4 or more than 1024 columns) and candidates that would not
 * fit side by side are allocated plainly.  Results never depend on the choice.
 *   lfpsqp_mat_alloc_placed: as lfpsqp_mat_alloc; the candidates are tried with a scratch vector set.  For a long-lived matrix whose
 *     vectors come and go (Jct, src/optimize.jl:190).
 *   lfpsqp_vecs_alloc_placed: `count` (>= 3) zero-filled n-vectors carved from ONE allocation (they share its speed), tried against the
 *     matrix M (leading ncols columns) they will be streamed with; out[0..2] play the residual / direction / operator-diagonal roles in the
 *     trial (hand them to lfpsqp_projcg_work.g, .d and lfpsqp_diag_op.dg).  M == NULL: one plain allocation.  Each vector is released with
 *     lfpsqp_vec_free as usual; the allocation goes with the last of them.
 *   lfpsqp_basis_work_alloc_placed: the basis (n x m) AND `count` vectors of nvec >= n doubles together -- which allocation of the one is
 *     fast depends on the other, so every (matrix candidate, vector-set candidate) pair is tried (tries^2 trials of ~4 launches, two
 *     rounds so that a GPU coming out of idle does not handicap the first candidates) and the fastest PAIR kept.  What `optimize` wants for
 *     the basis Z and ProjCGWork (src/projcg.jl:1-11), both allocated once at src/optimize.jl:191,214. */
int lfpsqp_ctx_set_placement(lfpsqp_ctx* ctx, int tries);
int lfpsqp_mat_alloc_placed(lfpsqp_ctx* ctx, int64_t n, int64_t m, lfpsqp_mat** out);
int lfpsqp_vecs_alloc_placed(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols, int64_t n, int count, lfpsqp_vec** out);
int lfpsqp_basis_work_alloc_placed(lfpsqp_ctx* ctx, int64_t n, int64_t m, int64_t nvec, int count, lfpsqp_mat** M_out, lfpsqp_vec** out);
/* what the last placed allocation measured: trials made (0: none were; for the joint call tries_m * tries_v, row = matrix candidate), the one
 * kept, fused-kernel ms per trial (ms: ms_cap doubles, may be NULL) */
int lfpsqp_placement_info(const lfpsqp_ctx* ctx, int* tries, int* picked, double* ms, int ms_cap);
/* one trial on its own: average ms of `reps` launches of the fused kernel over M[:, :ncols] with the ZERO-filled n-vectors g, d, a (-1: no one-pass kernel for the shape) */
int lfpsqp_placement_probe(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols, lfpsqp_vec* g, lfpsqp_vec* d, lfpsqp_vec* a, int reps, double* ms);
/* synthetic inputs of SURVEY §8(d): v[i] = u(seed, offset+i);
 * M[i,j] = scale * u(seed, j*n_global + row0 + i)  (splitmix64-finaliser hash in [-1,1));
 * a power-of-two scale keeps the values bit-identical to the numpy generator */
int lfpsqp_vec_hash_fill(lfpsqp_ctx* ctx, lfpsqp_vec* v, uint64_t seed, int64_t offset, double scale, double shift);
int lfpsqp_mat_hash_fill(lfpsqp_ctx* ctx, lfpsqp_mat* M, uint64_t seed, int64_t row0, int64_t n_global, double scale, int64_t nrows,
                         int64_t ncols); /* only the leading nrows x ncols block is written */

/* ---- BLAS-1/2 primitives on the tall-skinny layout ------------------------ */
/* Replace the reference's mul!/gemv!/kgemv!/dot/norm/axpy!/broadcast call sites
 * (src/projcg.jl:55-118, src/la_helper.jl:36-44, src/retractions.jl:140-160,
 * 213-235).  n-vectors are sharded, results of reductions are global
 * (all-reduced) and replicated. */
/* t[0:ncols] = M[:, 0:ncols]' * v            (kgemv!('T', rank, 1, M, v, 0, t); mul!(t, U', v)) */
int lfpsqp_gemv_t(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols, const lfpsqp_vec* v, lfpsqp_vec* t);
/* y = alpha * M[:, 0:ncols] * t + beta * y    (kgemv!('N', ...); mul!(y, U, t, alpha, beta)) */
int lfpsqp_gemv_n(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols, double alpha, const lfpsqp_vec* t, double beta, lfpsqp_vec* y);
int lfpsqp_dot(lfpsqp_ctx* ctx, const lfpsqp_vec* x, const lfpsqp_vec* y, double* out);  /* dot(x, y) */
/* dot over the first `count` local entries only (norm(view(step, 1:n)), src/linesearch.jl:66) */
int lfpsqp_dot_head(lfpsqp_ctx* ctx, const lfpsqp_vec* x, const lfpsqp_vec* y, int64_t count, double* out);
int lfpsqp_nrm2(lfpsqp_ctx* ctx, const lfpsqp_vec* x, double* out);                      /* norm(x) */
int lfpsqp_amax(lfpsqp_ctx* ctx, const lfpsqp_vec* x, double* out);                      /* norm(x, Inf) */
int lfpsqp_axpby(lfpsqp_ctx* ctx, double a, const lfpsqp_vec* x, double b, lfpsqp_vec* y); /* y = a*x + b*y */
/* z = a*x + b*y (z may alias x or y) */
int lfpsqp_waxpby(lfpsqp_ctx* ctx, double a, const lfpsqp_vec* x, double b, const lfpsqp_vec* y, lfpsqp_vec* z);
/* y = d .* x (diagonal operator apply; z may alias) */
int lfpsqp_vmul(lfpsqp_ctx* ctx, const lfpsqp_vec* d, const lfpsqp_vec* x, lfpsqp_vec* y);
/* v[offset : offset+count) = value */
int lfpsqp_vec_fill_range(lfpsqp_ctx* ctx, lfpsqp_vec* v, int64_t offset, int64_t count, double value);
/* y[i] = a*x[i] + c for i < count (entries >= count untouched): gradients of quadratic objectives */
int lfpsqp_affine_head(lfpsqp_ctx* ctx, double a, const lfpsqp_vec* x, double c, int64_t count, lfpsqp_vec* y);
/* *out = sum_{i < count} (x[i] - c)^2   (all-reduced): quadratic objectives */
int lfpsqp_sumsq_shift(lfpsqp_ctx* ctx, const lfpsqp_vec* x, int64_t count, double c, double* out);
/* Separable objectives f(x) = sum_{i < count} phi(x_i - c_i; a_i) of device-resident problem classes (the analogue of the
 * reference's user f / grad! / the diagonal of hess_lag_vec!, src/autodiff_generators.jl:72-107, for objectives that need no AD):
 *   kind 0: a t^2      kind 1: a t^4 + t^2      kind 2: a (sqrt(1 + t^2) - 1)  (pseudo-Huber)
 *   mode 0: *out_sum = f (all-reduced);  mode 1: out_vec[i] = phi'(x_i);  mode 2: out_vec[i] = phi''(x_i)  (entries >= count untouched)
 * a / c: per-variable parameter vectors, or NULL for the constants a0 / c0. */
int lfpsqp_separable(lfpsqp_ctx* ctx, int kind, int mode, const lfpsqp_vec* a, double a0, const lfpsqp_vec* c, double c0,
                     const lfpsqp_vec* x, int64_t count, lfpsqp_vec* out_vec, double* out_sum);
/* sum-all-reduce a replicated-partials device vector across ranks (no-op for 1 rank) */
int lfpsqp_allreduce(lfpsqp_ctx* ctx, lfpsqp_vec* v, int64_t count);

/* ---- sparse constraint gradients (the reference's README.md:80 to-do) ---------------------- */
/* Jct with a few nonzeros per ROW (every variable in a few constraints; e.g. the system of test/test_retractions.jl:34-54).
 * Built from host triplets (0-based LOCAL row, column, value; duplicates add up; any order); at most 256 nonzeros per
 * row.  Stored twice on the device: ELL by rows for Jct*t (row-local) and CSC cut into fixed chunks for Jct'*v
 * (fixed-order sums, no atomics: bit-reproducible); both products move nnz*(8+4) bytes instead of 8*n*m.
 * A handle in lfpsqp_constraints.Jsp / lfpsqp_basis.S makes lfpsqp_constraints_eval and lfpsqp_pcg use them. */
int lfpsqp_spmat_create(lfpsqp_ctx* ctx, int64_t n, int64_t m, int64_t nnz, const int64_t* rows, const int64_t* cols,
                        const double* vals, lfpsqp_spmat** out);
int lfpsqp_spmat_free(lfpsqp_ctx* ctx, lfpsqp_spmat* S);
int lfpsqp_spmat_info(const lfpsqp_spmat* S, int64_t* n, int64_t* m, int64_t* nnz, int64_t* ell_width);
/* t[0:m) = S' * v  (all-reduced over ranks);  y = alpha * S * t + beta * y */
int lfpsqp_spmv_t(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const lfpsqp_vec* v, lfpsqp_vec* t);
int lfpsqp_spmv_n(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, double alpha, const lfpsqp_vec* t, double beta, lfpsqp_vec* y);
/* A second object with the STRUCTURE of S (same rows / columns, shared on the device) and its own copy of the values: the
 * x-dependent constraint gradients diag(phi'(x)) A of lfpsqp_elementwise, rescaled in place by lfpsqp_spmat_rowscale. */
int lfpsqp_spmat_clone(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, lfpsqp_spmat** out);
/* dst.values = diag(v) * src.values (v: n-vector); dst is src itself or a clone of it */
int lfpsqp_spmat_rowscale(lfpsqp_ctx* ctx, lfpsqp_spmat* dst, const lfpsqp_spmat* src, const lfpsqp_vec* v);
/* M[:, 0:m) = S as a dense matrix (the tangent setup -- lfpsqp_factorize, whose basis Z is dense anyway -- and the Newton
 * retraction keep using the dense kernels) */
int lfpsqp_spmat_to_dense(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, lfpsqp_mat* M);
/* G (host, M x M column-major) = A' diag(w2) A for A = [S | Jct[:, S.m : Jct.m)] (Jct NULL: A = S, M = S.m; w2 NULL: no weights), summed
 * over the ranks: the Gram matrix of the tangent setup from the NONZEROS.  The scattered accumulation is exact (every term cut into two
 * fixed-point limbs, 64-bit integer sums), so the result does not depend on the order in which rows are visited -- bit-reproducible, and
 * identical for any permutation of the rows.  LFPSQP_ERR_UNSUPPORTED for rows wider than 32 nonzeros or non-finite / extreme values. */
int lfpsqp_spmat_gram(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const lfpsqp_mat* Jct, const lfpsqp_vec* w2, double* G);

/* ---- bound manifolds (src/inequality_helper.jl, src/retractions.jl:451-500) ------- */
/* With bounds the reference doubles the variables: xaug = [x; y] (length 2N), each bounded
 * x_i living with its partner y_i on a line / parabola / circle.  On the device such a
 * "stacked" vector keeps the x-half at [0, N) and the y-half at [hs, hs + N) with
 * hs = lfpsqp_half_stride(N) (N rounded up to whole tiles, gap kept zero), so both halves are
 * tile-aligned and every BLAS-1 primitive above works on it unchanged (length hs + N). */
int64_t lfpsqp_half_stride(int64_t N);
/* InequalityData(xl, xu) (src/inequality_helper.jl:39-89): per-variable q, r, s, t (N-vectors);
 * the curve type is implied (s == 0 line, q == 0 parabola, else circle).  xl/xu use +-Inf. */
typedef struct lfpsqp_ineq_data {
    const lfpsqp_vec* q;
    const lfpsqp_vec* r;
    const lfpsqp_vec* s;
    const lfpsqp_vec* t;
    int64_t n; /* N */
} lfpsqp_ineq_data;
int lfpsqp_ineq_data_build(lfpsqp_ctx* ctx, const lfpsqp_vec* xl, const lfpsqp_vec* xu, lfpsqp_vec* q, lfpsqp_vec* r, lfpsqp_vec* s,
                           lfpsqp_vec* t);
/* generate_initial_y! (:92-109): fills the y-half of the stacked xaug */
int lfpsqp_generate_initial_y(lfpsqp_ctx* ctx, lfpsqp_vec* xaug, const lfpsqp_ineq_data* id);
/* calculate_h! (:112-122): h[0:N) (N-vector), *hmax = norm(h, Inf) (hmax may be NULL) */
int lfpsqp_calculate_h(lfpsqp_ctx* ctx, lfpsqp_vec* h, const lfpsqp_vec* xaug, const lfpsqp_ineq_data* id, double* hmax);
/* inequality_gradient! (:125-141): Dx, Dy, S; additionally the row scalings of the stacked
 * basis, sx = Dy.^2 and sy = -Dx.*Dy (so U = [sx .* Z; sy .* Z], see lfpsqp_factorize) */
int lfpsqp_inequality_gradient(lfpsqp_ctx* ctx, const lfpsqp_vec* xaug, const lfpsqp_ineq_data* id, lfpsqp_vec* Dx, lfpsqp_vec* Dy,
                               lfpsqp_vec* S, lfpsqp_vec* sx, lfpsqp_vec* sy);
/* calculate_lambda_kkt! second half (src/inequality_helper.jl:302-305):
 * lamy = (-Dx .* (Jct * lam) + w) ./ S   with lam (device, ncols) and w = (Q'd)[0:N] */
int lfpsqp_calculate_lambda_y(lfpsqp_ctx* ctx, const lfpsqp_mat* Jct, int64_t ncols, const lfpsqp_vec* lam, const lfpsqp_vec* Dx,
                              const lfpsqp_vec* S, const lfpsqp_vec* w, lfpsqp_vec* lamy);
/* augmented_hess_lag_vec! for a diagonal Lagrangian Hessian (src/inequality_helper.jl:144-158):
 * a (stacked) = [hx + 2 lamy.*q ; 2 lamy.*s]   (hx = diag of the user's Hessian on the x-half) */
int lfpsqp_augmented_diag(lfpsqp_ctx* ctx, const lfpsqp_vec* hx, const lfpsqp_vec* lamy, const lfpsqp_ineq_data* id, lfpsqp_vec* a);
/* e[0:N) = |Dy| .* dx - Dx .* sgn(Dy) .* dy for a stacked d = [dx; dy]: the right-hand column with which lfpsqp_factorize_rhs (w2 = sx = Dy.^2)
 * returns Jct'(sx .* dx + sy .* dy), the m-part of Q'd (src/inequality_helper.jl:197-212), from the Gram pass */
int lfpsqp_ineq_rhs(lfpsqp_ctx* ctx, const lfpsqp_vec* daug, const lfpsqp_vec* Dx, const lfpsqp_vec* Dy, lfpsqp_vec* e);
/* y_retract!(xnewaug, xaug, idata) (src/retractions.jl:451-500) */
int lfpsqp_y_retract(lfpsqp_ctx* ctx, lfpsqp_vec* xnewaug, const lfpsqp_vec* xaug, const lfpsqp_ineq_data* id);

/* ---- tangent setup: the replacement of ksvd! ----------------------------------- */
/* The reference factors the n x m constraint-gradient matrix with LAPACK dgesvd every
 * outer iteration (src/la_helper.jl:8-34, called at src/optimize.jl:291/293, O(n m^2)) and
 * then only uses: the projector U_r U_r', lambda = V S^-1 U_r'd, and NR's D = S^-1 Vt.  On
 * the device this is a Gram-based factorisation (the "(JJ') normal-equation solve" of the
 * north star): G = A' diag(w2) A on the device (MFMA), the replicated m x m eigen/SVD problems by one-sided
 * Jacobi, Z = A * W on the device (MFMA), refinement rounds when A is ill-conditioned (FINDINGS.md §5.3).
 *
 * G_host (ncols x ncols, column-major) = M[:, :ncols]' diag(w2) M[:, :ncols], all-reduced.
 * w2 == NULL means unit weights.  Weights must be >= 0: the kernel applies sqrt(w2) to both operands (they are squares in every use of the
 * hot path: Dy.^2 with bounds, phi'(x).^2 behind a view); a negative weight is answered with LFPSQP_ERR_ARG (lfpsqp_gram / _gram_rhs) or
 * LFPSQP_ERR_NUMERIC "non-finite Gram matrix" (lfpsqp_factorize*), never with NaN data. */
int lfpsqp_gram(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols, const lfpsqp_vec* w2, double* G_host);
/* Out[:, :rcols] = In[:, :kcols] * W   (W_host: kcols x rcols, column-major).  Out != In. */
int lfpsqp_rmul(lfpsqp_ctx* ctx, const lfpsqp_mat* In, int64_t kcols, const double* W_host, int64_t rcols, lfpsqp_mat* Out);
/* Thin factorisation A = diag(sqrt(w2)) * Jct = U S Vt with U = diag(sqrt(w2)) * Z:
 *   Z (n x m, device, Z != Jct): Z' diag(w2) Z = I on the leading `rank` columns, the rest zero
 *   Sigma[m] (host, descending), Vt[m*m] (host, column-major; rows >= rank are zero),
 *   rank = #{Sigma_j >= eps_rank} -- the reference's ABSOLUTE rule on dgesvd's singular values (src/optimize.jl:297-302,
 *   eps_rank = 1e-10), which decides between the Newton and the ProjPenalty retraction (:396-412).
 * Well-conditioned blocks (cond^2 <= 10: the dense random blocks of the BASELINE configs) cost one Gram + one rmul
 * pass.  Otherwise refinement rounds (one rmul + one Gram pass each, a one-sided Jacobi correction of the replicated
 * m x m factor in between) resolve the singular values to dgesvd's own accuracy, eps * Sigma_1 absolute: a Gram matrix
 * alone could not see below ~1e-8 * Sigma_1.  Z's columns are orthonormal to the rounding floor of the product
 * Jct * w_j, eps * Sigma_1 / Sigma_j.
 * With w2 == NULL, Z is the U of ksvd! up to the sign/rotation freedom of the SVD, to which
 * every use in the reference is invariant.  With bounds, w2 = Dy.^2 and the reference's
 * 2N x M factor of PJct (src/optimize.jl:288-291) is [Dy.^2 .* Z ; -Dx.*Dy .* Z].
 * W (optional, host, m x m column-major): the small factor the basis was formed with, Z = Jct * W
 * exactly as computed (columns >= rank are zero).  Handing it back through lfpsqp_basis.A / .W lets
 * the Newton retraction run both of its products over Jct alone (one matrix stream instead of two). */
/* WARM START for the next lfpsqp_factorize / lfpsqp_factorize_sp call on this context with the same m: Vt_prev (m x m, host, column-major) is
 * the Vt a previous call returned for a NEARBY matrix -- in `optimize`, the previous outer iteration's (src/optimize.jl:286-302 factorises
 * afresh every iteration; the constraint gradients move little between iterations, and with linear constraints and no bounds not at all).
 * The small eigenproblem then starts from a nearly diagonal matrix: one or two Jacobi sweeps instead of eight.  The hint is an optimisation
 * only: it is checked (orthogonal to 1e-8, m <= 256, positive definite Gram matrix) and ignored otherwise; Sigma, rank and the span of every
 * kept column are those of the cold call, the basis may differ by a rotation inside clusters of equal singular values (to which every use is
 * invariant).  Consumed by the first factorisation that follows, whatever its outcome. */
int lfpsqp_factorize_hint(lfpsqp_ctx* ctx, const double* Vt_prev, int64_t m);
int lfpsqp_factorize(lfpsqp_ctx* ctx, const lfpsqp_mat* Jct, const lfpsqp_vec* w2, lfpsqp_mat* Z, double* Sigma, double* Vt,
                     double* W, int64_t* rank, double eps_rank);
/* EXTRA RIGHT-HAND COLUMNS of the Gram pass.  Right behind the factorisation the outer iteration projects its step, d -= U (U'd)
 * (src/optimize.jl:305-307; with bounds :315-316), which starts with Jct'd -- a GEMV-T pass over the matrix the Gram kernel has just read.
 * d is known before jac! runs, so the product rides along: the Gram kernel sums, for up to two device n-vectors e_k,
 *     X[:, k] = M[:, :ncols]' (sqrt(w2) .* e_k)          (ncols x nx, host, column-major, all-reduced; w2 == NULL: M'e_k)
 * from the operand values it stages for the matrix cores anyway (vector pipe, no extra memory or LDS traffic; border columns beyond a
 * multiple of 128 by a GEMV-T over those columns alone).  The weighted form is stated in the kernel's scaled space on purpose: with bounds
 * the projection needs Jct'(sx .* dx + sy .* dy) with sx = Dy.^2 = w2, sy = -Dx.*Dy, i.e. e = |Dy| dx - Dx sgn(Dy) dy -- no division by
 * a weight that may be zero.  M may be a view (lfpsqp_mat_view): X = V'(sqrt(w2) .* e) for V = diag(rs) A + u w'.
 * The kernel reads e_k in 16-row steps WITHOUT a mask (a compare and a select between the matrix-core instructions cost 0.45 ms of the pass):
 * the entries of the rows n .. round_up(n, 16) must be finite -- they meet the matrix's zero rows.  Vectors from lfpsqp_vec_alloc have zero
 * padding and no library call writes it; columns the library itself stages (views, border columns) are written with zeros there.
 * lfpsqp_gram_rhs: G as lfpsqp_gram plus X; lfpsqp_factorize_rhs: lfpsqp_factorize plus Jte_host[0:m) = Jct'(sqrt(w2) .* e) (e may be
 * NULL: then exactly lfpsqp_factorize) and, when G_host != NULL, the m x m Gram matrix Jct' diag(w2) Jct the factors were computed from
 * (column-major; with it U'U = W'GW is known on the host: lfpsqp_tangent_step's LFPSQP_TANGENT_INIT_PROJCG). */
int lfpsqp_gram_rhs(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols, const lfpsqp_vec* w2, int64_t nx, const lfpsqp_vec* const* e,
                    double* G_host, double* X_host);
int lfpsqp_factorize_rhs(lfpsqp_ctx* ctx, const lfpsqp_mat* Jct, const lfpsqp_vec* w2, lfpsqp_mat* Z, double* Sigma, double* Vt,
                         double* W, int64_t* rank, double eps_rank, const lfpsqp_vec* e, double* Jte_host, double* G_host);

/* The same factorisation when the constraint gradients are sparse: A = [S | Jct[:, S.m : Jct.m)], i.e. the sparse object holds the
 * leading columns and the dense twin Jct (n x M, M - S.m <= 4: the ball / slack columns; NULL when there are none) the rest.  The
 * basis-forming products Z = A * W stream the NONZEROS and write the dense basis (bound by that write: a third of the dense MFMA
 * product at m = 128, K = 4); the Gram matrix comes from the nonzeros as well (lfpsqp_spmat_gram; for rows wider than 8 nonzeros, or
 * where that is refused, from the dense twin, or from S expanded into Z).  Results as lfpsqp_factorize(ctx, <dense A>, ...) up to the rounding of the Gram matrix and of the product. */
int lfpsqp_factorize_sp(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const lfpsqp_mat* Jct, const lfpsqp_vec* w2, lfpsqp_mat* Z, double* Sigma,
                        double* Vt, double* W, int64_t* rank_out, double eps_rank);
/* The replicated small step of lfpsqp_factorize on its own: thin SVD A = U diag(S) V' of a small host matrix (rows x cols,
 * column-major, rows >= 1, rows + cols <= 1024 for the device path) by one-sided Jacobi -- on the device from 64 columns on
 * (block Jacobi, csrc/jacobi.hip), on the host below that.  U: rows x cols (normalised columns), S: cols (descending),
 * V: cols x cols (may be NULL).  High relative accuracy on column-scaled matrices, which is what the refinement rounds of
 * lfpsqp_factorize rely on; the reference gets the same job done inside LAPACK's dgesvd (src/la_helper.jl:22). */
int lfpsqp_small_svd(lfpsqp_ctx* ctx, int64_t rows, int64_t cols, const double* A, double* U, double* S, double* V);

/* ---- projected CG (src/projcg.jl:40-121) --------------------------------- */
/* The symmetric operator A of the QP ("B*p", the Lagrangian Hessian action that the
 * reference wraps in a LinearMap at src/optimize.jl:228-230).  Device-resident
 * forms: A = a0*I + diag(dg)  (dg may be NULL).  Covers hess_lag_vec! for
 * quadratic f with linear/ball constraints and augmented_hess_lag_vec!
 * (src/inequality_helper.jl:144-158), whose bound terms are diagonal. */
typedef struct lfpsqp_diag_op {
    double a0;
    const lfpsqp_vec* dg; /* optional, length n */
} lfpsqp_diag_op;

/* The orthonormal operator U of projcg! ("Qview", src/optimize.jl:366-372):
 *   plain   : U = Z[:, 0:ncols]                         (view(U, :, 1:rank))
 *   stacked : Q = [[diag Dx; diag Dy], [sx .* Z; sy .* Z][:, 0:ncols]]
 *             (InequalityDecompProject, src/inequality_helper.jl:161-212, with
 *             the 2N x M factor stored as row scalings of one N x M matrix; vectors
 *             are [x-half; y-half], each half of length n_half = rows(Z)). */
typedef struct lfpsqp_basis {
    const lfpsqp_mat* Z;
    int64_t ncols;
    const lfpsqp_vec* Dx; /* NULL => plain */
    const lfpsqp_vec* Dy;
    const lfpsqp_vec* sx;
    const lfpsqp_vec* sy;
    /* optional generator of the basis: Z[:, 0:ncols] == A * W with W host, (A->m) x ncols column-major
     * (lfpsqp_factorize's W).  NULL/NULL when unknown; only an optimisation hint, never required. */
    const lfpsqp_mat* A;
    const double* W;
    /* optional sparse form of Z[:, 0:ncols] (same entries): lfpsqp_pcg then makes two sparse products per iteration
     * instead of a dense pass.  NULL when Z is dense only. */
    const lfpsqp_spmat* S;
    /* optional sparse twin of the GENERATOR: the leading SA.m columns of A (at most 4 more dense columns behind them).  With A, W and
     * SA set, lfpsqp_projcg applies the basis in factored form, U t = A (W t) and U'v = W'(A'v), on the nonzeros: no dense n x m matrix is
     * read in the loop (plain and bound-stacked bases; same projector, iterates equal to the dense form up to rounding). */
    const lfpsqp_spmat* SA;
} lfpsqp_basis;

/* mul!(dest, Q', v) (src/inequality_helper.jl:197-212): w[0:N) = Dx.*vx + Dy.*vy,
 * t[0:ncols) = Z'(sx.*vx + sy.*vy)  -- ONE pass over the N x M matrix Z (the reference makes a
 * pass over a 2N x M matrix).  For a plain basis (Dx == NULL) this is lfpsqp_gemv_t and w is unused; a plain basis that carries
 * A, W and SA is applied in factored form on the nonzeros, t = W'(A'v) (likewise lfpsqp_q_gemv_n: y = alpha A (W t) + beta y). */
int lfpsqp_q_gemv_t(lfpsqp_ctx* ctx, const lfpsqp_basis* Q, const lfpsqp_vec* v, lfpsqp_vec* w, lfpsqp_vec* t);
/* mul!(y, Q, [w; t], alpha, beta) (:161-194): y = alpha * Q [w; t] + beta * y, y stacked; w may be
 * NULL (treated as zero, e.g. the Newton-retraction update xnew += U*delta, src/retractions.jl:141) */
int lfpsqp_q_gemv_n(lfpsqp_ctx* ctx, const lfpsqp_basis* Q, double alpha, const lfpsqp_vec* w, const lfpsqp_vec* t, double beta,
                    lfpsqp_vec* y);

/* ProjCGWork (src/projcg.jl:1-11): caller-owned scratch, allocated once.
 * Only three n-vectors are needed on the device (r == g throughout; gp, Ad and -- inside the
 * loop -- rp are never materialised; rp holds the initial residual) plus the m-vector Utr. */
typedef struct lfpsqp_projcg_work {
    lfpsqp_vec* g;
    lfpsqp_vec* d;
    lfpsqp_vec* rp;
    lfpsqp_vec* Utr;
} lfpsqp_projcg_work;

#define LFPSQP_PROJCG_WANT_LAMBDA 1 /* compute lambda = U'(b - A x) (src/projcg.jl:115-118) */
/* Carry on iterating where the previous lfpsqp_projcg call on this context stopped at its iteration limit: same x, A, U, b,
 * work; `maxit` MORE iterations; *iters counts from the start of the solve.  The iterates are those of one call with the
 * larger limit (the loop state -- d, g, the CG scalars -- lives in `work` and in the context).  Only after a call that
 * ran the one-pass iteration and returned by the iteration limit, with no other library call in between; otherwise
 * LFPSQP_ERR_UNSUPPORTED.  (bench.py times K iterations of a running solve with it; optimize never needs it.) */
#define LFPSQP_PROJCG_RESUME 2
/* The start of the solve is given: x0 = 0 (c == NULL), work->rp holds r0 = A x0 - b = -b and work->Utr holds U'r0 -- what
 * lfpsqp_tangent_step leaves behind, which formed both while it projected b.  The call then skips its own residual pass (src/projcg.jl:56-59)
 * and starts with the initial projection (:60-62).  Basis in factored form over a dense generator only, plain or bound-stacked (Utr: the
 * m-part of Q'r0; the diagonal block is formed row by row) -- LFPSQP_ERR_UNSUPPORTED otherwise; iterates as without the flag up to the
 * rounding of U'r0. */
#define LFPSQP_PROJCG_START_GIVEN 4
/* The initial projection itself is done (x0 = 0, c == NULL): work->g = g0, work->d = -g0 and, behind the first A.m entries of work->Utr
 * (which must have >= A.m + 2 ncols + 5), the sums [U'g0; U'(A g0); r0'g0; g0'g0; g0'A g0; 0; 0] -- what lfpsqp_tangent_step leaves with
 * LFPSQP_TANGENT_INIT_PROJCG.  The call starts with the first iteration; b is not read (lambda is not offered: WANT_LAMBDA needs b and A x as
 * usual).  Factored basis over a dense generator, plain or bound-stacked, diagonal operator only. */
#define LFPSQP_PROJCG_START_PROJECTED 8

/* Can lfpsqp_projcg run on a basis kept in FACTORED form (lfpsqp_basis.Z == NULL, generator A (N rows) and W given; SA = the sparse twin of
 * A's leading columns, or NULL) with a diagonal operator on THIS context?  *yes = 1 / 0.  The factored form needs the fused one-pass
 * iteration over A (4 .. 1024 columns, leading dimension inside the 32-bit lane offsets, one-pass kernels not switched off by
 * lfpsqp_ctx_set_onepass / LFPSQP_ONEPASS=-1) or, with SA, a twin whose shape the nonzero path covers.  Callers (optimize) ask BEFORE they
 * decide not to allocate Z; a "no" means: materialise Z = A W (lfpsqp_factorize with Z != NULL), every path then has its two-pass form. */
int lfpsqp_factored_basis_supported(const lfpsqp_ctx* ctx, const lfpsqp_mat* A, const lfpsqp_spmat* SA, int* yes);

/* projcg!(x, lambda, A, U, b, c; tol, maxit, work) -> (iters, nr).
 * c == NULL means c = 0 (always the case in optimize, src/optimize.jl:213,368,371).
 * n_global = length(b) of the reference summed over ranks (2N for a stacked basis); the loop
 * bound is min(maxit, length(b) + length(c)).  For a stacked basis all n-vectors (x, b, work,
 * A.dg) are stacked [x-half | gap | y-half], c must be NULL and lambda (if wanted) has length
 * >= N + ncols and receives [Dx.*rx + Dy.*ry ; U'r] like the reference's lambda.
 * Exit semantics are the reference's: negative curvature => x = d/||d||,
 * lambda = NaN, *nr = +Inf; rg <= 0 => break; nr < tol => break. */
int lfpsqp_projcg(lfpsqp_ctx* ctx, lfpsqp_vec* x, lfpsqp_vec* lambda, const lfpsqp_diag_op* A,
                  const lfpsqp_basis* U, const lfpsqp_vec* b, const lfpsqp_vec* c, double tol, int64_t maxit,
                  int64_t n_global, int flags, const lfpsqp_projcg_work* work, int64_t* iters, double* nr);

/* The same solver, still ONE pass over U per iteration, for a Hessian of the form
 *     A = a0*I + diag(dg) + V diag(sigma) V'        (V: n x k device matrix, k <= 8; sigma host, NULL = ones)
 * -- a separable objective with a few coupling terms, a quasi-Newton (L-BFGS-like) model of the Lagrangian Hessian ... -- which the reference
 * would wrap in a LinearMap (src/optimize.jl:228-230) and lfpsqp_projcg_op would run with two passes per iteration.  The low-rank term is
 * not row-local, but every product the iteration needs follows from k-vectors: (A d)_i = D_i d_i + V_i (sigma .* V'd) with V'd updated by
 * d's own recurrence, U'(A g) = U'(D g) + (U'V)(sigma .* V'g) with U'V formed once per solve (k thin passes) and V'g as k more reduction
 * terms of the pass.  Plain dense basis (materialised or factored), 4 .. 1024 columns; no RESUME / START_GIVEN; k == 0 is lfpsqp_projcg.
 * Iterates, counts and exits as projcg! with A as a matrix (src/projcg.jl:40-121), to rounding. */
typedef struct lfpsqp_lowrank_op {
    double a0;
    const lfpsqp_vec* dg; /* optional, length n */
    const lfpsqp_mat* V;  /* n x (>= k), plain */
    int64_t k;
    const double* sigma;  /* host, k; NULL = ones */
} lfpsqp_lowrank_op;
int lfpsqp_projcg_lowrank(lfpsqp_ctx* ctx, lfpsqp_vec* x, lfpsqp_vec* lambda, const lfpsqp_lowrank_op* A, const lfpsqp_basis* U,
                          const lfpsqp_vec* b, const lfpsqp_vec* c, double tol, int64_t maxit, int64_t n_global, int flags,
                          const lfpsqp_projcg_work* work, int64_t* iters, double* nr);

/* ... and for a TRIDIAGONAL Hessian (a chain / finite-difference / smoothing term: src/autodiff_generators.jl:72-107 builds hess_lag_vec!
 * for such objectives, the reference wraps it in a LinearMap, src/optimize.jl:228-230):
 *     (A v)_i = (a0 + dg_i) v_i + off_{i-1} v_{i-1} + off_i v_{i+1}      (off: length n, off_i couples rows i and i+1, off_{n-1} is ignored)
 * still ONE pass over U per iteration where lfpsqp_projcg_op pays two.  The projected residual's neighbours do not exist while a pass runs, but
 * gp = rr - U t with t known before the pass and rr = g + alpha A d made of stored vectors: the second product carries A rr (row-local) and the
 * post-op subtracts (U'A U) t; U'A U (m x m) is formed once per solve by two or three weighted Gram passes over U on the matrix cores.  Av: a scratch vector of length(b) (it receives A d of every iteration).  Plain dense basis (materialised or
 * factored, no matrix view), 4 .. 1024 columns, no bounds, one rank, no RESUME / START_PROJECTED (START_GIVEN as for lfpsqp_projcg: r0 and U'r0 do not involve A) -- otherwise
 * LFPSQP_ERR_UNSUPPORTED (use lfpsqp_projcg_op).  Iterates, counts and exits as projcg! with A as a matrix (src/projcg.jl:40-121), to rounding.
 * lfpsqp_tridiag_mul: out = A v (out != v), the operator on its own (mul! of the LinearMap). */
typedef struct lfpsqp_tridiag_op {
    double a0;
    const lfpsqp_vec* dg;  /* optional, length n */
    const lfpsqp_vec* off; /* length n */
} lfpsqp_tridiag_op;
int lfpsqp_projcg_tridiag(lfpsqp_ctx* ctx, lfpsqp_vec* x, lfpsqp_vec* lambda, const lfpsqp_tridiag_op* A, lfpsqp_vec* Av, const lfpsqp_basis* U,
                          const lfpsqp_vec* b, const lfpsqp_vec* c, double tol, int64_t maxit, int64_t n_global, int flags,
                          const lfpsqp_projcg_work* work, int64_t* iters, double* nr);
int lfpsqp_tridiag_mul(lfpsqp_ctx* ctx, const lfpsqp_tridiag_op* A, const lfpsqp_vec* v, lfpsqp_vec* out);

/* The same solver for a GENERAL symmetric operator A -- the reference's LinearMap closure around hess_lag_vec! /
 * augmented_hess_lag_vec! (src/optimize.jl:228-230, applied at src/projcg.jl:57,74,116): `A(user, src, dest)` must produce
 * dest = A * src for device vectors of length(b) (stacked [x | gap | y] when U is a stacked basis), return 0, and leave
 * src untouched.  It is called from the host once per iteration (plus once at the start and once for lambda); it may
 * be synchronous, or queue its work on the context's stream (lfpsqp_ctx_stream; every lfpsqp_* primitive does) -- then
 * nothing in the loop waits for the device: scalars, exits and d'(A d) stay on the device as in lfpsqp_projcg, and the two
 * passes over U per iteration read the product from `Av` (caller-owned n-vector scratch, the reference's work.Ad).
 * Everything else (arguments, exit semantics, outputs) as lfpsqp_projcg; LFPSQP_PROJCG_RESUME is not supported. */
typedef int (*lfpsqp_opfun)(void* user, const lfpsqp_vec* src, lfpsqp_vec* dest);
int lfpsqp_projcg_op(lfpsqp_ctx* ctx, lfpsqp_vec* x, lfpsqp_vec* lambda, lfpsqp_opfun A, void* user, lfpsqp_vec* Av,
                     const lfpsqp_basis* U, const lfpsqp_vec* b, const lfpsqp_vec* c, double tol, int64_t maxit,
                     int64_t n_global, int flags, const lfpsqp_projcg_work* work, int64_t* iters, double* nr);
/* the HIP stream (hipStream_t) all of the context's work is queued on, for callbacks that launch their own kernels */
int lfpsqp_ctx_stream(lfpsqp_ctx* ctx, void** stream);

/* ---- retractions (src/retractions.jl) ---------------------------------------------- */
/* Device-resident NONLINEAR equality constraints (problem classes beyond linear / ball / box) -- the contract
 * of the reference's user callbacks c!(cval, x), jac!(J, cval, x) and of the constraint part of hess_lag_vec!
 * (src/autodiff_generators.jl:72-107, entry src/optimize.jl:119) for constraints of the form
 *
 *     c(x) = A' phi(x) + qw * sum_{i < n_x} x_i^2 - b,        phi(x)_i = phi_{kind_i}(x_i),
 *
 * kind_i in {0: t, 1: sin t, 2: t^2} per VARIABLE (stored as doubles; NULL = all 0).  The reference's own nonlinear test
 * systems are of this form: the sin system c_i = x_{2i} - sin x_{2i-1} (test/test_retractions.jl:34-54: A sparse with
 * entries +-1, kind = sin on the odd variables) and the sphere system c_i = |x - center_i|^2 - R_i^2
 * (test/test_retractions.jl:1-31: A = -2 [center_1 ... center_m], kind = NULL, qw = 1, b_i = R_i^2 - |center_i|^2).
 * The constraint gradients are a ROW-SCALED copy of the constant A plus a rank-one term,
 *     Jct(x) = diag(phi'(x)) A + 2 [x_i (i < n_x)] qw',
 * refreshed in place by lfpsqp_constraints_jac (the reference's jac! fills J the same way every outer iteration); the
 * constraint part of the Lagrangian Hessian is DIAGONAL, diag(phi''(x) .* (A lam)) + 2 (qw'lam) I_{i < n_x}
 * (lfpsqp_constraints_hess_diag), so projcg! keeps its fused diagonal-operator path.  Every solver that takes an
 * lfpsqp_constraints (lfpsqp_constraints_eval / _jac, lfpsqp_retract_nr, lfpsqp_retract_pp) honours `ew`; c! streams A
 * (or its nonzeros) once with phi applied on the fly -- nothing n-sized crosses PCIe.
 *   A    : constant n x m_lin coefficients (may be NULL when Asp is given).  lfpsqp_constraints.Jct is a DIFFERENT matrix
 *          (>= m_lin columns) -- or, STREAMED gradients, a view of A itself: lfpsqp_mat_view(A, rs, u, w) with rs given iff kind != NULL and
 *          u, w given iff qw != NULL (w holding qw on the device).  Then Jct(x) is never written: lfpsqp_constraints_jac refreshes
 *          rs = phi'(x) and u = 2 x [i < n_x] (n-vectors instead of n x m_lin doubles) and the tangent setup, projcg!, both retractions and the
 *          multiplier products stream A with the scale and the rank-one term applied in registers.  Needs a dense A and no ball.
 *   Asp  : optional sparse form of A (same entries).  lfpsqp_constraints.Jsp must then be a structural clone
 *          (lfpsqp_spmat_clone): lfpsqp_constraints_jac rescales its values and expands it into Jct[:, 0:m_lin).
 *          With Asp, qw must be NULL (the rank-one term would make the gradients dense) and `work` is required.
 *   qw   : host, m_lin weights of the common quadratic term, or NULL.  Not together with has_ball.
 *   work : n-vector scratch (phi(x) for the sparse product). */
typedef struct lfpsqp_elementwise {
    const lfpsqp_mat* A;
    const lfpsqp_spmat* Asp;
    const lfpsqp_vec* kind;
    const double* qw;
    lfpsqp_vec* work;
} lfpsqp_elementwise;
/* Device-resident equality constraints of the BASELINE configs:
 *     c(x) = [ J x - b ;  sum_{i < n_x} x_i^2 - R2 - x[slack_row] ]
 * Jct is the N x M constraint-gradient matrix (the reference's Jct, src/optimize.jl:190,284);
 * its first m_lin columns are the constant gradients of the linear equalities; when has_ball,
 * column m_lin is the ball gradient [2x; -1] (refreshed by lfpsqp_constraints_jac) and the
 * inequality x'x <= R2 has been turned into an equality with a slack variable
 * (src/optimize.jl:23-51).  n_x / slack_row are LOCAL row indices on this rank
 * (slack_row = -1 if another rank owns the slack variable). */
typedef struct lfpsqp_constraints {
    const lfpsqp_mat* Jct;
    int64_t m_lin;
    const double* b; /* host, m_lin */
    int has_ball;
    double R2;
    int64_t n_x;
    int64_t slack_row;
    /* optional sparse form of Jct[:, 0:m_lin] (same entries): c! then streams its nonzeros instead of the dense block */
    const lfpsqp_spmat* Jsp;
    /* optional: the first m_lin constraints are NONLINEAR (elementwise-transformed linear, above); NULL = linear */
    const lfpsqp_elementwise* ew;
} lfpsqp_constraints;

/* hx[i] += phi''(x_i) (A lam)_i + 2 (qw'lam + [has_ball] lam[m_lin]) [i < n_x]   -- the diagonal of sum_j lam_j grad^2 c_j(x) added to
 * the caller's diagonal of grad^2 f (hess_lag_vec!, src/autodiff_generators.jl:80-104, for this constraint class; for a linear
 * class only the ball term remains).  lam: host, m_lin + has_ball.  Entries of hx beyond rows(Jct) are untouched. */
int lfpsqp_constraints_hess_diag(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, const lfpsqp_vec* x, const double* lam, lfpsqp_vec* hx);
/* c!(cval, x): cval (host, m_lin + has_ball).  x has >= rows(Jct) entries (the x-half of a
 * stacked vector is fine). */
int lfpsqp_constraints_eval(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, const lfpsqp_vec* x, double* cval);
/* jac!(J, cval, x): refreshes the x-dependent column(s) of Jct in place and evaluates cval.  cval == NULL: the gradients only -- for a caller
 * that holds c(x) already: the point an outer iteration starts from is the line search's accepted trial point, whose retraction returned
 * c!(xnew) (src/optimize.jl:284 evaluates it again: one pass over the constraint gradients per outer iteration for nothing). */
int lfpsqp_constraints_jac(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, const lfpsqp_vec* x, lfpsqp_mat* Jct, double* cval);
/* The TANGENT STEP of an outer iteration in one pass over the constraint gradients (src/optimize.jl:305-343, 366-381 and the head of projcg!,
 * src/projcg.jl:56-59), for a basis kept in factored form U = A W (lfpsqp_basis.Z == NULL) of rank U->ncols <= m.
 * PLAIN basis (no bounds; A may be a view):
 *   Jtd (host, m) = A'd -- from lfpsqp_factorize_rhs, which summed it during the Gram pass;
 *   Utd (host, m) <- U'd = W'Jtd (the reference's tmp_m, zero beyond the rank);  lam (host, m) <- V S^-1 Utd  (lambda_kkt, :331-343);
 *   d <- d - U Utd (:306-307);   work->rp <- r0 = -d;   work->Utr <- U'r0   (projcg!'s start for b = d, x0 = 0: LFPSQP_PROJCG_START_GIVEN);
 *   *d_sumsq <- d'd of the projected step (all-reduced; norm(d) of the truncated-Newton tolerance, :373-375);
 *   cons != NULL: hdiag (which the caller filled with the objective's part of the diagonal Lagrangian Hessian) += the constraint class's
 *   term for lam, exactly lfpsqp_constraints_hess_diag(ctx, cons, x, lam, hdiag) -- inside the same pass when the class streams its
 *   gradients from A's storage (phi''(x) .* (A lam) needs a product over the matrix being read anyway), by that call otherwise.
 *   idata = hx = S = lamy = NULL.
 * STACKED basis (bounds: U->Dx .. U->sy set, vectors [x-half | y-half]; A plain):
 *   Jtd = A'(sx .* dx + sy .* dy) -- lfpsqp_factorize_rhs with w2 = sx and the column e of lfpsqp_ineq_rhs;
 *   d <- d - Q Q'd (:315-316);  lam as above;  lamy (n, optional) <- (Dx .* dx + Dy .* dy - Dx .* (A lam)) ./ S  (calculate_lambda_kkt!,
 *   src/inequality_helper.jl:302-305);  hdiag (stacked) <- [hx (+ the class's constant term on the rows < n_x) + 2 lamy .* q ; 2 lamy .* s]
 *   (augmented_hess_lag_vec! for a diagonal Hessian, :144-158; hx: n, the objective's part -- or, with cons == NULL, the whole diagonal);
 *   work->rp <- -d, work->Utr <- Z'(sx .* r0x + sy .* r0y): the m-part of Q'r0.   S = the norms of lfpsqp_inequality_gradient.
 * Five passes of the statement-by-statement sequence (GEMV-T, GEMV-N, the GEMV-N of the Hessian term / of lambda_y, projcg!'s GEMV-T, its
 * initial projection) become two (this one and the initial projection).  LFPSQP_ERR_UNSUPPORTED for a materialised or sparse-twinned basis,
 * and for bounds together with a view or a nonlinear class's phi'' term. */
/* flags: LFPSQP_TANGENT_INIT_PROJCG -- the pass is ALSO projcg!'s initial projection (src/projcg.jl:58-62) for the solve that follows with
 * A = diag(hdiag), b = the projected d, x0 = 0: U'r0 = -(I - U'U) U'd follows from G (host, m x m: the Gram matrix lfpsqp_factorize_rhs returns,
 * U'U = W'GW) instead of a pass of its own, a further first product brings (U U'r0)_i, and the row goes on to g0 = r0 - (U U'r0)_i, work->d = -g0,
 * the second products of g0 and A g0 and the sums the first iteration needs -- left in work->g, work->d and work->Utr[m ..] for
 * LFPSQP_PROJCG_START_PROJECTED (work->Utr must have >= 3 m + 5 entries; work->rp is not written).  One more pass less per outer iteration.
 * Only for a full-rank block with cond^2 = (Sigma_1 / Sigma_m)^2 <= 10 -- where lfpsqp_factorize itself takes its factors from the Gram matrix
 * alone; beyond that G cannot resolve I - U'U (its rounding enters divided by Sigma_j^2) and the call answers LFPSQP_ERR_UNSUPPORTED: run it
 * without the flag and let projcg! measure U'r0 (LFPSQP_PROJCG_START_GIVEN). */
#define LFPSQP_TANGENT_INIT_PROJCG 1
int lfpsqp_tangent_step(lfpsqp_ctx* ctx, const lfpsqp_basis* U, const double* Sigma, const double* Vt, int64_t m, const double* Jtd,
                        const double* G, lfpsqp_vec* d, const lfpsqp_constraints* cons, const lfpsqp_vec* x, lfpsqp_vec* hdiag,
                        const lfpsqp_ineq_data* idata, const lfpsqp_vec* hx, const lfpsqp_vec* S, lfpsqp_vec* lamy,
                        const lfpsqp_projcg_work* work, int flags, double* Utd, double* lam, double* d_sumsq);
/* A user c!: x is the DEVICE vector (download it if the function is host code); return 0. */
typedef int (*lfpsqp_cfun)(void* user, const lfpsqp_vec* x, double* cval);

/* retract!(cval, xnew, c!, xtilde, x, method::NR) (src/retractions.jl:75-177): Newton-Raphson on
 * c(xtilde + U delta) with the inverse Jacobian Sigma^-1 Vt frozen at x and good-Broyden updates.
 * U / Sigma / Vt are the factors of lfpsqp_factorize (NR.U, NR.Sigma, NR.Vt); idata != NULL means
 * bounds are present (NR.ineq): vectors are stacked and y_retract! runs before every c!.
 * Exactly one of cons / cfun is used (cfun wins if non-NULL).
 * Per Newton step the device-resident path streams U (x += U*delta) and Jct (c!) in one launch; when U carries
 * its generator (U->A == cons->Jct, U->W from lfpsqp_factorize) the step is x += [sx;sy] .* (Jct*(W*delta)) and
 * Jct is streamed ONCE per step, each row tile held in registers between the two products (same iterates up
 * to rounding: Z*delta vs Jct*(W*delta)).  With a sparse twin of the linear block besides (cons->Jsp, all but <= 4 of
 * the generator's columns) no dense matrix is read at all: the step runs on the nonzeros (+ those few dense columns).
 * Outputs: xnew, cval[m] (== c!(xnew) on exit), *flag (0 ok, 1 = maxiter reached), *iters. */
int lfpsqp_retract_nr(lfpsqp_ctx* ctx, const lfpsqp_basis* U, const double* Sigma, const double* Vt, int64_t m,
                      const lfpsqp_constraints* cons, lfpsqp_cfun cfun, void* cuser, const lfpsqp_ineq_data* idata,
                      const lfpsqp_vec* xtilde, const lfpsqp_vec* x, lfpsqp_vec* xnew, double tol, int64_t maxiter, double* cval,
                      int* flag, int64_t* iters);

/* nb (2..16) Newton retractions at once: the trial points xtilde[b] of one linesearch (x + alpha d, x + alpha s d, ...;
 * src/linesearch.jl:49-60) are independent, so they share every pass over Jct -- one launch advances all unfinished
 * trials by one Newton step.  Per trial the arithmetic, the convergence test and the outputs are those of
 * lfpsqp_retract_nr: xnew[b], cval[b*m .. b*m+m), flags[b], iters[b].  Needs the one-stream step (U->A / U->W known,
 * device-resident constraints without a sparse twin, 4..1024 columns); returns LFPSQP_ERR_UNSUPPORTED otherwise (retract one by one then).
 * Two modes (lfpsqp_ctx_set_nr_batch_mode):
 *   LFPSQP_NR_BATCH_EXACT (default): up to 4 trials per pass on the VALU form of the one-pass kernel, with the rows cut into the spans of
 *     the single-trial step's launch and the second stage shaped like its own -- every sum of a trial is formed from the same operands in
 *     the same order as in lfpsqp_retract_nr, so xnew[b], cval, flags[b] and iters[b] are BIT FOR BIT those of the one-by-one call.  A
 *     linesearch that batches its failing trial steps (src/linesearch.jl:57-60) is then the same search, also where it is chaotic.
 *   LFPSQP_NR_BATCH_MATRIX_CORES (opt-in): 3..16 trials on the matrix cores (v_mfma_f64_16x16x4_f64: both products of a step are contractions
 *     once the trials are stacked; up to 132 generator columns and 128 linear constraints; 8 trials from 133 to 528 columns).  Sums in
 *     groups of four rows: equal to lfpsqp_retract_nr up to rounding (1e-12 for trials that converge from nearby) -- a trial that runs into
 *     its iteration limit may end elsewhere, and a chaotic search may then accept another step than the one-by-one search.
 * lfpsqp_retract_nr_batch_width: how many trials a pass takes for this basis and these constraints in the current mode -- 16, 8, 4 or 0
 * (cannot batch). */
enum { LFPSQP_NR_BATCH_EXACT = 0, LFPSQP_NR_BATCH_MATRIX_CORES = 1 };
int lfpsqp_ctx_set_nr_batch_mode(lfpsqp_ctx* ctx, int mode);
int lfpsqp_retract_nr_batch_width(const lfpsqp_ctx* ctx, const lfpsqp_basis* U, const lfpsqp_constraints* cons, int* width);
int lfpsqp_retract_nr_batch(lfpsqp_ctx* ctx, const lfpsqp_basis* U, const double* Sigma, const double* Vt, int64_t m,
                            const lfpsqp_constraints* cons, const lfpsqp_ineq_data* idata, int nb, const lfpsqp_vec* const* xtilde,
                            const lfpsqp_vec* x, lfpsqp_vec* const* xnew, double tol, int64_t maxiter, double* cval, int* flags,
                            int64_t* iters);

/* pcg!(mu, J, no_precondition, x, r, p, z, tmp_m, tol, maxiter) (src/retractions.jl:179-246): CG on
 * (J'J + mu I) x = b, the inner solve of the ProjPenalty retraction (:375), fused on the device
 * (3 kernels per iteration, scalars and exit status in device memory).  J' is given in the
 * lfpsqp_basis form: plain J' = Z[:, :ncols] (= Jct; no bounds), or stacked
 * J' = [[diag Dx, sx.*Z]; [diag Dy, sy.*Z]] -- with Dx := Dx.*S, Dy := Dy.*S, sx = 1, sy = 0 this is
 * the reference's InequalityDecomp (src/inequality_helper.jl:215-271, fulljac at src/retractions.jl:324).
 * In: x = initial guess (the reference passes zeros), r = right-hand side.  Out: x, r = residual,
 * *flag = 1 iff *iters == maxiter (reference quirk :240-243), *iters.  p, z: n-vector work;
 * tmp_w: N-vector work (stacked only); tmp_m: >= ncols work. */
int lfpsqp_pcg(lfpsqp_ctx* ctx, double mu, const lfpsqp_basis* Jop, lfpsqp_vec* x, lfpsqp_vec* r, lfpsqp_vec* p, lfpsqp_vec* z,
               lfpsqp_vec* tmp_w, lfpsqp_vec* tmp_m, double tol, int64_t maxiter, int* flag, int64_t* iters);

/* pcg! with an EXACT preconditioner M! (src/retractions.jl:209): the operator is A_f = D0 + E E' with E = [Jct; 0] and D0 = mu I (plain) or
 * mu I plus the 2 x 2 blocks of the inequality rows (stacked), so M^-1 = D0^-1 - D0^-1 E K E' D0^-1 with the m x m matrix
 * K = (I + E' D0^-1 E)^-1.  For the plain operator and K = mu W diag(s^2 / (mu + s^2)) W' (U = Jct W, Sigma = s) this is the reference's
 * proj_precondition!(z, r, mu, U, Sigma, rank, tmp_m) (src/retractions.jl:248-257, the call commented out at :374); with K from the CURRENT
 * Jct the solve converges in one iteration (test/test_retractions.jl:126-139).  Two passes over Jct per iteration (lfpsqp_pcg: one; its
 * iteration counts on ill-conditioned bound problems run into the thousands).
 *   K   host, m x m column-major (symmetric)
 *   i11, i12, i22   stacked operator only: the rows of D0^-1 = [i11 i12; i12 i22] (N-vectors); NULL for the plain operator (D0^-1 = 1/mu)
 *   q   work n-vector (A_f p)
 * Same outputs and exit semantics as lfpsqp_pcg.  LFPSQP_ERR_UNSUPPORTED without the one-pass kernels (sparse twin, m < 4 or > 1024). */
typedef struct lfpsqp_pcg_precond {
    const double* K;
    const lfpsqp_vec *i11, *i12, *i22;
    lfpsqp_vec* q;
} lfpsqp_pcg_precond;
int lfpsqp_pcg_pre(lfpsqp_ctx* ctx, double mu, const lfpsqp_basis* Jop, const lfpsqp_pcg_precond* P, lfpsqp_vec* x, lfpsqp_vec* r,
                   lfpsqp_vec* p, lfpsqp_vec* z, double tol, int64_t maxiter, int* flag, int64_t* iters);

/* A user jac!: refresh the DEVICE matrix Jct (rows(Jct) x m, the reference's Jct = Jc') and cval at x. */
typedef int (*lfpsqp_jacfun)(void* user, const lfpsqp_vec* x, lfpsqp_mat* Jct, double* cval);

/* ProjPenaltyWork (src/retractions.jl:21-33): n-vectors r, p, z, dx, g (stacked when bounds exist), the
 * m-vector tmp_m; with bounds additionally the N-vectors tmp_w, h, DxS, DyS, ones (filled with 1), zeros. */
typedef struct lfpsqp_pp_work {
    lfpsqp_vec *r, *p, *z, *dx, *g, *tmp_m;
    lfpsqp_vec *tmp_w, *h, *DxS, *DyS, *ones, *zeros;
    /* OPTIONAL (all NULL / 0 = the reference's live path, no_precondition): q != NULL and precondition != 0 make every inner solve a
     * lfpsqp_pcg_pre with the exact preconditioner of ITS operator -- K = (I + Jct' D0^-1 Jct)^-1 from a Gram pass over the current Jct
     * (weighted by i11 with bounds), rebuilt whenever mu or the point changes; q: n-vector work (stacked with bounds); i11, i12, i22:
     * N-vector work, bounds only.  The inner solves then take one or two iterations instead of hundreds to thousands. */
    lfpsqp_vec *q, *i11, *i12, *i22;
    int precondition;
} lfpsqp_pp_work;

/* retract!(cval, xnew, c!, xtilde, x, method::ProjPenalty) (src/retractions.jl:265-441), the reference's
 * DEFAULT retraction: Gauss-Newton on 1/2 |c(z)|^2 + mu/2 |z - xtilde|^2 with mu -> 0, inner lfpsqp_pcg,
 * Armijo backtracking -- including the reference's quirks (stale cval in the backtracking test :417,
 * flag 3 leaves only the inner loop :422-425, flag 1 iff iters == maxiter :435-437).
 * Constraints: cons (device-resident) or the callbacks cfun + jacfun (both non-NULL).  With bounds
 * (idata != NULL) Dx, Dy, S are the driver's decomposition vectors and are OVERWRITTEN at the trial points,
 * exactly as the reference's shared `idecomp` is (src/optimize.jl:235, src/retractions.jl:344).
 * Outputs: xnew, cval[m], *flag (0 ok, 1 maxiter, 2 pcg maxiter, 3 backtracking failed), *iters, *pcg_iters. */
int lfpsqp_retract_pp(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, lfpsqp_cfun cfun, lfpsqp_jacfun jacfun, void* user,
                      lfpsqp_mat* Jct, int64_t m, const lfpsqp_ineq_data* idata, lfpsqp_vec* Dx, lfpsqp_vec* Dy, lfpsqp_vec* S,
                      const lfpsqp_vec* xtilde, const lfpsqp_vec* x, lfpsqp_vec* xnew, double mu0, double tol, int64_t maxiter,
                      int64_t maxiter_pcg, const lfpsqp_pp_work* work, double* cval, int* flag, int64_t* iters, int64_t* pcg_iters);

/* per-kernel-family device time (ms) of every 4th launch (sampled: event records between kernels are not free), accumulated by the last lfpsqp_projcg call when
 * the context was created with profiling on (lfpsqp_ctx_set_profiling); used by
 * bench.py for the roofline object.  slots: 0 = K1 (direction update), 3 = F (the one-pass kernel:
 * rp, gp = rp - U t, U'gp, U'(A gp) and the dots in ONE pass over U); on the two-pass fallback instead
 * 1 = K2 (x/rp update fused with U' rp) and 2 = K3 (gp = rp - U t fused with dots).  counts[] = launches per slot. */
int lfpsqp_ctx_set_profiling(lfpsqp_ctx* ctx, int on);
int lfpsqp_profile_read(lfpsqp_ctx* ctx, double ms[8], int64_t counts[8]);

#ifdef __cplusplus
}
#endif
#endif /* LFPSQP_HIP_H */
