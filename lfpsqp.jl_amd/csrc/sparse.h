// Sparse constraint gradients: the device object and the row accessor shared by sparse.hip and the solvers that fuse a
// sparse product into their row functors (pcg.hip).
#pragma once
#include "internal.h"

struct lfpsqp_spmat {
    int64_t n = 0, m = 0, nnz = 0;
    int K = 0;              // ELL width
    double amax = 0.0;      // largest |value| (NaN when an entry is not finite)
    double* col_scale = nullptr;   // [2 m] (device): 2^-e_j with |value| 2^-e_j < 1 for every entry of column j, then 2^e_j (sp_gram)
    int64_t ld = 0;         // rows rounded up to whole tiles
    double* ell_val = nullptr;    // [K][ld]
    int32_t* ell_col = nullptr;   // [K][ld]
    int32_t* csc_row = nullptr;   // [nnz], ascending inside a column
    double* csc_val = nullptr;    // [nnz]
    int64_t nchunks = 0;
    int64_t* colptr = nullptr;      // [m + 1] (device): nonzero range of a column
    int64_t* chunk_beg = nullptr;   // [nchunks + 1] (device): nonzero range of a chunk
    int32_t* col_chunk = nullptr;   // [m + 1] (device): chunk range of a column
    int32_t* box_lo = nullptr;      // [ceil(n / kSpBox)] (device): smallest / largest column index among the nonzeros of a block of kSpBox rows
    int32_t* box_hi = nullptr;      //   (an empty block: lo > hi) -- the Gram kernels skip the blocks that cannot touch their tile
    double run_frac = 0.0;          // share of the rows whose column set is that of the row 128 places earlier (structured systems: ~1)
    bool owns_structure = true;     // false for lfpsqp_spmat_clone's objects: the index arrays belong to the original
};

namespace lfpsqp {

constexpr int kSpBox = 4096;      // rows per column bounding box of lfpsqp_spmat

// acc(i), acc(i+1) = (Jct t)[i], [i+1] from the ELL arrays
struct EllRows {
    const double* val;
    const int32_t* col;
    int64_t ld;
    int K;
    const double* t;
    __device__ __forceinline__ double2 acc(int64_t i) const {
        double2 a = make_double2(0.0, 0.0);
        for (int k = 0; k < K; ++k) {
            const double2 v = ld2(val + (int64_t)k * ld + i);
            const int2 c = *reinterpret_cast<const int2*>(col + (int64_t)k * ld + i);
            a.x = fma(v.x, t[c.x], a.x);
            a.y = fma(v.y, t[c.y], a.y);
        }
        return a;
    }
};
inline EllRows ell_rows(const lfpsqp_spmat* S, const double* t) { return EllRows{S->ell_val, S->ell_col, S->ld, S->K, t}; }

// t_out[0:m) = S' v (global: all-reduced).  v must hold at least S->n entries.
int spmv_t(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const double* v, double* t_out);

// Out[:, :r] = [S | X[:, x0 : x0+nx)] * W   (W on the device, (S->m + nx) x r column-major with leading dimension ldw; X dense, may be
// null with nx = 0; nx <= 4).  Row-parallel, fixed summation order (ELL order, then the dense columns): bit-reproducible.
// t_out[0:m) = W' tA and optionally u_out[0:wm) = W t_out (W on the device, wm x m column-major): the replicated small step of a basis in
// factored form U = A W
int sp_basis_small(lfpsqp_ctx* ctx, const double* W_dev, int wm, int m, const double* tA, double* t_out, double* u_out);
// W (host, wm x m) onto the device (ctx->small) with two wm-vectors of scratch behind it: the replicated part of a basis in factored form
// U = A W, with or without a sparse twin of A (dense generator: lfpsqp_basis.Z == NULL)
int factored_setup(lfpsqp_ctx* ctx, const lfpsqp_mat* A, const double* W_host, int m, double** dW, double** tA, double** uA);
// u_out[0:wm) = W t (t: m entries on the device)
int factored_w_times_t(lfpsqp_ctx* ctx, const double* W_dev, int wm, int m, const double* t, double* u_out);
// the factored basis U = A W, A = [SA | A[:, SA.m:)] (at most 4 dense columns), W host (A.m x m), applied without a dense n x m matrix:
// t_out = U'v   /   y = alpha U t + beta y   (v, t, y device pointers)
int sp_factored_gemv_t(lfpsqp_ctx* ctx, const lfpsqp_spmat* SA, const lfpsqp_mat* A, const double* W_host, int m, const double* v, double* t_out);
int sp_factored_gemv_n(lfpsqp_ctx* ctx, const lfpsqp_spmat* SA, const lfpsqp_mat* A, const double* W_host, int m, double alpha, const double* t,
                       double beta, double* y);

// G = [S | X]' diag(w2) [S | X] from the nonzeros, exactly accumulated (sparse.hip); LFPSQP_ERR_UNSUPPORTED: form it on a dense copy
int sp_gram(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const lfpsqp_mat* X, int x0, int nx, const lfpsqp_vec* w2, double* scratch, double* G, int kmax);

int spmm(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const lfpsqp_mat* X, int x0, int nx, const double* W_dev, int ldw, int r, lfpsqp_mat* Out);

}  // namespace lfpsqp
