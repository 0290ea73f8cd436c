// Sparse constraint gradients: the device object and the row accessor shared by sparse.hip and the solvers that fuse a
// sparse product into their row functors (pcg.hip).
#pragma once
#include "internal.h"

struct lfpsqp_spmat {
    int64_t n = 0, m = 0, nnz = 0;
    int K = 0;              // ELL width
    int64_t ld = 0;         // rows rounded up to whole tiles
    double* ell_val = nullptr;    // [K][ld]
    int32_t* ell_col = nullptr;   // [K][ld]
    int32_t* csc_row = nullptr;   // [nnz], ascending inside a column
    double* csc_val = nullptr;    // [nnz]
    int64_t nchunks = 0;
    int64_t* colptr = nullptr;      // [m + 1] (device): nonzero range of a column
    int64_t* chunk_beg = nullptr;   // [nchunks + 1] (device): nonzero range of a chunk
    int32_t* col_chunk = nullptr;   // [m + 1] (device): chunk range of a column
};

namespace lfpsqp {

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
int spmm(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const lfpsqp_mat* X, int x0, int nx, const double* W_dev, int ldw, int r, lfpsqp_mat* Out);

}  // namespace lfpsqp
