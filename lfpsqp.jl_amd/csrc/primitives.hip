// BLAS-1/2 primitives of the C ABI on the tall-skinny layout, and the synthetic
// input generators.  Replaces the reference's mul!/gemv!/kgemv!/dot/norm/axpy!
// call sites one-for-one (see include/lfpsqp_hip.h); the fused solvers in
// projcg.hip / retract.hip reuse the same kernels with richer functors.
#include "internal.h"

namespace lfpsqp {

// ---- GEMV-T producers / GEMV-N consumers ------------------------------------
struct PlainV {  // v straight from memory
    const double* v;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        const double2 a = ld2(v + r);
        return make_double2(v0 ? a.x : 0.0, v1 ? a.y : 0.0);
    }
};

#ifndef LFPSQP_GEMVN_ONEPASS
#define LFPSQP_GEMVN_ONEPASS 1
#endif
struct AxpbyEpi {  // y = alpha*acc + beta*y   (beta == 0 never reads y: BLAS semantics)
    double* y;
    double alpha, beta;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t r, double2 acc, bool v0, bool v1, double*) const {
        double2 o;
        if (beta == 0.0) {
            o = make_double2(alpha * acc.x, alpha * acc.y);
        } else {
            const double2 yy = ld2(y + r);
            o = make_double2(fma(alpha, acc.x, beta * yy.x), fma(alpha, acc.y, beta * yy.y));
        }
        if (v1) st2(y + r, o);
        else if (v0) y[r] = o.x;
    }
};

// The same update as a row functor of the one-pass kernel (kernels.h): the persistent grid stores y in a few device-wide bursts instead of a
// continuous trickle inside the matrix read stream (FINDINGS.md 6 "The thin store stream"); the kernel's second product is fed zeros and its
// sums are discarded.
struct GemvNRow {
    double* y;
    double alpha, beta;
    using Uni = NoUni;
    struct Row { double yy; };
    static constexpr bool kSplitRed = false;
    static constexpr int kStageStreams = 1;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ Uni uniform() const { return Uni{}; }
    __device__ __forceinline__ Row fetch(uint32_t o) const {
        return Row{beta == 0.0 ? 0.0 : *reinterpret_cast<const double*>(reinterpret_cast<const char*>(y) + o)};
    }
    __device__ __forceinline__ double* stage_out(int) const { return y; }
    __device__ __forceinline__ void apply(int64_t row, uint32_t o, const double (&acc)[1], bool valid, bool owner, bool lead, const Uni& u, const Row& w,
                                          double (&v)[1], double (&red)[1]) const {
        apply_staged(row, o, acc, valid, owner, lead, u, w, v, red, nullptr, 0);
    }
    __device__ __forceinline__ void apply_staged(int64_t, uint32_t o, const double (&acc)[1], bool valid, bool owner, bool, const Uni&, const Row& w,
                                                 double (&v)[1], double (&)[1], double* slot, int) const {
        const double out = beta == 0.0 ? alpha * acc[0] : fma(alpha, acc[0], beta * w.yy);       // (the expressions of AxpbyEpi)
        if (valid && owner) {
            if (slot) *slot = out;
            else *reinterpret_cast<double*>(reinterpret_cast<char*>(y) + o) = out;
        }
        v[0] = 0.0;
    }
};

// ---- elementwise functors ------------------------------------------------------
struct DotF {
    const double *x, *y;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        const double2 a = ld2(x + i), b = ld2(y + i);
        double s = 0.0;
        if (v0) s = a.x * b.x;
        if (v1) s = fma(a.y, b.y, s);
        red[0] += s;
    }
};
struct AmaxF {
    const double* x;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        const double2 a = ld2(x + i);
        if (v0) red[0] = nanmax(red[0], fabs(a.x));
        if (v1) red[0] = nanmax(red[0], fabs(a.y));
    }
};
struct CopyF {  // y = x
    const double* x;
    double* y;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double2 xx = ld2(x + i);
        if (v1) st2(y + i, xx);
        else if (v0) y[i] = xx.x;
    }
};
struct WaxpbyF {  // z = a*x + b*y
    double a, b;
    const double *x, *y;
    double* z;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        double2 o;
        if (b == 0.0) {
            const double2 xx = ld2(x + i);
            o = make_double2(a * xx.x, a * xx.y);
        } else if (a == 0.0) {
            const double2 yy = ld2(y + i);
            o = make_double2(b * yy.x, b * yy.y);
        } else {
            const double2 xx = ld2(x + i), yy = ld2(y + i);
            o = make_double2(fma(a, xx.x, b * yy.x), fma(a, xx.y, b * yy.y));
        }
        if (v1) st2(z + i, o);
        else if (v0) z[i] = o.x;
    }
};
struct VmulF {  // y = d .* x
    const double *d, *x;
    double* y;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double2 dd = ld2(d + i), xx = ld2(x + i);
        const double2 o = make_double2(dd.x * xx.x, dd.y * xx.y);
        if (v1) st2(y + i, o);
        else if (v0) y[i] = o.x;
    }
};
struct FillF {
    double* x;
    double value;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        if (v1) st2(x + i, make_double2(value, value));
        else if (v0) x[i] = value;
    }
};
struct FillRangeF {
    double* x;
    int64_t lo, hi;
    double value;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        if (v0 && i >= lo && i < hi) x[i] = value;
        if (v1 && i + 1 >= lo && i + 1 < hi) x[i + 1] = value;
    }
};
struct AffineHeadF {
    double a, c;
    const double* x;
    double* y;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double2 xx = ld2(x + i);
        if (v1) st2(y + i, make_double2(fma(a, xx.x, c), fma(a, xx.y, c)));
        else if (v0) y[i] = fma(a, xx.x, c);
    }
};
struct SumSqShiftF {
    const double* x;
    double c;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        const double2 a = ld2(x + i);
        double s = 0.0;
        if (v0) s = (a.x - c) * (a.x - c);
        if (v1) s = fma(a.y - c, a.y - c, s);
        red[0] += s;
    }
};
struct HashVecF {  // x[i] = scale*u(seed, offset+i) + shift
    double* x;
    uint64_t seed;
    int64_t offset;
    double scale, shift;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double a = fma(scale, hash_u(seed, (uint64_t)(offset + i)), shift);
        const double b = fma(scale, hash_u(seed, (uint64_t)(offset + i + 1)), shift);
        if (v1) st2(x + i, make_double2(a, b));
        else if (v0) x[i] = a;
    }
};

// M[i, j] = u(seed, j*n_global + row0 + i): one workgroup per (row tile, column)
__global__ __launch_bounds__(kThreads) void hash_mat_kernel(double* M, int64_t ld, int64_t n, uint64_t seed, int64_t row0,
                                                             int64_t n_global, double scale) {
    const int64_t j = blockIdx.y;
    double* col = M + j * ld;
    const uint64_t base = (uint64_t)j * (uint64_t)n_global + (uint64_t)row0;
#pragma unroll
    for (int s = 0; s < kMaxKS; ++s) {
        const int64_t r = (int64_t)blockIdx.x * kPadRows + (int64_t)s * kSlabRows + (int64_t)threadIdx.x * 2;
        if (r + 1 < n) st2(col + r, make_double2(scale * hash_u(seed, base + (uint64_t)r), scale * hash_u(seed, base + (uint64_t)r + 1)));
        else if (r < n) col[r] = scale * hash_u(seed, base + (uint64_t)r);
    }
}

// ---- separable objectives f(x) = sum_i phi(x_i; a_i, c_i) of the device-resident problem classes -----------
//   kind 0: a (x-c)^2            kind 1: a (x-c)^4 + (x-c)^2            kind 2: a (sqrt(1 + (x-c)^2) - 1)   (pseudo-Huber)
// mode 0: partial sums of phi; 1: out = phi'(x) ; 2: out = phi''(x) -- elementwise, no transcendental functions (the
// same bits as a numpy evaluation up to the rounding of sqrt and the summation order).
__device__ __forceinline__ double sep_eval(int kind, int mode, double a, double t) {   // t = x - c
    if (kind == 0) return mode == 0 ? a * t * t : (mode == 1 ? 2.0 * a * t : 2.0 * a);
    if (kind == 1) {
        const double t2 = t * t;
        return mode == 0 ? fma(a * t2, t2, t2) : (mode == 1 ? fma(4.0 * a * t2, t, 2.0 * t) : fma(12.0 * a, t2, 2.0));
    }
    const double s = sqrt(fma(t, t, 1.0));
    return mode == 0 ? a * (s - 1.0) : (mode == 1 ? a * t / s : a / (s * s * s));
}
struct SepF {
    int kind, mode;
    const double *a, *c, *x;   // a, c: per-variable parameters (may be null: a = a0, c = c0)
    double a0, c0;
    double* out;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        const double2 xx = ld2(x + i);
        const double2 aa = a ? ld2(a + i) : make_double2(a0, a0), cc = c ? ld2(c + i) : make_double2(c0, c0);
        const double r0 = v0 ? sep_eval(kind, mode, aa.x, xx.x - cc.x) : 0.0, r1 = v1 ? sep_eval(kind, mode, aa.y, xx.y - cc.y) : 0.0;
        if (mode == 0) {
            red[0] += r0 + r1;
        } else {
            if (v1) st2(out + i, make_double2(r0, r1));
            else if (v0) out[i] = r0;
        }
    }
};

}  // namespace lfpsqp

using namespace lfpsqp;

extern "C" {

int lfpsqp_vec_fill(lfpsqp_ctx* ctx, lfpsqp_vec* v, double value) {
    LF_ARG(ctx, ctx && v);
    if (v->n == 0) return 0;
    return run_vec<FillF, 0, NoPost>(ctx, v->n, FillF{v->p, value}, 0u, nullptr, NoPost());
}

int lfpsqp_vec_copy(lfpsqp_ctx* ctx, lfpsqp_vec* dst, const lfpsqp_vec* src) {
    LF_ARG(ctx, ctx && dst && src && dst->n == src->n);
    if (src->n == 0 || dst->p == src->p) return 0;
    // (the runtime's device-to-device copy streams at 4.97 TB/s on MI355X, one tile per block of the library's own kernel at 6.24:
    // tools/micro/vecprobe.hip)
    return run_vec<CopyF, 0, NoPost>(ctx, src->n, CopyF{src->p, dst->p}, 0u, nullptr, NoPost());
}

int lfpsqp_vec_copy_range(lfpsqp_ctx* ctx, lfpsqp_vec* dst, int64_t dst_off, const lfpsqp_vec* src, int64_t src_off, int64_t count) {
    LF_ARG(ctx, ctx && dst && src && count >= 0 && dst_off >= 0 && src_off >= 0 && dst_off + count <= dst->n && src_off + count <= src->n);
    if (count == 0) return 0;
    LF_HIP(ctx, hipMemcpyAsync(dst->p + dst_off, src->p + src_off, sizeof(double) * count, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

int lfpsqp_vec_hash_fill(lfpsqp_ctx* ctx, lfpsqp_vec* v, uint64_t seed, int64_t offset, double scale, double shift) {
    LF_ARG(ctx, ctx && v);
    if (v->n == 0) return 0;
    return run_vec<HashVecF, 0, NoPost>(ctx, v->n, HashVecF{v->p, seed, offset, scale, shift}, 0u, nullptr, NoPost());
}

int lfpsqp_mat_hash_fill(lfpsqp_ctx* ctx, lfpsqp_mat* M, uint64_t seed, int64_t row0, int64_t n_global, double scale, int64_t nrows,
                         int64_t ncols) {
    LF_ARG(ctx, ctx && plain_mat(M) && row0 >= 0 && nrows >= 0 && nrows <= M->n && ncols >= 0 && ncols <= M->m && n_global >= nrows);
    if (nrows == 0 || ncols == 0) return 0;
    LF_ARG(ctx, ncols <= 65535);
    hipLaunchKernelGGL(hash_mat_kernel, dim3((unsigned)((nrows + kPadRows - 1) / kPadRows), (unsigned)ncols), dim3(kThreads), 0, ctx->stream, M->p, M->ld,
                       nrows, seed, row0, n_global, scale);
    LF_LAUNCH_CHECK(ctx);
    return 0;
}

int lfpsqp_gemv_t(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols, const lfpsqp_vec* v, lfpsqp_vec* t) {
    LF_RANGE("lfpsqp_gemv_t");
    LF_ARG(ctx, ctx && M && v && t && ncols >= 0 && ncols <= M->m && v->n == M->n && t->n >= ncols);
    return run_gemv_t(ctx, M, (int)ncols, M->n, PlainV{v->p}, t->p);
}

int lfpsqp_gemv_n(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols, double alpha, const lfpsqp_vec* t, double beta, lfpsqp_vec* y) {
    LF_RANGE("lfpsqp_gemv_n");
    LF_ARG(ctx, ctx && M && y && ncols >= 0 && ncols <= M->m && y->n == M->n && (ncols == 0 || (t && t->n >= ncols)));
    // 97 .. 132 columns: through the one-pass kernel's persistent grid and staged stores (GemvNRow) -- same buffers, both forms
    // (tools/gpu_gemvn_ab.sh, n = 1e7): m = 128 1.590 / 1.604 ms (min / mean over 12 placements) against 1.617 / 1.658; at m = 32, 64 and
    // 512 the two-pass form is the faster one by 1-6 % (the one-pass kernel's idle second product costs more than its store pattern saves)
    if (LFPSQP_GEMVN_ONEPASS && ncols >= 97 && ncols <= 132 && onepass_cw(ctx, (int)ncols, M->ld, M->n) != 0)
        return run_onepass<GemvNRow, 1, 1>(ctx, M, (int)ncols, 0, M->n, t->p, GemvNRow{y->p, alpha, beta}, nullptr, -1, 0, 0, true);
    return run_gemv_n<AxpbyEpi, 0, NoPost>(ctx, M, (int)ncols, M->n, t ? t->p : nullptr, AxpbyEpi{y->p, alpha, beta}, nullptr, NoPost());
}

int lfpsqp_dot(lfpsqp_ctx* ctx, const lfpsqp_vec* x, const lfpsqp_vec* y, double* out) {
    LF_ARG(ctx, ctx && x && y && out && x->n == y->n);
    LF_TRY((run_vec<DotF, 1, NoPost>(ctx, x->n, DotF{x->p, y->p}, 0u, ctx->scal + 32, NoPost())));
    return read_back(ctx, ctx->scal + 32, out, 1);
}

int lfpsqp_dot_head(lfpsqp_ctx* ctx, const lfpsqp_vec* x, const lfpsqp_vec* y, int64_t count, double* out) {
    LF_ARG(ctx, ctx && x && y && out && count >= 0 && count <= x->n && count <= y->n);
    LF_TRY((run_vec<DotF, 1, NoPost>(ctx, count, DotF{x->p, y->p}, 0u, ctx->scal + 32, NoPost())));
    return read_back(ctx, ctx->scal + 32, out, 1);
}

int lfpsqp_nrm2(lfpsqp_ctx* ctx, const lfpsqp_vec* x, double* out) {
    LF_ARG(ctx, ctx && x && out);
    double s = 0.0;
    LF_TRY(lfpsqp_dot(ctx, x, x, &s));
    *out = sqrt(s);
    return 0;
}

int lfpsqp_amax(lfpsqp_ctx* ctx, const lfpsqp_vec* x, double* out) {
    LF_ARG(ctx, ctx && x && out);
    LF_TRY((run_vec<AmaxF, 1, NoPost>(ctx, x->n, AmaxF{x->p}, 1u, ctx->scal + 32, NoPost())));
    return read_back(ctx, ctx->scal + 32, out, 1);
}

int lfpsqp_waxpby(lfpsqp_ctx* ctx, double a, const lfpsqp_vec* x, double b, const lfpsqp_vec* y, lfpsqp_vec* z) {
    LF_ARG(ctx, ctx && x && y && z && x->n == y->n && x->n == z->n);
    if (z->n == 0) return 0;
    return run_vec<WaxpbyF, 0, NoPost>(ctx, z->n, WaxpbyF{a, b, x->p, y->p, z->p}, 0u, nullptr, NoPost());
}

int lfpsqp_axpby(lfpsqp_ctx* ctx, double a, const lfpsqp_vec* x, double b, lfpsqp_vec* y) {
    return lfpsqp_waxpby(ctx, a, x, b, y, y);
}

int lfpsqp_vmul(lfpsqp_ctx* ctx, const lfpsqp_vec* d, const lfpsqp_vec* x, lfpsqp_vec* y) {
    LF_ARG(ctx, ctx && d && x && y && d->n == x->n && x->n == y->n);
    if (y->n == 0) return 0;
    return run_vec<VmulF, 0, NoPost>(ctx, y->n, VmulF{d->p, x->p, y->p}, 0u, nullptr, NoPost());
}

int lfpsqp_vec_fill_range(lfpsqp_ctx* ctx, lfpsqp_vec* v, int64_t offset, int64_t count, double value) {
    LF_ARG(ctx, ctx && v && offset >= 0 && count >= 0 && offset + count <= v->n);
    if (count == 0) return 0;
    return run_vec<FillRangeF, 0, NoPost>(ctx, offset + count, FillRangeF{v->p, offset, offset + count, value}, 0u, nullptr, NoPost());
}

int lfpsqp_affine_head(lfpsqp_ctx* ctx, double a, const lfpsqp_vec* x, double c, int64_t count, lfpsqp_vec* y) {
    LF_ARG(ctx, ctx && x && y && count >= 0 && count <= x->n && count <= y->n);
    if (count == 0) return 0;
    return run_vec<AffineHeadF, 0, NoPost>(ctx, count, AffineHeadF{a, c, x->p, y->p}, 0u, nullptr, NoPost());
}

int lfpsqp_sumsq_shift(lfpsqp_ctx* ctx, const lfpsqp_vec* x, int64_t count, double c, double* out) {
    LF_ARG(ctx, ctx && x && out && count >= 0 && count <= x->n);
    LF_TRY((run_vec<SumSqShiftF, 1, NoPost>(ctx, count, SumSqShiftF{x->p, c}, 0u, ctx->scal + 32, NoPost())));
    return read_back(ctx, ctx->scal + 32, out, 1);
}

int lfpsqp_allreduce(lfpsqp_ctx* ctx, lfpsqp_vec* v, int64_t count) {
    LF_ARG(ctx, ctx && v && count >= 0 && count <= v->n);
    return allreduce_dev(ctx, v->p, count, 0);
}

int lfpsqp_separable(lfpsqp_ctx* ctx, int kind, int mode, const lfpsqp_vec* a, double a0, const lfpsqp_vec* c, double c0, const lfpsqp_vec* x,
                     int64_t count, lfpsqp_vec* out_vec, double* out_sum) {
    LF_ARG(ctx, ctx && x && kind >= 0 && kind <= 2 && mode >= 0 && mode <= 2 && count >= 0 && count <= x->n && (!a || a->n >= count) &&
                    (!c || c->n >= count) && (mode == 0 ? out_sum != nullptr : (out_vec && out_vec->n >= count)));
    const SepF f{kind, mode, a ? a->p : nullptr, c ? c->p : nullptr, x->p, a0, c0, out_vec ? out_vec->p : nullptr};
    if (mode == 0) {
        LF_TRY((run_vec<SepF, 1, NoPost>(ctx, count, f, 0u, ctx->scal + 32, NoPost())));
        return read_back(ctx, ctx->scal + 32, out_sum, 1);
    }
    if (count == 0) return 0;
    return run_vec<SepF, 0, NoPost>(ctx, count, f, 0u, nullptr, NoPost());
}

}  // extern "C"
