"""Device context and buffers (thin object layer over the C ABI)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import LfpsqpError, P, c_i64


class Context:
    """One GPU, one HIP stream, one communicator (lfpsqp_ctx)."""

    def __init__(self, device: int = 0, lib: _capi.Library | None = None):
        self.L = lib or _capi.load_library()
        h = P()
        rc = self.L.lfpsqp_ctx_create(device, C.byref(h))
        if rc != 0 or not h:
            raise LfpsqpError(f"lfpsqp_ctx_create(device={device}) failed (status {rc}): no usable MI355X/HIP device; "
                              "the product path has no CPU fallback")
        self.h = h
        self.rank, self.nranks = 0, 1
        self._cb_keep = None
        from .params import DeviceOptions
        self.options = DeviceOptions()          # device-only options (ls_batch, placement_tries): not part of LFPSQPParams

    # -- plumbing
    def check(self, rc: int):
        if rc != 0:
            raise LfpsqpError(f"status {rc}: {self.L.lfpsqp_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.lfpsqp_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        self.check(self.L.lfpsqp_ctx_sync(self.h))

    @property
    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self.check(self.L.lfpsqp_device_name(self.h, buf, 256))
        return buf.value.decode()

    def device_uuid(self) -> str:
        """"GPU-<16 hex digits>" of the device this context computes on (what rocminfo / rocm-smi print)."""
        buf = C.create_string_buffer(64)
        self.check(self.L.lfpsqp_device_uuid(self.h, buf, 64))
        return buf.value.decode()

    def timer_begin(self):
        self.check(self.L.lfpsqp_timer_begin(self.h))

    def timer_end(self) -> float:
        ms = C.c_double()
        self.check(self.L.lfpsqp_timer_end(self.h, C.byref(ms)))
        return ms.value

    def factored_basis_supported(self, A, SA=None) -> bool:
        """Can projcg_ run on a basis kept in factored form U = A W (no Z) on this context?  (lfpsqp_factored_basis_supported)"""
        yes = C.c_int(0)
        self.check(self.L.lfpsqp_factored_basis_supported(self.h, A.h, SA.h if SA is not None else None, C.byref(yes)))
        return bool(yes.value)

    def set_onepass(self, mode: int = 0):
        self.check(self.L.lfpsqp_ctx_set_onepass(self.h, int(mode)))

    def set_residual_buffers(self, mode: int = 0):
        """0 = the fused projected-CG iteration updates its residual in place (default), 1 = two buffers alternating."""
        self.check(self.L.lfpsqp_ctx_set_residual_buffers(self.h, int(mode)))

    def set_placement(self, tries: int = 3):
        """Candidate allocations tried by the placement-tuned allocators (lfpsqp_ctx_set_placement; 1 = off)."""
        self.check(self.L.lfpsqp_ctx_set_placement(self.h, int(tries)))
        self.options.placement_tries = int(tries)

    def set_nr_batch_mode(self, matrix_cores: bool = False):
        """Batched Newton retractions (lfpsqp_ctx_set_nr_batch_mode): False = the exact batch (bit for bit the one-by-one retractions, up to
        4 trials per pass), True = the matrix-core batch (up to 16 per pass, equal up to rounding)."""
        self.check(self.L.lfpsqp_ctx_set_nr_batch_mode(self.h, 1 if matrix_cores else 0))
        self.options.ls_batch_matrix_cores = bool(matrix_cores)

    def free_memory(self):
        """Free device memory in bytes (None when unknown): sizes the candidate count of placed allocations in bench.py."""
        try:
            import ctypes.util
            hip = C.CDLL("libamdhip64.so")
            f, t = C.c_size_t(), C.c_size_t()
            if hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0:
                return int(f.value)
        except OSError:
            pass
        return None

    def placement_info(self):
        """(candidates tried, index kept, fused-kernel ms per candidate) of the last placed allocation."""
        tries, pick = C.c_int(), C.c_int()
        ms = (C.c_double * 64)()
        self.check(self.L.lfpsqp_placement_info(self.h, C.byref(tries), C.byref(pick), ms, 64))
        return tries.value, pick.value, list(ms)[:tries.value]

    def _wrap_vecs(self, hh, n, count, stacked_N, hs):
        out = []
        for k in range(count):
            if stacked_N is not None:
                from .inequality import StackedVector
                v = StackedVector.__new__(StackedVector)
                v.N, v.hs = int(stacked_N), hs
            else:
                v = DeviceVector.__new__(DeviceVector)
            v.ctx, v.n, v.h = self, int(n), P(hh[k])
            out.append(v)
        return out

    def basis_and_vectors_placed(self, n: int, m: int, count: int, stacked_N: int | None = None):
        """The basis matrix (n x m) and ``count`` vectors streamed with it, allocated together by trial over every pair of candidate
        allocations (lfpsqp_basis_work_alloc_placed).  Returns (DeviceMatrix, [vectors])."""
        hs, nv = None, int(n)
        if stacked_N is not None:
            hs = int(self.L.lfpsqp_half_stride(int(stacked_N)))
            nv = hs + int(stacked_N)
        hh = (P * count)()
        mh = P()
        self.check(self.L.lfpsqp_basis_work_alloc_placed(self.h, int(n), int(m), nv, int(count), C.byref(mh), hh))
        M = DeviceMatrix.__new__(DeviceMatrix)
        M.ctx, M.n, M.m, M.h = self, int(n), int(m), mh
        return M, self._wrap_vecs(hh, nv, count, stacked_N, hs)

    def vectors_placed(self, M: "DeviceMatrix | None", n: int, count: int, ncols: int | None = None, stacked_N: int | None = None):
        """``count`` zero-filled n-vectors from ONE allocation, placement-tuned against the matrix M they will be streamed with
        (lfpsqp_vecs_alloc_placed).  ``stacked_N``: StackedVector objects of N + N entries (n is then ignored)."""
        hs = None
        if stacked_N is not None:
            hs = int(self.L.lfpsqp_half_stride(int(stacked_N)))
            n = hs + int(stacked_N)
        hh = (P * count)()
        self.check(self.L.lfpsqp_vecs_alloc_placed(self.h, M.h if M is not None else None, (M.m if ncols is None else ncols) if M is not None else 0,
                                                   int(n), int(count), hh))
        return self._wrap_vecs(hh, n, count, stacked_N, hs)

    def set_tuning(self, ks: int = 0, nt: bool = True):
        self.check(self.L.lfpsqp_ctx_set_tuning(self.h, int(ks), 1 if nt else 0))

    def set_profiling(self, on: bool):
        self.check(self.L.lfpsqp_ctx_set_profiling(self.h, 1 if on else 0))

    def profile_read(self):
        ms = (C.c_double * 8)()
        cnt = (c_i64 * 8)()
        self.check(self.L.lfpsqp_profile_read(self.h, ms, cnt))
        return list(ms), list(cnt)

    # -- sharding / communicator
    def shard_range(self, n: int, rank: int | None = None, nranks: int | None = None):
        r0, r1 = c_i64(), c_i64()
        rc = self.L.lfpsqp_shard_range(n, self.rank if rank is None else rank,
                                       self.nranks if nranks is None else nranks, C.byref(r0), C.byref(r1))
        if rc != 0:
            raise LfpsqpError("lfpsqp_shard_range: bad arguments")
        return r0.value, r1.value

    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self.check(self.L.lfpsqp_comm_unique_id(self.h, buf))
        return buf.raw

    def comm_init_rccl(self, rank: int, nranks: int, uid: bytes):
        buf = C.create_string_buffer(uid, 128)
        self.check(self.L.lfpsqp_comm_init_rccl(self.h, rank, nranks, buf))
        self.rank, self.nranks = rank, nranks

    def comm_p2p_export(self) -> bytes:
        """This rank's mailbox for the one-shot peer-to-peer all-reduce: its 64-byte IPC handle (lfpsqp_comm_p2p_export)."""
        buf = C.create_string_buffer(64)
        self.check(self.L.lfpsqp_comm_p2p_export(self.h, buf))
        return buf.raw

    def comm_p2p_allow_coarse(self, allow: bool = True):
        """Allow ordinary (coarse-grained) device memory for the mailbox when fine-grained memory cannot be exported (never silent)."""
        self.check(self.L.lfpsqp_comm_p2p_allow_coarse(self.h, 1 if allow else 0))

    def comm_p2p_info(self):
        """(memory kind of this rank's mailbox: 'fine-grained' | 'coarse-grained' | 'none', all-reduce launches so far)."""
        k, cnt = C.c_int(0), C.c_ulonglong(0)
        self.check(self.L.lfpsqp_comm_p2p_info(self.h, C.byref(k), C.byref(cnt)))
        return {0: "none", 1: "fine-grained", 2: "coarse-grained"}[k.value], cnt.value

    def comm_init_p2p(self, rank: int, nranks: int, handles):
        """``handles``: the 64-byte handles of all ranks, in rank order (lfpsqp_comm_init_p2p)."""
        blob = b"".join(handles)
        assert len(blob) == 64 * nranks
        self.check(self.L.lfpsqp_comm_init_p2p(self.h, rank, nranks, blob))
        self.rank, self.nranks = rank, nranks

    def comm_init_callback(self, rank: int, nranks: int, fn):
        """fn(ptr:int, count:int, op:int, stream:int) -> int; all-reduce `count` doubles at device pointer `ptr`."""
        def tramp(user, buf, count, op, stream):
            try:
                return int(fn(buf, count, op, stream) or 0)
            except Exception as e:  # never unwind through C
                print("all-reduce callback failed:", e)
                return 1
        cb = _capi.ALLREDUCE_FN(tramp)
        self._cb_keep = cb
        self.check(self.L.lfpsqp_comm_init_callback(self.h, rank, nranks, cb, None))
        self.rank, self.nranks = rank, nranks

    # -- buffers
    def vector(self, n: int, data=None) -> "DeviceVector":
        v = DeviceVector(self, n)
        if data is not None:
            v.upload(data)
        return v

    def matrix(self, n: int, m: int, data=None, placed: bool = False) -> "DeviceMatrix":
        M = DeviceMatrix(self, n, m, placed=placed)
        if data is not None:
            M.upload(data)
        return M


class DeviceVector:
    def __init__(self, ctx: Context, n: int):
        self.ctx, self.n = ctx, int(n)
        h = P()
        ctx.check(ctx.L.lfpsqp_vec_alloc(ctx.h, self.n, C.byref(h)))
        self.h = h

    def __len__(self):
        return self.n

    def free(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.L.lfpsqp_vec_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):   # device memory is library-owned; release it with the Python handle
        try:
            self.free()
        except Exception:
            pass

    def upload(self, data, offset: int = 0):
        a = np.ascontiguousarray(data, dtype=np.float64)
        self.ctx.check(self.ctx.L.lfpsqp_vec_upload(self.ctx.h, self.h, offset, a.ctypes.data, a.size))
        return self

    def download(self, count: int | None = None, offset: int = 0) -> np.ndarray:
        count = self.n - offset if count is None else count
        out = np.empty(count, dtype=np.float64)
        self.ctx.check(self.ctx.L.lfpsqp_vec_download(self.ctx.h, self.h, offset, out.ctypes.data, count))
        return out

    def fill(self, value: float):
        self.ctx.check(self.ctx.L.lfpsqp_vec_fill(self.ctx.h, self.h, float(value)))
        return self

    def copy_from(self, src: "DeviceVector"):
        self.ctx.check(self.ctx.L.lfpsqp_vec_copy(self.ctx.h, self.h, src.h))
        return self

    def copy_range_from(self, src: "DeviceVector", count: int, dst_off: int = 0, src_off: int = 0):
        self.ctx.check(self.ctx.L.lfpsqp_vec_copy_range(self.ctx.h, self.h, dst_off, src.h, src_off, count))
        return self

    def hash_fill(self, seed: int, offset: int = 0, scale: float = 1.0, shift: float = 0.0):
        self.ctx.check(self.ctx.L.lfpsqp_vec_hash_fill(self.ctx.h, self.h, seed, offset, scale, shift))
        return self


class DeviceMatrix:
    """Column-major n_loc x m matrix, every column contiguous (SURVEY §7 layout)."""

    def __init__(self, ctx: Context, n: int, m: int, placed: bool = False):
        """``placed``: allocate by trial (lfpsqp_mat_alloc_placed) -- for the long-lived matrices the hot loops stream."""
        self.ctx, self.n, self.m = ctx, int(n), int(m)
        h = P()
        ctx.check((ctx.L.lfpsqp_mat_alloc_placed if placed else ctx.L.lfpsqp_mat_alloc)(ctx.h, self.n, self.m, C.byref(h)))
        self.h = h

    @property
    def shape(self):
        return (self.n, self.m)

    def free(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.L.lfpsqp_mat_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, data, col0: int = 0):
        a = np.asfortranarray(data, dtype=np.float64)
        if a.ndim != 2 or a.shape[0] != self.n:
            raise ValueError("matrix upload: shape mismatch")
        self.ctx.check(self.ctx.L.lfpsqp_mat_upload(self.ctx.h, self.h, col0, a.shape[1], a.ctypes.data, max(a.shape[0], 1)))
        return self

    def download(self, col0: int = 0, ncols: int | None = None) -> np.ndarray:
        ncols = self.m - col0 if ncols is None else ncols
        if getattr(self, "is_view", False):                # a view: materialise it (lfpsqp_mat_copy), then read that back
            tmp = DeviceMatrix(self.ctx, self.n, self.m)
            try:
                return tmp.copy_from(self).download(col0, ncols)
            finally:
                tmp.free()
        out = np.empty((self.n, ncols), dtype=np.float64, order='F')
        self.ctx.check(self.ctx.L.lfpsqp_mat_download(self.ctx.h, self.h, col0, ncols, out.ctypes.data, max(self.n, 1)))
        return out

    def copy_from(self, src: "DeviceMatrix"):
        """(the copy of a row-scaled view is the matrix it stands for)"""
        self.ctx.check(self.ctx.L.lfpsqp_mat_copy(self.ctx.h, self.h, src.h))
        return self

    def view(self, rs: "DeviceVector | None" = None, u: "DeviceVector | None" = None, w: "DeviceVector | None" = None) -> "DeviceMatrix":
        """diag(rs) * self + u w' without a copy (lfpsqp_mat_view): every product kernel of the library accepts it wherever a matrix is only
        read; storage and vectors are borrowed (the view keeps them alive)."""
        v = DeviceMatrix.__new__(DeviceMatrix)
        v.ctx, v.n, v.m = self.ctx, self.n, self.m
        h = P()
        g = lambda x: x.h if x is not None else None
        self.ctx.check(self.ctx.L.lfpsqp_mat_view(self.ctx.h, self.h, g(rs), g(u), g(w), C.byref(h)))
        v.h = h
        v.base, v.rs, v.ru, v.rw = self, rs, u, w
        v.is_view = True
        return v

    def rowscaled_view(self, rs: "DeviceVector") -> "DeviceMatrix":
        return self.view(rs=rs)

    def hash_fill(self, seed: int, row0: int = 0, n_global: int | None = None, scale: float = 1.0, nrows: int | None = None,
                  ncols: int | None = None):
        nrows = self.n if nrows is None else nrows
        ncols = self.m if ncols is None else ncols
        self.ctx.check(self.ctx.L.lfpsqp_mat_hash_fill(self.ctx.h, self.h, seed, row0, nrows if n_global is None else n_global,
                                                       float(scale), nrows, ncols))
        return self


# ---- BLAS-1/2 primitives (mirror of the reference's mul!/dot/norm/axpy! call sites) -------

def gemv_t(M: DeviceMatrix, v: DeviceVector, t: DeviceVector, ncols: int | None = None):
    """t[:ncols] = M[:, :ncols]' v  (kgemv!('T', rank, ...), src/la_helper.jl:36-44)."""
    c = M.ctx
    c.check(c.L.lfpsqp_gemv_t(c.h, M.h, M.m if ncols is None else ncols, v.h, t.h))
    return t


def gemv_n(M: DeviceMatrix, t: DeviceVector, y: DeviceVector, alpha=1.0, beta=0.0, ncols: int | None = None):
    """y = alpha M[:, :ncols] t + beta y  (kgemv!('N', ...) / mul!(y, U, t, alpha, beta))."""
    c = M.ctx
    c.check(c.L.lfpsqp_gemv_n(c.h, M.h, M.m if ncols is None else ncols, float(alpha), t.h, float(beta), y.h))
    return y


def dot(x: DeviceVector, y: DeviceVector) -> float:
    out = C.c_double()
    x.ctx.check(x.ctx.L.lfpsqp_dot(x.ctx.h, x.h, y.h, C.byref(out)))
    return out.value


def nrm2(x: DeviceVector) -> float:
    out = C.c_double()
    x.ctx.check(x.ctx.L.lfpsqp_nrm2(x.ctx.h, x.h, C.byref(out)))
    return out.value


def nrm2_head(x: DeviceVector, count: int) -> float:
    """norm(view(x, 1:count))."""
    out = C.c_double()
    x.ctx.check(x.ctx.L.lfpsqp_dot_head(x.ctx.h, x.h, x.h, int(count), C.byref(out)))
    return float(np.sqrt(out.value))


def amax(x: DeviceVector) -> float:
    out = C.c_double()
    x.ctx.check(x.ctx.L.lfpsqp_amax(x.ctx.h, x.h, C.byref(out)))
    return out.value


def axpby(a: float, x: DeviceVector, b: float, y: DeviceVector):
    """y = a x + b y."""
    x.ctx.check(x.ctx.L.lfpsqp_axpby(x.ctx.h, float(a), x.h, float(b), y.h))
    return y


def waxpby(a: float, x: DeviceVector, b: float, y: DeviceVector, z: DeviceVector):
    x.ctx.check(x.ctx.L.lfpsqp_waxpby(x.ctx.h, float(a), x.h, float(b), y.h, z.h))
    return z


def vmul(d: DeviceVector, x: DeviceVector, y: DeviceVector):
    x.ctx.check(x.ctx.L.lfpsqp_vmul(x.ctx.h, d.h, x.h, y.h))
    return y


class SparseMatrix:
    """Sparse n_loc x m constraint-gradient matrix with a few nonzeros per row (lfpsqp_spmat): built from triplets
    (0-based local row, column, value) or from a scipy.sparse matrix of shape (n, m)."""

    def __init__(self, ctx: Context, n: int, m: int, rows, cols, vals):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        cols = np.ascontiguousarray(cols, dtype=np.int64)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        assert rows.shape == cols.shape == vals.shape
        self.ctx, self.n, self.m = ctx, int(n), int(m)
        h = P()
        ctx.check(ctx.L.lfpsqp_spmat_create(ctx.h, self.n, self.m, rows.size, rows.ctypes.data, cols.ctypes.data, vals.ctypes.data, C.byref(h)))
        self.h = h
        nn, mm, nnz, k = _capi.c_i64(), _capi.c_i64(), _capi.c_i64(), _capi.c_i64()
        ctx.L.lfpsqp_spmat_info(self.h, C.byref(nn), C.byref(mm), C.byref(nnz), C.byref(k))
        self.nnz, self.ell_width = nnz.value, k.value

    @classmethod
    def from_scipy(cls, ctx: Context, A):
        A = A.tocoo()
        return cls(ctx, A.shape[0], A.shape[1], A.row, A.col, A.data)

    def free(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.L.lfpsqp_spmat_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def clone(self) -> "SparseMatrix":
        """A second object with this one's STRUCTURE (shared on the device: keep the original alive) and its own values
        (lfpsqp_spmat_clone) -- the x-dependent constraint gradients of ElementwiseConstraints."""
        c = object.__new__(SparseMatrix)
        c.ctx, c.n, c.m, c.nnz, c.ell_width = self.ctx, self.n, self.m, self.nnz, self.ell_width
        c._structure_of = self
        h = P()
        self.ctx.check(self.ctx.L.lfpsqp_spmat_clone(self.ctx.h, self.h, C.byref(h)))
        c.h = h
        return c

    def rowscale_from(self, src: "SparseMatrix", v: "DeviceVector"):
        """values = diag(v) * src.values (lfpsqp_spmat_rowscale)."""
        self.ctx.check(self.ctx.L.lfpsqp_spmat_rowscale(self.ctx.h, self.h, src.h, v.h))
        return self

    def to_dense(self, M: "DeviceMatrix | None" = None) -> "DeviceMatrix":
        M = M if M is not None else DeviceMatrix(self.ctx, self.n, self.m)
        self.ctx.check(self.ctx.L.lfpsqp_spmat_to_dense(self.ctx.h, self.h, M.h))
        return M

    def gram(self, Jct: "DeviceMatrix | None" = None, w2: "DeviceVector | None" = None) -> np.ndarray:
        """[S | Jct[:, m:]]' diag(w2) [S | Jct[:, m:]] from the nonzeros, exactly accumulated (lfpsqp_spmat_gram)."""
        M = Jct.m if Jct is not None else self.m
        G = np.zeros((M, M), order="F")
        self.ctx.check(self.ctx.L.lfpsqp_spmat_gram(self.ctx.h, self.h, Jct.h if Jct is not None else None, w2.h if w2 is not None else None,
                                                    G.ctypes.data))
        return G


def spmv_t(S: SparseMatrix, v: DeviceVector, t: DeviceVector) -> DeviceVector:
    """t = S' v (all-reduced)."""
    S.ctx.check(S.ctx.L.lfpsqp_spmv_t(S.ctx.h, S.h, v.h, t.h))
    return t


def spmv_n(S: SparseMatrix, t: DeviceVector, y: DeviceVector, alpha: float = 1.0, beta: float = 0.0) -> DeviceVector:
    """y = alpha S t + beta y."""
    S.ctx.check(S.ctx.L.lfpsqp_spmv_n(S.ctx.h, S.h, float(alpha), t.h, float(beta), y.h))
    return y
