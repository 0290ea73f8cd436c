// Batched Newton-retraction step on the matrix cores: up to 16 trial points of one linesearch (src/linesearch.jl:49-60: the steps
// alpha, alpha*s, alpha*s^2, ... whose retractions fail one after another) advance together, ONE pass over Jct per Newton step
// (src/retractions.jl:140-149) for all of them.  Included by retract.hip after NRStepE.
//
// Both products of a step are genuine contractions once the trials are stacked:
//   first   Y'[trial, row] = (W ddelta_trial)[col] * Jct[row, col]       16 trials x 16 rows x m columns per wave tile
//   second  C [col, trial] = Jct[row, col] * v_trial[row]                m columns x 16 trials x 16 rows
// v_mfma_f64_16x16x4_f64 (lane l holds A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D[i = (l >> 4) + 4 reg][j = l & 15]) takes
// the tile exactly as the one-pass kernels load it -- lane (row = l & 15, kq = l >> 4), register c = column 4c + kq: 16 rows of one
// 128-byte line per column -- as the B operand of the first product, with the trials' coefficients (LDS, ts[column][trial]) as A.  A
// lane then owns ONE row and the four trials kq, kq + 4, kq + 8, kq + 12: the row's shared inputs (old point, bound data) are loaded
// once for its trials, and the per-trial iterates are read and written as 128-byte segments of 16 consecutive rows.
//   The second product contracts over the ROWS, which sit in the low lane bits of the loaded tile, while the instruction contracts over
// the high ones: the tile is transposed through LDS, 64 columns at a time (per-wave staging area, XOR-swizzled so that the column-wise
// writes and the row-wise reads are both conflict-free; LDS operations of one wave execute in order, so no barrier is involved), and the
// new iterates' values go through a 16 x 16 LDS transpose into the B operand.  The accumulators C[16 columns x 16 trials] cost 8
// registers per 16-column block (the VALU form kept 4 x 32 running sums per lane for FOUR trials, 248 registers).  Arithmetic: 65
// MFMAs of 64 cycles per 16-row tile at m = 128 against ~6200 cycles of HBM time per tile and SIMD -- the pass stays HBM-bound.
#pragma once

namespace lfpsqp {

constexpr int kNRBW = 16;                 // trials per pass (the N dimension of the instruction)
constexpr int kNRBChunk = 64;             // tile columns transposed through LDS at a time
typedef double nrb_f64x4 __attribute__((vector_size(32)));

// (writes of one wave to LDS followed by reads of other lanes' words: the hardware executes a wave's LDS instructions in order; the
// compiler must not reorder them, and the emulator's lanes -- fibers -- must all have written)
// keep the instruction scheduler from interleaving what stands on either side (the four trials' row updates, each with a dozen
// temporaries of its square roots and divisions, would otherwise run side by side and spill)
__device__ __forceinline__ void sched_fence() {
#ifndef LFPSQP_HIP_EMULATED
    __builtin_amdgcn_sched_barrier(0);
#endif
}
__device__ __forceinline__ void wave_lds_fence() {
#ifdef LFPSQP_HIP_EMULATED
    hipemu::wave_sync();
#else
    asm volatile("" ::: "memory");
#endif
}

// The trials' iterates are reached through pointers that come out of an LDS table: left generic, the compiler addresses them with flat_*
// instructions, whose completions are out of order with respect to everything else -- every wait then becomes vmcnt(0) lgkmcnt(0).  As
// GLOBAL pointers their loads and stores count in issue order with the tile's buffer loads, and the waits are exact.
#ifdef LFPSQP_HIP_EMULATED
typedef double* nrb_gptr;
#else
typedef __attribute__((address_space(1))) double* nrb_gptr;
#endif

struct NRBatchArgs {
    NRStepE e;                            // shared row fields (e.xnew unused)
    double* xnew[kNRBW];                  // the trials' iterates
    const int64_t* ist;                   // status words: ist[4 * b] != 0 => trial b has finished (keeps its iterate, contributes nothing)
    const int64_t* all;                   // every trial finished: the launch is a no-op
    int nb;
};

// Transposed-tile staging area: element (row, col) of a 16 x 64 chunk lives at word row * 66 + col.  The odd-ish row stride makes the
// column-wise writes conflict-free (word = (66 r16 + kq) + 4c: lanes differ by 2 r16 + kq mod 32) and leaves the row-wise reads
// ((66 kq + r16) + 66 * 4ks + 16 cb) with 2-way conflicts -- but BOTH address one lane-dependent register plus an immediate offset.  An
// XOR swizzle is conflict-free on both sides and costs a register per (k-step, block) address: 25 registers this kernel does not have
// (the bound-constrained form spilled 30, and a spilled load forces a full vmcnt(0) drain where it is reloaded).
constexpr int kNRBLdr = kNRBChunk + 2;

// CPL: column groups (of 4) of the first product held per lane, ncN <= 4 * CPL; NBLK: 16-column blocks of the second product, ncT <= 16 * NBLK.
// part: one row PER WAVE: [trial * ncT + col] (16 x ncT), then the 16 ball partials.
template <bool ST, int CPL, int NBLK>
__global__ __launch_bounds__(kThreads, 2) void nrb_mfma_kernel(const double* __restrict__ M, int64_t ld, int ncN, int ncT, int64_t n, int64_t rounds,
                                                               const double* __restrict__ t, int t_stride, NRBatchArgs ep, double* __restrict__ part,
                                                               int part_ld) {
    if (ld_stat(ep.all) != 0) return;
    constexpr int NCH = (NBLK * 16 + kNRBChunk - 1) / kNRBChunk;
    constexpr int kStep = 16 * kWaves;
    __shared__ double ts[CPL * 4][kNRBW];
    __shared__ double tj[kWaves][16 * kNRBLdr];
    __shared__ double vs[kWaves][16 * 17];
    __shared__ double* xp[kNRBW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int glast = (ncN - 1) / 4, lastc0 = ncN - 4;
    for (int j = threadIdx.x; j < CPL * 4 * kNRBW; j += kThreads) {
        const int slot = j >> 4, b = j & 15, g = slot >> 2, hh = slot & 3;
        double v = 0.0;
        if (b < ep.nb) {
            const double* tb = t + (int64_t)b * t_stride;
            if (g < glast) v = ld_scal(tb + slot);
            else if (g == glast && lastc0 + hh >= glast * 4) v = ld_scal(tb + lastc0 + hh);
        }
        ts[slot][b] = v;
    }
    unsigned active = 0u;                                     // uniform: the status words change between launches only
    for (int b = 0; b < ep.nb; ++b)
        if (ld_stat(ep.ist + 4 * b) == 0) active |= 1u << b;
    // the (unconditional, see fetch) loads of a finished or missing trial go to a RUNNING trial's iterate: lines that are being read anyway,
    // instead of 16 n more bytes per row from HBM for every trial that has converged or been retired
    if (threadIdx.x < kNRBW) {
        const int b = (int)threadIdx.x;
        const int first = active ? __builtin_ctz(active) : 0;
        xp[b] = ep.xnew[(b < ep.nb && ((active >> b) & 1u)) ? b : first];
    }
    __syncthreads();
    nrb_gptr xpv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) xpv[v] = (nrb_gptr)xp[kq + 4 * v];
    const unsigned myact = (active >> kq) & 0x1111u;         // bit 4v: my v-th trial (kq + 4v) is running
    // persistent grid, contiguous balanced spans of 64-row rounds (as onepass_kernel)
    const int64_t q = rounds / gridDim.x, rem = rounds % gridDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * q + ((int64_t)blockIdx.x < rem ? (int64_t)blockIdx.x : rem);
    const int cnt = (int)(q + ((int64_t)blockIdx.x < rem ? 1 : 0));
    const int64_t row0 = t0 * kStep;
    const int lrow = wave * 16 + r16;
    const uint32_t vo = (uint32_t)(lrow * 8) + (uint32_t)((int64_t)kq * ld * 8);
    const int64_t cs = (int64_t)4 * ld * 8;
    const char* Mb = reinterpret_cast<const char*>(M + row0);
    const int64_t last_off = (int64_t)lastc0 * ld * 8;
    double a[CPL];
    auto load_cols = [&](int k, int c0, int c1) {
        const char* tb = Mb + (int64_t)k * (kStep * 8);
        const char* lastb = tb + last_off;
#pragma unroll
        for (int c = c0; c < c1; ++c) a[c] = buf_load_f64<true>((c < glast) ? tb + (int64_t)c * cs : lastb, vo);
    };
    const NRStepE e = ep.e;
    struct RowIn { NRStepE::Row sh; double xn[4], yn[4]; };
    auto fetch = [&](int64_t row) {
        RowIn w;
        const bool ok = row < n;                              // (vectors are padded to whole tiles, but a span may end beyond them)
        const int64_t rr = ok ? row : 0;
        w.sh.xn = w.sh.yn = w.sh.kk = 0.0;
        if (ST) {
            w.sh.xo = e.xold[rr]; w.sh.yo = e.xold[e.hs + rr]; w.sh.ax = e.sx[rr]; w.sh.ay = e.sy[rr];
            w.sh.q = e.q[rr]; w.sh.r = e.r[rr]; w.sh.s = e.s[rr]; w.sh.t = e.t[rr];
        } else {
            w.sh.xo = w.sh.yo = w.sh.ax = w.sh.ay = w.sh.q = w.sh.r = w.sh.s = w.sh.t = 0.0;
        }
        // (unconditional: a finished or missing trial's pointer is a valid one, its values are not used -- branches around loads would make
        // the compiler lose count of the outstanding ones and drain them all before the next first product)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            w.xn[v] = xpv[v][rr];
            w.yn[v] = ST ? xpv[v][e.hs + rr] : 0.0;
        }
        return w;
    };
    nrb_f64x4 C[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) C[b] = nrb_f64x4{0.0, 0.0, 0.0, 0.0};
    double ball[4] = {0.0, 0.0, 0.0, 0.0};
    double* tjw = tj[wave];
    double* vsw = vs[wave];
    double* const tjwr = tjw + r16 * kNRBLdr + kq;            // my write position (column-wise: + 4 per register)
    const double* const tjrd = tjw + kq * kNRBLdr + r16;      // my read position (row-wise: + 4 ks rows, + 16 per block)
    // Copy of the last group's register, taken after the first product: every chunk's staging write needs that group, but its reload for the
    // next tile must not wait for the last chunk -- without a separate store counter on this architecture the compiler drains ALL outstanding
    // memory operations (s_waitcnt vmcnt(0)) before the next first product, so a load issued right before it costs a full memory latency.
    double alast = 0.0;
    load_cols(0, 0, CPL);
    RowIn in = fetch(row0 + lrow);
    // transposed-tile chunk h: registers [16h, 16h + 16) hold its columns (true column 4c + kq), the shifted last group goes where its true columns are
    auto tj_write = [&](int h) {
#pragma unroll
        for (int c = 16 * h; c < 16 * h + 16 && c < CPL; ++c)
            if (c < glast) tjwr[4 * (c - 16 * h)] = a[c];
        const int lc = lastc0 + kq - kNRBChunk * h;          // the last group (register CPL - 1 always holds it; `alast` = its copy, below)
        if (lc >= 0 && lc < kNRBChunk) tjw[r16 * kNRBLdr + lc] = alast;
    };
    // first product: Y'[trial, row] over all columns -- two accumulator chains, the coefficient operands read from LDS two MFMAs ahead
    // of their use (left to itself the compiler emits read -> wait -> MFMA, the LDS latency exposed 33 times per tile)
    nrb_f64x4 y;
    auto first_product = [&]() {
        nrb_f64x4 y0 = nrb_f64x4{0.0, 0.0, 0.0, 0.0}, y1 = nrb_f64x4{0.0, 0.0, 0.0, 0.0};
        double t0 = ts[kq][r16], t1 = ts[(CPL > 1 ? 4 : 0) + kq][r16];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const double tn = ts[(c + 2 < CPL ? 4 * (c + 2) : 0) + kq][r16];
            if (c & 1) y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(t0, a[c], y1, 0, 0, 0);
            else y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(t0, a[c], y0, 0, 0, 0);
            t0 = t1; t1 = tn;
        }
        y = y0 + y1;
    };
    // The loop is ROTATED: an iteration = row update and second product of tile k, then the first product of tile k + 1 -- so that the
    // loads of tile k + 1 (issued inside the second product) and their first use stand in the same iteration and the compiler counts
    // the outstanding loads exactly (s_waitcnt vmcnt(N): the first product starts on the oldest columns while the youngest are still
    // on their way) instead of draining everything at the loop head.
    auto tile_step = [&](int k, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        compiler_fence();
        // ---- row update of my row for my four trials ----
        const int64_t row = row0 + lrow + (int64_t)k * kStep;
        const bool valid = row < n;
        double vv[4];
        // the row's part of y_retract! (a square root and two divisions, or a square root) once for this lane's four trial points
        const YRowPre ypre = ST ? y_retract_pre(in.sh.yo, in.sh.q, in.sh.r, in.sh.s, in.sh.t) : YRowPre{0.0, 0.0, 0.0};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            vv[v] = 0.0;
            if ((myact >> (4 * v)) & 1u) {
                NRStepE::Row w = in.sh;
                w.xn = in.xn[v]; w.yn = in.yn[v];
                double so[2] = {0.0, 0.0};                  // the new iterate (x and y halves), stored below through the GLOBAL pointer
                vv[v] = e.apply1<ST>(row, 0u, y[v], valid, true, w, ball[v], so, 1, 0.0, ST ? &ypre : nullptr);
                if (valid && !e.eval_only) {                 // (eval_only: the launch that evaluates c! at the trial points: nothing is updated)
                    xpv[v][row] = so[0];
                    if (ST) xpv[v][e.hs + row] = so[1];
                }
            }
            if (ST) sched_fence();
        }
        // (the registers of chunk 0 have been dead since tj_write(0); their reload is issued only now: the row update's square roots and
        // divisions needed the registers meanwhile, and the loads still have the whole second product as head start)
        compiler_fence();
        if (MORE) {
            load_cols(k + 1, 0, NCH == 1 ? CPL : (16 < CPL - 1 ? 16 : CPL - 1));     // every column staged (one chunk): the whole tile
            if (NCH > 1) load_cols(k + 1, CPL - 1, CPL);                             // the last group (its copy `alast` serves the later chunks)
            if (NCH > 1 && 16 * NCH < CPL - 1) load_cols(k + 1, 16 * NCH, CPL - 1);  // column groups beyond the staged chunks
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) vsw[r16 * 17 + kq + 4 * v] = vv[v];
        wave_lds_fence();
        double bop[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bop[ks] = vsw[(4 * ks + kq) * 17 + r16];     // B[k = row 4ks + kq][j = trial r16]
        if (MORE) in = fetch(row + kStep);
        // ---- second product, chunk by chunk; A[i = col 16cb + r16][k = row 4ks + kq] from the staging area, read two MFMAs ahead ----
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            if (h > 0) {
                wave_lds_fence();                            // (the previous chunk's reads precede these writes in program order)
                tj_write(h);
                compiler_fence();
                if (MORE) load_cols(k + 1, 16 * h, (16 * h + 16 < CPL - 1) ? 16 * h + 16 : CPL - 1);
            }
            wave_lds_fence();
            constexpr int kBlkMax = 4;
            const int nsteps = 4 * ((NBLK - 4 * h) < kBlkMax ? (NBLK - 4 * h) : kBlkMax);
            auto rd = [&](int i) { return tjrd[4 * (i & 3) * kNRBLdr + 16 * (i >> 2)]; };     // row 4ks + kq, column 16cb + r16
            double p0 = rd(0), p1 = rd(1);
#pragma unroll
            for (int i = 0; i < 4 * kBlkMax; ++i) {
                if (i < nsteps) {
                    const double pn = rd(i + 2 < nsteps ? i + 2 : 0);
                    C[4 * h + (i >> 2) < NBLK ? 4 * h + (i >> 2) : 0] =
                        __builtin_amdgcn_mfma_f64_16x16x4f64(p0, bop[i & 3], C[4 * h + (i >> 2) < NBLK ? 4 * h + (i >> 2) : 0], 0, 0, 0);
                    p0 = p1; p1 = pn;
                }
            }
        }
        wave_lds_fence();                                    // the next tile's writes to vs / tj come after this tile's reads
        if (MORE) {
            first_product();
            alast = a[CPL - 1];
            tj_write(0);
        }
    };
    first_product();
    alast = a[CPL - 1];
    tj_write(0);
#pragma unroll 1
    for (int k = 0; k < cnt - 1; ++k) tile_step(k, std::true_type());
    tile_step(cnt - 1, std::false_type());
    // ---- this wave's partial row ----
    double* prow = part + ((int64_t)blockIdx.x * kWaves + wave) * part_ld;
#pragma unroll
    for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int col = 16 * cb + kq + 4 * v;            // D[i = kq + 4v][j = trial r16]
            if (col < ncT) prow[r16 * ncT + col] = C[cb][v];
        }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        double s = ball[v];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        if (r16 == 0) prow[kNRBW * ncT + kq + 4 * v] = s;
    }
}


// ---- WIDE form: 132 < columns <= 528 --------------------------------------------------------------------------------------------------
// A 16-row tile of 512 columns is 128 doubles per lane: it does not fit one wave.  As in the wide form of the one-pass kernel the FOUR waves of a
// workgroup share one 16-row tile and split its columns -- wave w holds the column groups [w CPL, (w + 1) CPL) -- and a pass takes up to
// EIGHT trials (the coefficient table of 528 columns x 16 trials would be 66 KB of LDS: one workgroup per CU; with eight it is 33 KB and two fit):
//   first product   per wave over its columns, Y'[trial, row] partial sums; the four partials meet in LDS (barrier 1);
//   row update      16 rows x 8 trials = 128 updates: waves 0 and 1, one (row, trial) per lane (trial kq + 4 wave); the new values go to the
//                   workgroup's shared B-operand area (barrier 2);
//   second product  per wave over its own columns (transposed through its own LDS staging area, exactly as the narrow form), C[column, trial].
// The MFMA's N dimension is 16: trials 8 .. 15 do not exist -- their coefficient lanes read trial r16 & 7 (a duplicate: it only reaches output
// rows nobody reads), their B-operand entries are zero.  Arithmetic per tile and wave at 512 columns: 32 + 32 MFMAs of 64 cycles against ~12000
// cycles of HBM time per tile pair and CU: still HBM-bound.  One partial row per WORKGROUP: [trial * ncT + col] (8 x ncT), then the 8 ball partials.
constexpr int kNRBWideTrials = 8;

template <bool ST, int CPL>
__global__ __launch_bounds__(kThreads, 2) void nrb_mfma_wide_kernel(const double* __restrict__ M, int64_t ld, int ncN, int ncT, int64_t n, int64_t rounds,
                                                                    const double* __restrict__ t, int t_stride, NRBatchArgs ep, double* __restrict__ part,
                                                                    int part_ld) {
    if (ld_stat(ep.all) != 0) return;
    constexpr int NT = kNRBWideTrials;
    constexpr int NBLK = (CPL + 3) / 4;                       // 16-column blocks of this wave's second product
    constexpr int NCH = (CPL + 15) / 16;                      // chunks of 16 column registers (64 columns) staged at a time
    constexpr int kStep = 16;
    __shared__ double ts[kWaves * CPL * 4][NT];
    __shared__ double tj[kWaves][16 * kNRBLdr];
    __shared__ double vs[16 * 17];
    __shared__ double ysh[kWaves][2][64];
    __shared__ double* xp[NT];
    // (the wave index as a SCALAR: the column bases below then live in scalar registers -- left as threadIdx.x >> 6 the compiler treats every
    // buffer base as lane-dependent, wraps each of the tile's loads in a readfirstlane loop and keeps 64-bit addresses in vector registers)
    const int lane = threadIdx.x & 63, wave = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int g0 = wave * CPL;                                // first column group of this wave
    const int glast = (ncN - 1) / 4, lastc0 = ncN - 4;
    for (int j = threadIdx.x; j < kWaves * CPL * 4 * NT; j += kThreads) {
        const int slot = j / NT, b = j - slot * NT, g = slot >> 2, hh = slot & 3;
        double v = 0.0;
        if (b < ep.nb) {
            const double* tb = t + (int64_t)b * t_stride;
            if (g < glast) v = ld_scal(tb + slot);
            else if (g == glast && lastc0 + hh >= glast * 4) v = ld_scal(tb + lastc0 + hh);
        }
        ts[slot][b] = v;
    }
    for (int j = threadIdx.x; j < 16 * 17; j += kThreads) vs[j] = 0.0;       // (entries of the trials 8 .. 15 stay zero)
    unsigned active = 0u;
    for (int b = 0; b < ep.nb && b < NT; ++b)
        if (ld_stat(ep.ist + 4 * b) == 0) active |= 1u << b;
    if (threadIdx.x < NT) {
        const int b = (int)threadIdx.x;
        const int first = active ? __builtin_ctz(active) : 0;
        xp[b] = ep.xnew[(b < ep.nb && ((active >> b) & 1u)) ? b : first];
    }
    __syncthreads();
    const bool updater = wave < 2;                            // waves 0, 1: the row update of trial kq + 4 wave for row r16
    const int mytrial = kq + 4 * (wave & 1);
    const nrb_gptr xpv = (nrb_gptr)xp[mytrial];
    const bool myact = updater && ((active >> mytrial) & 1u);
    const int64_t q = rounds / gridDim.x, rem = rounds % gridDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * q + ((int64_t)blockIdx.x < rem ? (int64_t)blockIdx.x : rem);
    const int cnt = (int)(q + ((int64_t)blockIdx.x < rem ? 1 : 0));
    const int64_t row0 = t0 * kStep;
    const uint32_t vo = (uint32_t)(r16 * 8) + (uint32_t)((int64_t)kq * ld * 8);
    const int64_t cs = (int64_t)4 * ld * 8;
    const char* Mb = reinterpret_cast<const char*>(M + row0);
    const int64_t first_off = (int64_t)g0 * cs, last_off = (int64_t)lastc0 * ld * 8;
    double a[CPL];
    auto load_cols = [&](int k, int c0, int c1) {
        const char* tb = Mb + (int64_t)k * (kStep * 8);
        const char* lastb = tb + last_off;
#pragma unroll
        for (int c = c0; c < c1; ++c) a[c] = buf_load_f64<true>(uniform_ptr((g0 + c < glast) ? tb + first_off + (int64_t)c * cs : lastb), vo);
    };
    const NRStepE e = ep.e;
    struct RowIn { NRStepE::Row sh; double xn, yn; };
    auto fetch = [&](int64_t row) {
        RowIn w;
        const bool ok = row < n;
        const int64_t rr = ok ? row : 0;
        w.sh.xn = w.sh.yn = w.sh.kk = 0.0;
        if (ST) {
            w.sh.xo = e.xold[rr]; w.sh.yo = e.xold[e.hs + rr]; w.sh.ax = e.sx[rr]; w.sh.ay = e.sy[rr];
            w.sh.q = e.q[rr]; w.sh.r = e.r[rr]; w.sh.s = e.s[rr]; w.sh.t = e.t[rr];
        } else {
            w.sh.xo = w.sh.yo = w.sh.ax = w.sh.ay = w.sh.q = w.sh.r = w.sh.s = w.sh.t = 0.0;
        }
        w.xn = xpv[rr];
        w.yn = ST ? xpv[e.hs + rr] : 0.0;
        return w;
    };
    nrb_f64x4 C[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) C[b] = nrb_f64x4{0.0, 0.0, 0.0, 0.0};
    double ball = 0.0;
    double* tjw = tj[wave];
    double* const tjwr = tjw + r16 * kNRBLdr + kq;
    const double* const tjrd = tjw + kq * kNRBLdr + r16;
    // register CPL - 1 is staged from a COPY (alast) so that its reload for the next tile need not wait for the last chunk; where it goes: its
    // true columns when it is an ordinary group of this wave, the shifted last group's true columns (relative to this wave's first column) when
    // this wave holds the end of the matrix, nowhere when the wave lies beyond it
    const int lrel = ((g0 + CPL - 1 < glast) ? 4 * (g0 + CPL - 1) + kq : lastc0 + kq) - 4 * g0;
    double alast = 0.0;
    load_cols(0, 0, CPL);
    RowIn in = RowIn{};
    if (updater) in = fetch(row0 + r16);
    auto tj_write = [&](int h) {
#pragma unroll
        for (int c = 16 * h; c < 16 * h + 16 && c < CPL - 1; ++c)
            if (g0 + c < glast) tjwr[4 * (c - 16 * h)] = a[c];
        const int lc = lrel - kNRBChunk * h;
        if (lc >= 0 && lc < kNRBChunk) tjw[r16 * kNRBLdr + lc] = alast;
    };
    nrb_f64x4 y;
    auto first_product = [&]() {
        nrb_f64x4 y0 = nrb_f64x4{0.0, 0.0, 0.0, 0.0}, y1 = nrb_f64x4{0.0, 0.0, 0.0, 0.0};
        const int tr = r16 & (NT - 1);
        double t0 = ts[4 * g0 + kq][tr], t1 = ts[4 * (g0 + (CPL > 1 ? 1 : 0)) + kq][tr];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const double tn = ts[4 * (g0 + (c + 2 < CPL ? c + 2 : 0)) + kq][tr];
            if (c & 1) y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(t0, a[c], y1, 0, 0, 0);
            else y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(t0, a[c], y0, 0, 0, 0);
            t0 = t1; t1 = tn;
        }
        y = y0 + y1;
    };
    auto tile_step = [&](int k, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        compiler_fence();
        // ---- the four waves' partial first products meet: trials kq (v = 0) and kq + 4 (v = 1) of row r16 ----
        ysh[wave][0][lane] = y[0];
        ysh[wave][1][lane] = y[1];
        __syncthreads();
        const int64_t row = row0 + r16 + (int64_t)k * kStep;
        const bool valid = row < n;
        if (updater) {
            const int v = wave;                              // (wave 0: trials 0 .. 3, wave 1: trials 4 .. 7)
            const double ysum = (ysh[0][v][lane] + ysh[1][v][lane]) + (ysh[2][v][lane] + ysh[3][v][lane]);
            double vv = 0.0;
            if (myact) {
                const YRowPre ypre = ST ? y_retract_pre(in.sh.yo, in.sh.q, in.sh.r, in.sh.s, in.sh.t) : YRowPre{0.0, 0.0, 0.0};
                NRStepE::Row w = in.sh;
                w.xn = in.xn; w.yn = in.yn;
                double so[2] = {0.0, 0.0};
                vv = e.apply1<ST>(row, 0u, ysum, valid, true, w, ball, so, 1, 0.0, ST ? &ypre : nullptr);
                if (valid && !e.eval_only) {
                    xpv[row] = so[0];
                    if (ST) xpv[e.hs + row] = so[1];
                }
            }
            vs[r16 * 17 + mytrial] = vv;
        }
        compiler_fence();
        if (MORE) {
            load_cols(k + 1, 0, NCH == 1 ? CPL : (16 < CPL - 1 ? 16 : CPL - 1));
            if (NCH > 1) load_cols(k + 1, CPL - 1, CPL);
        }
        __syncthreads();
        double bop[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bop[ks] = vs[(4 * ks + kq) * 17 + r16];      // B[k = row 4ks + kq][j = trial r16] (zero for r16 >= 8)
        if (MORE && updater) in = fetch(row + kStep);
        // ---- second product over this wave's columns, chunk by chunk ----
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            if (h > 0) {
                wave_lds_fence();
                tj_write(h);
                compiler_fence();
                if (MORE) load_cols(k + 1, 16 * h, (16 * h + 16 < CPL - 1) ? 16 * h + 16 : CPL - 1);
            }
            wave_lds_fence();
            constexpr int kBlkMax = 4;
            const int nsteps = 4 * ((NBLK - 4 * h) < kBlkMax ? (NBLK - 4 * h) : kBlkMax);
            auto rd = [&](int i) { return tjrd[4 * (i & 3) * kNRBLdr + 16 * (i >> 2)]; };
            double p0 = rd(0), p1 = rd(1);
#pragma unroll
            for (int i = 0; i < 4 * kBlkMax; ++i) {
                if (i < nsteps) {
                    const double pn = rd(i + 2 < nsteps ? i + 2 : 0);
                    C[4 * h + (i >> 2) < NBLK ? 4 * h + (i >> 2) : 0] =
                        __builtin_amdgcn_mfma_f64_16x16x4f64(p0, bop[i & 3], C[4 * h + (i >> 2) < NBLK ? 4 * h + (i >> 2) : 0], 0, 0, 0);
                    p0 = p1; p1 = pn;
                }
            }
        }
        wave_lds_fence();
        if (MORE) {
            first_product();
            alast = a[CPL - 1];
            tj_write(0);
        }
    };
    first_product();
    alast = a[CPL - 1];
    tj_write(0);
#pragma unroll 1
    for (int k = 0; k < cnt - 1; ++k) tile_step(k, std::true_type());
    tile_step(cnt - 1, std::false_type());
    // ---- the workgroup's partial row: every wave its own columns ----
    double* prow = part + (int64_t)blockIdx.x * part_ld;
#pragma unroll
    for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int rel = 16 * cb + kq + 4 * v;            // D[i = kq + 4v][j = trial r16]
            const int col = 4 * g0 + rel;
            if (rel < 4 * CPL && col < ncT && r16 < NT) prow[r16 * ncT + col] = C[cb][v];
        }
    if (updater) {
        double s = ball;
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        if (r16 == 0) prow[NT * ncT + mytrial] = s;
    }
}


}  // namespace lfpsqp
