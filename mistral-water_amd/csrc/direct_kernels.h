// direct_kernels.h -- separable direct-sum evaluation of FFTMesh.EvaluateWaves for grids the FFT cannot express:
// non-power-of-two N, N < 64, or unit_width != length/N -- the SHIPPED scene (N = 12, unitWidth 1, length 12.39,
// D/FFT Mesh.unity:147-150) and the Inspector defaults (resolution 50, length 1, unitWidth 1, S/FFTMesh.cs:13-19).
//
// O(N^3) instead of the reference's O(N^4) by
//   sum_ij F(i,j) e^{i(kx_i x_a + kz_j z_b)} = sum_i e^{i kx_i x_a} [ sum_j F(i,j) e^{i kz_j z_b} ]        (S/FFTMesh.cs:199-217)
// and the two sums are dense matrix products -- the one place on this path where the matrix cores apply (80 N^3 flop against
// O(N^2) bytes).  With E[j][b] = e^{i k_j pos_b} (phase formed in f64, fixed per handle) and the five multiplier spectra F_f:
//   step 1 (z sum)   T_f = F_f E           complex (N x N)(N x N), as two REAL products with K = 2N:
//                      Tr_f = [Fr_f | Fi_f] [Er ; -Ei]      Ti_f = [Fr_f | Fi_f] [Ei ; Er]
//   step 2 (x sum)   only ONE real component of each output is used (:211-218: H = Re, Dx Dz Sx Sz = Im), so it is a real
//                    product with K = 2N:   Re: [Er^T | -Ei^T] [Tr_f ; Ti_f]      Im: [Ei^T | Er^T] [Tr_f ; Ti_f]
// = 60 N^3 flop per step on v_mfma_f32_32x32x2_f32 (exact f32: the result is an fmaf chain in k order), instead of five
// complex accumulators per thread walking global memory.  All operands are zero-padded to Np = a multiple of 64, so the
// GEMM has no bounds checks and only aligned 16-byte loads.  E and the step-2 left factors are built ONCE per handle
// (they do not depend on t); per step: the spectrum kernel, 4 GEMM launches, one assembly kernel, the whitecap kernel.
#pragma once
#include "mw_switches.h"
#include "fftmesh_kernels.h"
#include "czt_kernels.h"

namespace mw {

// chirp-z form (czt_kernels.h; the default for N <= 2048): tables + the two complex work arrays
struct CztState {
    int M = 0;
    cf *w1 = nullptr, *w2 = nullptr, *Hh = nullptr, *TWf = nullptr, *TWi = nullptr;
    cf *TT = nullptr, *O = nullptr;  // packed planes (czt_packed_value): [3][N][N + 1] after the z sum (transposed), [3][N][N] after the x sum
    float *Om = nullptr, *K = nullptr;  // CztArgs::Om / K (k_czt_tables)
    float table_gravity = -1.f;
    float table_length = -1.f, table_unit_width = -1.f;
};

struct DirectState {
    CztState czt;
    bool use_czt = false;
    int N = 0, Np = 0;
    float* A1 = nullptr;     // [5][Np][2Np]   (Fr_f | Fi_f), rebuilt every step
    float* T = nullptr;      // [5][2Np][Np]   (Tr_f ; Ti_f)
    float* B1re = nullptr;   // [2Np][Np]      [Er ; -Ei]
    float* B1im = nullptr;   // [2Np][Np]      [Ei ;  Er]
    float* A2re = nullptr;   // [Np][2Np]      [Er^T | -Ei^T]
    float* A2im = nullptr;   // [Np][2Np]      [Ei^T |  Er^T]
    float* out = nullptr;    // [5][Np][Np]    H, Dx, Dz, Sx, Sz
    cf* hds = nullptr;       // [N*N]
    float table_length = -1.f, table_unit_width = -1.f;  // what the E tables were built for
};

#if defined(__HIPCC__)
// e^{i k_j pos_b}, S/FFTMesh.cs:201-208 (phase in f64), scattered into the four fixed GEMM operands
__global__ void k_direct_tables(int N, int Np, float length, float unit_width, float* B1re, float* B1im, float* A2re, float* A2im) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    const int j = idx / N, b = idx % N;
    const double k = (double)wave_k(N, length, j), pos = (double)rest_coord(N, unit_width, b);
    double s, c;
    sincos(k * pos, &s, &c);
    const float er = (float)c, ei = (float)s;
    B1re[(size_t)j * Np + b] = er;  B1re[(size_t)(Np + j) * Np + b] = -ei;
    B1im[(size_t)j * Np + b] = ei;  B1im[(size_t)(Np + j) * Np + b] = er;
    // step 2 contracts over i with E[i][a]: here (j, b) plays (i, a)
    A2re[(size_t)b * 2 * Np + j] = er;  A2re[(size_t)b * 2 * Np + Np + j] = -ei;
    A2im[(size_t)b * 2 * Np + j] = ei;  A2im[(size_t)b * 2 * Np + Np + j] = er;
}

// S/FFTMesh.cs:178-190 htilde + the five multiplier spectra of :211-215, written as the left factors of step 1
__global__ void k_direct_spec(OceanConsts C, int Np, const cf* h0, const cf* h0c, float t, float* A1) {
    const int N = C.N;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    int i = idx / N, j = idx % N;
    float s, c;
    mw_sincos(omega_t_f32(N, C.length, C.gravity, i, j, t), &s, &c);
    cf a = h0[idx], b = h0c[idx];
    cf h = mk(a.x * c - a.y * s + b.x * c + b.y * s, a.x * s + a.y * c - b.x * s + b.y * c);  // :188
    float kx = wave_k(N, C.length, i), kz = wave_k(N, C.length, j);
    float kl = sqrtf(kx * kx + kz * kz);
    float ux = 0.f, uzn = 0.f;
    if (!(kl < MW_EPS_F)) { ux = kx / kl; uzn = -kz / kl; }  // :213-215
    const float mul[5] = {1.f, ux, uzn, kx, kz};
    const size_t plane = (size_t)Np * 2 * Np, row = (size_t)i * 2 * Np;
#pragma unroll
    for (int f = 0; f < 5; f++) {
        A1[f * plane + row + j] = h.x * mul[f];
        A1[f * plane + row + Np + j] = h.y * mul[f];
    }
}

// C[z] = A[z] B[z]: real f32, row-major, every dimension a multiple of the tile (operands are zero-padded), on
// v_mfma_f32_32x32x2_f32.  64 x 64 tile per 256-thread workgroup, one 32 x 32 accumulator per wave, K in steps of 16 through
// LDS with the next step's global loads in flight behind the current step's 8 MFMAs per wave.
//   A operand of the MFMA: lane l holds A[i = l & 31][k = l >> 5]; B: B[k = l >> 5][j = l & 31];
//   D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)          (cdna_hip_programming.md section 3)
// LDS tiles are k-major (As[k][m], Bs[k][n]) so that both operand reads are 32 consecutive dwords per half-wave: conflict-free
// ds_read_b32.  Row strides: As 66 (the transposing ds_write_b32 of a staged A fragment hits rows 4 q + c: 4 * 66 = 8 mod 32
// spreads the four q of a 32-lane group over distinct banks), Bs 68 (rows stay 16-byte aligned for the ds_write_b128).
typedef float f16v __attribute__((ext_vector_type(16)));
#define MW_GEMM_BM 64
#define MW_GEMM_BN 64
#define MW_GEMM_BK 16
__global__ __launch_bounds__(256) void k_gemm_f32_mfma(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                       int K, int lda, int ldb, int ldc, long long sA, long long sB, long long sC) {
    constexpr int BM = MW_GEMM_BM, BN = MW_GEMM_BN, BK = MW_GEMM_BK;
    __shared__ float As[2][BK][BM + 2];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + 4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    A += (size_t)blockIdx.z * sA + (size_t)m0 * lda;
    B += (size_t)blockIdx.z * sB + n0;
    C += (size_t)blockIdx.z * sC + (size_t)m0 * ldc + n0;
    // global -> register staging: A tile 64 x 16 (thread: row t / 4, four k), B tile 16 x 64 (thread: k t / 16, four n)
    const int ar = t >> 2, ak = (t & 3) * 4, bk = t >> 4, bn = (t & 15) * 4;
    const f4* Ag = reinterpret_cast<const f4*>(A + (size_t)ar * lda + ak);
    const f4* Bg = reinterpret_cast<const f4*>(B + (size_t)bk * ldb + bn);
    f4 ra = Ag[0], rb = Bg[0];
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32, li = lane & 31, lk = lane >> 5;
    const int nk = K / BK;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        As[buf][ak + 0][ar] = ra.x; As[buf][ak + 1][ar] = ra.y; As[buf][ak + 2][ar] = ra.z; As[buf][ak + 3][ar] = ra.w;
        *reinterpret_cast<f4*>(&Bs[buf][bk][bn]) = rb;
        __syncthreads();  // one barrier per step: the other buffer is only rewritten after the NEXT barrier
        if (kt + 1 < nk) {
            ra = *reinterpret_cast<const f4*>(reinterpret_cast<const float*>(Ag) + (size_t)(kt + 1) * BK);
            rb = *reinterpret_cast<const f4*>(reinterpret_cast<const float*>(Bg) + (size_t)(kt + 1) * BK * ldb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][kk + lk][wm + li], Bs[buf][kk + lk][wn + li], acc, 0, 0, 0);
    }
    float* Cw = C + (size_t)wm * ldc + wn + li;
#pragma unroll
    for (int r = 0; r < 16; r++) Cw[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lk) * ldc] = acc[r];
}

// vertices / normals / hds from the five real output planes (S/FFTMesh.cs:218, 243-247)
__global__ void k_direct_assemble(OceanConsts C, int Np, const float* out, cf* hds, float* vertices, float* normals) {
    const int N = C.N;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    int a = idx / N, b = idx % N;
    const size_t plane = (size_t)Np * Np, o = (size_t)a * Np + b;
    const float h = out[o], dx = out[plane + o], dz = out[2 * plane + o], sx = out[3 * plane + o], sz = out[4 * plane + o];
    const float mag = sqrtf(sx * sx + 1.0f + sz * sz);  // up - n, S/FFTMesh.cs:218
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (mag > 1e-5f) { nx = sx / mag; ny = 1.0f / mag; nz = sz / mag; }
    normals[3 * idx] = nx; normals[3 * idx + 1] = ny; normals[3 * idx + 2] = nz;
    vertices[3 * idx + 0] = ssub(rest_coord(N, C.unit_width, a), smul(dx, C.choppiness));  // :245
    vertices[3 * idx + 1] = h;                                                             // :243
    vertices[3 * idx + 2] = ssub(rest_coord(N, C.unit_width, b), smul(dz, C.choppiness));  // :244
    hds[idx] = mk(dx, dz);                                                                 // :247
}

__global__ void k_direct_white(int N, const cf* hds, const float* normals, float* white, int white_stride) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    int i = idx / N, j = idx % N;
    const bool hi = (i != N - 1), hj = (j != N - 1);
    cf z = mk(0.f, 0.f);
    float xx = whitecap(hds[idx], hi ? hds[idx + N] : z, hj ? hds[idx + 1] : z, hi, hj, normals[3 * idx], normals[3 * idx + 2]);
    if (white_stride == 1) white[idx] = xx;
    else { white[4 * idx] = xx; white[4 * idx + 1] = xx; white[4 * idx + 2] = xx; white[4 * idx + 3] = xx; }
}

// ---- chirp-z launches ---------------------------------------------------------------------------------------------------
// one axis of the sum for RW rows per workgroup: pre-chirp + zero padding, forward transform, kernel product, inverse
// transform, post-chirp, transposed store (czt_kernels.h).  Twiddles beyond the first table come from global memory (L1 / L2 hits).
// Barriers are workgroup-uniform: rows past the end compute on zeros and store nothing.
#ifndef MW_CZT_XCD_GROUP
#define MW_CZT_XCD_GROUP 1
#endif
// Small transforms (round 5, second pass): a step on these grids is a chain of dependent latencies, not bandwidth -- and up to here every
// table value was asked for where it is used: the twiddle rows of all passes of both transforms from global memory (an L2 round trip in
// front of each pass), the kernel's transform after the forward transform's last barrier, the post-chirp after the inverse one.  Now the
// twiddle tables of both directions are staged in LDS at the top of the kernel (CztTw; published by the first barrier of czt_line) and the
// kernel's transform / the post-chirp are requested before the first input load, so that ONE memory latency covers them all (M <= 512 only: at
// M = 2048 the 64 extra registers of the early loads cost a wave per SIMD, N = 1000 98 -> 122 us).  Same table
// values, same expressions: the same bits (tests/test_zz_frame_plan.py, every direct-path parity test).
#ifndef MW_CZT_PRELOAD
#define MW_CZT_PRELOAD 1
#endif
#ifndef MW_CZT_STAGE_MAX_M
#define MW_CZT_STAGE_MAX_M 512  // larger transforms keep their tables in global memory (their LDS decides how many workgroups share a CU)
#endif
template <int M, int P>
struct CztTw {
    static constexpr bool STAGE = MW_CZT_PRELOAD && M <= MW_CZT_STAGE_MAX_M && TwGeom<M, P>::IN_LDS;
    static constexpr int HALF = STAGE ? ((TwGeom<M, P>::LDS_ALL + 1) & ~1) : 0;  // cf entries per direction, 16-B aligned
    static constexpr int CF = 2 * HALF;                                          // in front of the line buffers
};
// the workgroup's share of both tables: requested at the top of the kernel (load), written to LDS behind the requests of the first input
// row and in front of the first barrier of czt_line_core (store; TwStage's two halves, mw_math.h).  NT_MIN <= the workgroup's thread count.
template <int M, int P, int NT_MIN>
struct CztTwRegs {
    static constexpr int IT = CztTw<M, P>::STAGE ? (TwGeom<M, P>::LDS_CF + NT_MIN - 1) / NT_MIN : 0;
    cf f[IT > 0 ? IT : 1], b[IT > 0 ? IT : 1];
    int tid = 0, nthreads = NT_MIN;
    cf* twl = nullptr;
    __device__ __forceinline__ void load(const CztArgs& A, cf* twl_, int tid_, int nthreads_) {
        tid = tid_; nthreads = nthreads_; twl = twl_;
#pragma unroll
        for (int k = 0; k < IT; k++) {
            const int i = tid + k * nthreads;
            const bool in = i < TwGeom<M, P>::LDS_CF;
            f[k] = in ? A.TWf[i] : mk(0.f, 0.f);
            b[k] = in ? A.TWi[i] : mk(0.f, 0.f);
        }
    }
    __device__ __forceinline__ void operator()() const {
#pragma unroll
        for (int k = 0; k < IT; k++) {
            const int i = tid + k * nthreads;
            if (i < TwGeom<M, P>::LDS_CF) { twl[i] = f[k]; twl[CztTw<M, P>::HALF + i] = b[k]; }
        }
    }
};
struct CztNoHook { __device__ __forceinline__ void operator()() const {} };
// the transforms of a line whose pre-chirped, zero-padded input is in x (x = the inverse transform's output afterwards); hh = the kernel's
// transform at this thread's slots where STAGE.  Barriers inside: every thread of the workgroup must call it.
// before_first_barrier: called once in front of the first barrier (the LDS writes of the staged tables)
template <int M, int P, class Hook>
__device__ __forceinline__ void czt_line_core(const CztArgs& A, int u, cf* buf, cf (&x)[P], const cf* twl, const cf (&hh)[P], const Hook& before_first_barrier) {
    constexpr bool ST = CztTw<M, P>::STAGE;
    constexpr int T = M / P;
    const Twiddles twf = ST ? TwGeom<M, P>::view(A.TWf, twl) : TwGeom<M, P>::view(A.TWf);
    const Twiddles twi = ST ? TwGeom<M, P>::view(A.TWi, twl + CztTw<M, P>::HALF) : TwGeom<M, P>::view(A.TWi);
    // A line's buffer is written and read by the line's own T threads only: where those sit in ONE wave (T <= 64: M <= 512) its exchanges
    // need that wave's LDS operations in order and no workgroup barrier (the lines drift apart: a dozen barriers less on the dependent
    // chain of a small-grid step).  The first barrier stays a workgroup barrier: it publishes the staged tables.
#ifndef MW_CZT_WAVE_SYNC
#define MW_CZT_WAVE_SYNC 1
#endif
    constexpr bool WS = MW_CZT_WAVE_SYNC && T <= 64 && 64 % T == 0;
    auto line_sync = [&]() {
        if (WS) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    stage0_store<M, P, -1>(x, u, buf);
    before_first_barrier();
    __syncthreads();
#pragma unroll
    for (int s = 1; s < FftGeom<M, P>::S; s++) {
        const bool in_regs = LastInRegs<M, P>::value && s == FftGeom<M, P>::S - 1;  // M = P^S (4096): no identity round trip through LDS
        load_slots<M, P>(x, u, buf, s - 1);
        if (in_regs) { stage_regs<M, P, -1, false>(x, u, twf, s); continue; }
        line_sync();
        stage_store<M, P, -1, false>(x, u, buf, twf, s);
        line_sync();
    }
    if (!LastInRegs<M, P>::value) load_last<M, P>(x, u, buf);
    final_stage<M, P, -1>(x, u, twf.TF);
    if (ST) {
#pragma unroll
        for (int q = 0; q < P; q++) x[q] = cmul(x[q], hh[q]);  // czt_mul_kernel's expression
    } else {
        czt_mul_kernel<M, P>(A, u, x);
    }
    line_sync();  // every read of the forward transform's last exchange is done
    stage0_store<M, P, +1>(x, u, buf);
    line_sync();
#pragma unroll
    for (int s = 1; s < FftGeom<M, P>::S; s++) {
        const bool in_regs = LastInRegs<M, P>::value && s == FftGeom<M, P>::S - 1;
        load_slots<M, P>(x, u, buf, s - 1);
        if (in_regs) { stage_regs<M, P, +1, false>(x, u, twi, s); continue; }
        line_sync();
        stage_store<M, P, +1, false>(x, u, buf, twi, s);
        line_sync();
    }
    if (!LastInRegs<M, P>::value) load_last<M, P>(x, u, buf);
    final_stage<M, P, +1>(x, u, twi.TF);
}
// one line of one axis: pre-chirp + zero padding, forward transform, kernel product, inverse transform; x = the inverse transform's output
// (element n = u + T q in slot q), the post-chirp is the caller's.  Barriers inside: every thread of the workgroup must call it.
// twl: where the staged tables will be (CztTwRegs: written by before_first_barrier) where CztTw<M, P>::STAGE
template <int M, int P, class Hook>
__device__ __forceinline__ void czt_line(const CztArgs& A, int f, int row, bool live, int u, cf* buf, cf (&x)[P], const cf* twl, const Hook& before_first_barrier) {
    constexpr int T = M / P;
    cf hh[P];
    if (CztTw<M, P>::STAGE) {
#pragma unroll
        for (int q = 0; q < P; q++) hh[q] = A.Hh[u + T * q];
    }
    czt_load<M, P>(A, f, live ? row : 0, u, live, x);
    czt_line_core<M, P>(A, u, buf, x, twl, hh, before_first_barrier);
}
template <int M, int P, int RW>
__global__ __launch_bounds__((RW * M / P)) void k_czt(CztArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int T = M / P, BUF = FftGeom<M, P>::LBUF + 4;
    // Row blocks of one 128-B line of the TRANSPOSED output (16 consecutive rows = 16 / RW consecutive blocks, 8 B each per column) are
    // issued on ONE XCD (the dispatcher places block b on XCD b % 8), so that their 8 RW-byte pieces of a line meet in that XCD's L2
    // instead of leaving four L2s as partial lines.  Any bijection is correct; blocks past the last whole group of 8 x G keep their index.
    constexpr int G = 16 / RW > 0 ? 16 / RW : 1;
    int rb = (int)blockIdx.x;
    if (MW_CZT_XCD_GROUP && G > 1 && rb < (int)gridDim.x / (8 * G) * (8 * G)) {
        const int grp = rb / (8 * G), in = rb % (8 * G), xcd = in % 8, slot = in / 8;
        rb = grp * (8 * G) + xcd * G + slot;
    }
    const int tid = threadIdx.x, w = tid / T, u = tid % T, row = rb * RW + w, f = blockIdx.y;
    const bool live = row < A.rows;
    CztTwRegs<M, P, RW * T> twr;
    twr.load(A, lds, tid, RW * T);
    cf w2r[P];
    if (CztTw<M, P>::STAGE) {
#pragma unroll
        for (int q = 0; q < P; q++) w2r[q] = (u + T * q < A.nout) ? A.w2[u + T * q] : mk(0.f, 0.f);
    }
    cf x[P];
    czt_line<M, P>(A, f, row, live, u, lds + CztTw<M, P>::CF + (size_t)w * BUF, x, lds, twr);
    if (!live) return;
    if (CztTw<M, P>::STAGE) {  // czt_store with the post-chirp already here
        cf* __restrict__ o = A.out + (size_t)f * A.out_plane + row;
#pragma unroll
        for (int q = 0; q < P; q++) {
            const int n = u + T * q;
            if (n < A.nout) o[(size_t)n * A.out_ld] = cmul(x[q], w2r[q]);
        }
    } else {
        czt_store<M, P>(A, f, row, u, x);
    }
}
// Small grids (M <= 256, i.e. N <= 128: the reference's Inspector default N = 50 and its shipped scene N = 12): the SECOND axis and the
// assembly in ONE launch (round 5; a step is launch latency there -- three launches of a few workgroups, 15.5 us at N = 50).  A workgroup
// owns RW output rows b of ALL three packed planes plus the next row of the two planes that carry the displacement (the forward difference
// of the Jacobian reads (a, b + 1), S/FFTMesh.cs:264-267): 3 RW + 2 lines transformed side by side, post-chirped into their own LDS
// buffers, then vertices / normals / whitecap straight from there -- the plane O never exists in memory.  The arithmetic of a line and of
// a vertex is the three-launch plan's: the same bits (tests/test_zz_frame_plan.py::test_small_grid_fused_czt_equals_three_launches).
template <int M, int P, int RW>
__global__ __launch_bounds__(((3 * RW + 2) * M / P)) void k_czt_rows_assemble(CztArgs A, cf* hds, float* vertices, float* normals, float* white,
                                                                              int white_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int T = M / P, BUF = FftGeom<M, P>::LBUF + 4, NT = (3 * RW + 2) * T;
    static_assert(BUF >= M, "a line's exchange buffer holds its post-chirped outputs afterwards");
    const int tid = threadIdx.x, gi = tid / T, u = tid % T, b0 = (int)blockIdx.x * RW;
    const int f = gi < 3 * RW ? gi / RW : (gi == 3 * RW ? 0 : 2);         // halo lines: planes 0 (Dx in its imaginary part) and 2 (Dz)
    const int row = gi < 3 * RW ? b0 + gi % RW : b0 + RW;
    const bool live = row < A.rows;
    CztTwRegs<M, P, NT> twr;
    twr.load(A, lds, tid, NT);
    cf* const rows = lds + CztTw<M, P>::CF;
    cf* buf = rows + (size_t)gi * BUF;
    cf w2r[P];
    if (CztTw<M, P>::STAGE) {
#pragma unroll
        for (int q = 0; q < P; q++) w2r[q] = (u + T * q < A.nout) ? A.w2[u + T * q] : mk(0.f, 0.f);
    }
    cf x[P];
    czt_line<M, P>(A, f, row, live, u, buf, x, lds, twr);
    __syncthreads();  // the final pass has read its last exchange: the buffer now takes the post-chirped outputs, plain index a
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int n = u + T * q;
        if (n < A.nout) buf[n] = cmul(x[q], CztTw<M, P>::STAGE ? w2r[q] : A.w2[n]);  // czt_store's expression
    }
    __syncthreads();
    const OceanConsts& C = A.C;
    const int N = C.N;
    const cf* P0 = rows;                              // [w][a] = H + i Dx
    const cf* P1 = rows + (size_t)RW * BUF;           //          Sx + i Sz
    const cf* P2 = rows + (size_t)2 * RW * BUF;       //          Dz (real part)
    const cf* H0 = rows + (size_t)3 * RW * BUF;       // row b0 + RW of plane 0
    const cf* H2 = H0 + BUF;                         //                 plane 2
    for (int e = tid; e < RW * N; e += NT) {         // consecutive lanes -> consecutive b: RW x 12-byte runs
        const int a = e / RW, w = e % RW, b = b0 + w;
        if (b >= N) continue;
        const int idx = a * N + b;
        const cf p0 = P0[(size_t)w * BUF + a], p1 = P1[(size_t)w * BUF + a];
        const float h = p0.x, dx = p0.y, sx = p1.x, sz = p1.y, dz = P2[(size_t)w * BUF + a].x;
        const float mag = sqrtf(sx * sx + 1.0f + sz * sz);  // up - n, S/FFTMesh.cs:218 (k_czt_assemble_white's expressions from here on)
        float nx = 0.f, ny = 0.f, nz = 0.f;
        if (mag > 1e-5f) { nx = sx / mag; ny = 1.0f / mag; nz = sz / mag; }
        normals[3 * idx] = nx; normals[3 * idx + 1] = ny; normals[3 * idx + 2] = nz;
        vertices[3 * idx + 0] = ssub(rest_coord(N, C.unit_width, a), smul(dx, C.choppiness));  // :245
        vertices[3 * idx + 1] = h;                                                             // :243
        vertices[3 * idx + 2] = ssub(rest_coord(N, C.unit_width, b), smul(dz, C.choppiness));  // :244
        if (hds) hds[idx] = mk(dx, dz);                                                        // :247
        const bool hi = a != N - 1, hj = b != N - 1;
        const cf z = mk(0.f, 0.f);
        const cf di = hi ? mk(P0[(size_t)w * BUF + a + 1].y, P2[(size_t)w * BUF + a + 1].x) : z;
        const cf dj = hj ? (w + 1 < RW ? mk(P0[(size_t)(w + 1) * BUF + a].y, P2[(size_t)(w + 1) * BUF + a].x) : mk(H0[a].y, H2[a].x)) : z;
        const float xx = whitecap(mk(dx, dz), di, dj, hi, hj, nx, nz);
        if (white_stride == 1) white[idx] = xx;
        else { white[4 * idx] = xx; white[4 * idx + 1] = xx; white[4 * idx + 2] = xx; white[4 * idx + 3] = xx; }
    }
}
// Tiny grids (N <= 20, transform size 64; the reference's shipped scene is N = 12, D/FFT Mesh.unity:147-150): BOTH axes and the assembly in
// ONE workgroup and ONE launch (round 5) -- the whole step of such a grid is 3 (N + 1) + 3 N lines of 8 threads, and a second dependent launch
// costs more than one CU loses by doing all of them.  The plane between the axes (TT[p][b][i]) and the output planes live in LDS; every line
// and every vertex is formed by the expressions of the two-launch plan: the same bits (test_small_grid_fused_czt_equals_three_launches).
// Measured (us per step back to back, one launch / two): N = 4 6.5 / 8.8, 12 8.2 / 9.4, 16 9.5 / 10.1, 20 10.1 / 10.7, 24 13.5 / 11.8, 32 17.8 / 13.4
// (the geometry allows any N <= 32).
#ifndef MW_CZT_ONE_MAX_N
#define MW_CZT_ONE_MAX_N 20
#endif
template <int M, int P>
__global__ __launch_bounds__(1024) void k_czt_one(CztArgs A, cf* hds, float* vertices, float* normals, float* white, int white_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int T = M / P, BUF = FftGeom<M, P>::LBUF + 4;
    static_assert(BUF >= M, "a line's exchange buffer holds its post-chirped outputs afterwards");
    const OceanConsts& C = A.C;
    const int N = C.N, N1 = N + 1;
    const int tid = threadIdx.x, nthreads = blockDim.x, gi = tid / T, u = tid % T;
    cf* const TT = lds + CztTw<M, P>::CF;                       // [p][b][i], N + 1 entries per row
    cf* const rows = TT + (((size_t)MW_CZT_PLANES * N * N1 + 1) & ~(size_t)1);
    cf* const buf = rows + (size_t)gi * BUF;                    // blockDim.x / T line buffers
    CztTwRegs<M, P, 128> twr;  // (at least 128 threads: czt_launch_one rounds 3 (N + 1) lines of 8 up to whole waves, N >= 2)
    twr.load(A, lds, tid, nthreads);
    cf hh[P], w2r[P], w1r[P];
#pragma unroll
    for (int q = 0; q < P; q++) {
        hh[q] = A.Hh[u + T * q];
        w2r[q] = (u + T * q < N) ? A.w2[u + T * q] : mk(0.f, 0.f);
        w1r[q] = (u + T * q < N1) ? A.w1[u + T * q] : mk(0.f, 0.f);
    }
    cf x[P];
    {   // along j: line (p, i) formed from the spectrum, i = 0 .. N  ->  TT[p][b][i]
        const int f = gi / N1, row = gi % N1;
        const bool live = gi < MW_CZT_PLANES * N1;
        czt_load<M, P, 1>(A, f, live ? row : 0, u, live, x);  // element by element here (1024 threads: 128 registers, 48 of them early table values; measured 8.0 against 8.4 us at N = 12)
        czt_line_core<M, P>(A, u, buf, x, lds, hh, twr);
        if (live) {
#pragma unroll
            for (int q = 0; q < P; q++) {
                const int n = u + T * q;
                if (n < N) TT[((size_t)f * N + n) * N1 + row] = cmul(x[q], w2r[q]);  // czt_store's expression
            }
        }
    }
    __syncthreads();
    const int f = gi / N, b = gi % N;
    const bool live = gi < MW_CZT_PLANES * N;
    {   // along i: line (p, b) read from TT  ->  post-chirped outputs O[p][a][b] in the line's own buffer, index a
        const cf* r = TT + ((size_t)(live ? f : 0) * N + (live ? b : 0)) * N1;
#pragma unroll
        for (int q = 0; q < P; q++) {
            const int n = u + T * q;
            x[q] = (live && n < N1) ? cmul(r[n], w1r[q]) : mk(0.f, 0.f);  // czt_load's expression
        }
        czt_line_core<M, P>(A, u, buf, x, lds, hh, CztNoHook());
        __syncthreads();  // the final pass has read its last exchange
#pragma unroll
        for (int q = 0; q < P; q++) {
            const int n = u + T * q;
            if (n < N) buf[n] = cmul(x[q], w2r[q]);
        }
    }
    __syncthreads();
    for (int e = tid; e < N * N; e += nthreads) {  // k_czt_rows_assemble's expressions on the planes in LDS
        const int a = e / N, bb = e % N;
        const int idx = a * N + bb;
        const cf* P0 = rows + (size_t)bb * BUF;             // plane 0, row b: H + i Dx
        const cf* P1 = rows + (size_t)(N + bb) * BUF;       //          Sx + i Sz
        const cf* P2 = rows + (size_t)(2 * N + bb) * BUF;   //          Dz (real part)
        const cf p0 = P0[a], p1 = P1[a];
        const float h = p0.x, dx = p0.y, sx = p1.x, sz = p1.y, dz = P2[a].x;
        const float mag = sqrtf(sx * sx + 1.0f + sz * sz);  // up - n, S/FFTMesh.cs:218
        float nx = 0.f, ny = 0.f, nz = 0.f;
        if (mag > 1e-5f) { nx = sx / mag; ny = 1.0f / mag; nz = sz / mag; }
        normals[3 * idx] = nx; normals[3 * idx + 1] = ny; normals[3 * idx + 2] = nz;
        vertices[3 * idx + 0] = ssub(rest_coord(N, C.unit_width, a), smul(dx, C.choppiness));   // :245
        vertices[3 * idx + 1] = h;                                                              // :243
        vertices[3 * idx + 2] = ssub(rest_coord(N, C.unit_width, bb), smul(dz, C.choppiness));  // :244
        if (hds) hds[idx] = mk(dx, dz);                                                         // :247
        const bool hi = a != N - 1, hj = bb != N - 1;
        const cf z = mk(0.f, 0.f);
        const cf di = hi ? mk(P0[a + 1].y, P2[a + 1].x) : z;
        const cf dj = hj ? mk(P0[BUF + a].y, P2[BUF + a].x) : z;
        const float xx = whitecap(mk(dx, dz), di, dj, hi, hj, nx, nz);
        if (white_stride == 1) white[idx] = xx;
        else { white[4 * idx] = xx; white[4 * idx + 1] = xx; white[4 * idx + 2] = xx; white[4 * idx + 3] = xx; }
    }
}
// vertices / normals / whitecap from the three packed output planes O[p][a][b] = (H + i Dx, Sx + i Sz, Dz) (czt_packed_value;
// S/FFTMesh.cs:211-218), and the forward-difference Jacobian (:258-274) from the displacement of the two neighbours -- one launch;
// hds also serves the test hook (mw_debug_evaluate_hds)
__global__ void k_czt_assemble_white(OceanConsts C, const cf* O, cf* hds, float* vertices, float* normals, float* white, int white_stride) {
    const int N = C.N;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    int a = idx / N, b = idx % N;
    const size_t NN = (size_t)N * N;
    const float h = O[idx].x, dx = O[idx].y, sx = O[NN + idx].x, sz = O[NN + idx].y, dz = O[2 * NN + idx].x;
    const float mag = sqrtf(sx * sx + 1.0f + sz * sz);  // up - n, S/FFTMesh.cs:218
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (mag > 1e-5f) { nx = sx / mag; ny = 1.0f / mag; nz = sz / mag; }
    normals[3 * idx] = nx; normals[3 * idx + 1] = ny; normals[3 * idx + 2] = nz;
    vertices[3 * idx + 0] = ssub(rest_coord(N, C.unit_width, a), smul(dx, C.choppiness));  // :245
    vertices[3 * idx + 1] = h;                                                             // :243
    vertices[3 * idx + 2] = ssub(rest_coord(N, C.unit_width, b), smul(dz, C.choppiness));  // :244
    if (hds) hds[idx] = mk(dx, dz);                                                        // :247
    const bool hi = a != N - 1, hj = b != N - 1;
    const cf z = mk(0.f, 0.f);
    const cf di = hi ? mk(O[idx + N].y, O[2 * NN + idx + N].x) : z, dj = hj ? mk(O[idx + 1].y, O[2 * NN + idx + 1].x) : z;
    const float xx = whitecap(mk(dx, dz), di, dj, hi, hj, nx, nz);
    if (white_stride == 1) white[idx] = xx;
    else { white[4 * idx] = xx; white[4 * idx + 1] = xx; white[4 * idx + 2] = xx; white[4 * idx + 3] = xx; }
}

std::vector<cf> build_twiddle_table(int N, int P, int sgn);  // mistral_water.hip

static inline void czt_free(CztState& z) {
    hipFree(z.w1); hipFree(z.w2); hipFree(z.Hh); hipFree(z.TWf); hipFree(z.TWi); hipFree(z.TT); hipFree(z.O); hipFree(z.Om); hipFree(z.K);
    z = CztState();
}
static inline int czt_alloc(CztState& z, int N) {
    z.M = czt_size(N);
    if (!z.M) return 1;
    const int P = czt_points(z.M);
    const std::vector<cf> tf = build_twiddle_table(z.M, P, -1), ti = build_twiddle_table(z.M, P, +1);
    const size_t NN = (size_t)N * N;
    if (hipMalloc((void**)&z.w1, sizeof(cf) * (N + 1)) != hipSuccess || hipMalloc((void**)&z.w2, sizeof(cf) * N) != hipSuccess ||
        hipMalloc((void**)&z.Hh, sizeof(cf) * z.M) != hipSuccess || hipMalloc((void**)&z.TWf, sizeof(cf) * tf.size()) != hipSuccess ||
        hipMalloc((void**)&z.TWi, sizeof(cf) * ti.size()) != hipSuccess ||
        hipMalloc((void**)&z.TT, sizeof(cf) * MW_CZT_PLANES * (size_t)N * (N + 1)) != hipSuccess ||
        hipMalloc((void**)&z.O, sizeof(cf) * MW_CZT_PLANES * NN) != hipSuccess ||
        hipMalloc((void**)&z.Om, sizeof(float) * (size_t)(N + 1) * (N + 1)) != hipSuccess || hipMalloc((void**)&z.K, sizeof(float) * (N + 1)) != hipSuccess ||
        hipMemcpy(z.TWf, tf.data(), sizeof(cf) * tf.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(z.TWi, ti.data(), sizeof(cf) * ti.size(), hipMemcpyHostToDevice) != hipSuccess) {
        czt_free(z);
        return 4;
    }
    return 0;
}
// chirps and the transform of the wrapped kernel (f64 on the host): uploaded when the handle is created and when its length changes
// (mw_ocean_create / mw_ocean_reinit_spectrum) -- an enqueue never synchronises or copies
__global__ void k_czt_tables(int N, float length, float gravity, float* Om, float* K) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (N + 1) * (N + 1)) czt_table_element(N, length, gravity, e, Om, K);
}
static inline hipError_t czt_upload_tables(CztState& z, int N, float unit_width, float length, float gravity, hipStream_t st) {
    std::vector<cf> w1, w2, Hh;
    czt_build_tables(N, N + 1, z.M, unit_width, length, w1, w2, Hh);
    z.table_length = z.table_unit_width = z.table_gravity = -1.f;  // half-written tables belong to no length: a failure below forces a rebuild
    hipError_t e = hipStreamSynchronize(st);  // a step still in flight may be reading the old tables
    if (e == hipSuccess) {
        const unsigned ne = (unsigned)((N + 1) * (N + 1));
        hipLaunchKernelGGL(k_czt_tables, dim3((ne + 255) / 256), dim3(256), 0, st, N, length, gravity, z.Om, z.K);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e == hipSuccess) e = hipMemcpy(z.w1, w1.data(), sizeof(cf) * (N + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(z.w2, w2.data(), sizeof(cf) * N, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(z.Hh, Hh.data(), sizeof(cf) * z.M, hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    z.table_length = length;
    z.table_unit_width = unit_width;
    z.table_gravity = gravity;
    return hipSuccess;
}
#ifndef MW_CZT_FUSED_MAX_M
#define MW_CZT_FUSED_MAX_M 256  // grids whose second axis and assembly run in one launch (environment MW_CZT_FUSED=0: three launches, A/B)
#endif
constexpr int czt_fused_rows(int M) { return M <= 128 ? 8 : 4; }
template <int M>
static hipError_t czt_launch_rows_assemble(const CztArgs& A, cf* hds, float* dv, float* dn, float* dw, int white_stride, hipStream_t st) {
    constexpr int P = czt_points(M), RW = czt_fused_rows(M), NG = 3 * RW + 2, LB = (CztTw<M, P>::CF + NG * (FftGeom<M, P>::LBUF + 4)) * (int)sizeof(cf);
    static_assert(NG * (M / P) <= 1024 && LB <= 160 * 1024, "fused small-grid launch: geometry");
    static AttrOnce attr;
    hipError_t e = attr.set(reinterpret_cast<const void*>(&k_czt_rows_assemble<M, P, RW>), LB);
    if (e != hipSuccess) return e;
    k_czt_rows_assemble<M, P, RW><<<dim3((A.rows + RW - 1) / RW), dim3(NG * M / P), LB, st>>>(A, hds, dv, dn, dw, white_stride);
    return hipGetLastError();
}
// N <= MW_CZT_ONE_MAX_N (transform size 64): one launch (MW_CZT_ONE=0, read per call: the two-launch plan, for A/B and the bit-identity test)
static inline bool czt_one_launch(const CztState& z, int N) {
    return z.M == 64 && N <= MW_CZT_ONE_MAX_N && sw(SW_CZT_ONE) != 0 && sw(SW_CZT_FUSED) != 0;
}
// which launches a chirp-z step is: the ONE place that decides (czt_evaluate and the measurement hook's kernel names both ask here)
enum CztPlan { CZT_PLAN_ONE, CZT_PLAN_TWO, CZT_PLAN_THREE };
static inline CztPlan czt_plan(const CztState& z, int N) {
    if (czt_one_launch(z, N)) return CZT_PLAN_ONE;
    return (sw(SW_CZT_FUSED) != 0 && z.M <= MW_CZT_FUSED_MAX_M) ? CZT_PLAN_TWO : CZT_PLAN_THREE;
}
static hipError_t czt_launch_one(const CztArgs& A, cf* hds, float* dv, float* dn, float* dw, int white_stride, hipStream_t st) {
    constexpr int M = 64, P = czt_points(M), T = M / P, BUF = FftGeom<M, P>::LBUF + 4;
    const int N = A.C.N, lines = MW_CZT_PLANES * (N + 1), nthreads = ((lines * T + 63) / 64) * 64;
    const size_t cf_count = (size_t)CztTw<M, P>::CF + (((size_t)MW_CZT_PLANES * N * (N + 1) + 1) & ~(size_t)1) + (size_t)(nthreads / T) * BUF;
    const int LB = (int)(cf_count * sizeof(cf));
    if (nthreads > 1024 || LB > 160 * 1024) return hipErrorInvalidValue;
    static AttrOnce attr;
    hipError_t e = attr.set(reinterpret_cast<const void*>(&k_czt_one<M, P>), 160 * 1024);
    if (e != hipSuccess) return e;
    k_czt_one<M, P><<<dim3(1), dim3(nthreads), LB, st>>>(A, hds, dv, dn, dw, white_stride);
    return hipGetLastError();
}
template <int M>
static hipError_t czt_launch(const CztArgs& A, hipStream_t st) {
    constexpr int P = czt_points(M), RW = czt_rows(M), LB = (CztTw<M, P>::CF + RW * (FftGeom<M, P>::LBUF + 4)) * (int)sizeof(cf);
    static AttrOnce attr;
    hipError_t e = attr.set(reinterpret_cast<const void*>(&k_czt<M, P, RW>), LB);
    if (e != hipSuccess) return e;
    k_czt<M, P, RW><<<dim3((A.rows + RW - 1) / RW, MW_CZT_PLANES), dim3(RW * M / P), LB, st>>>(A);
    return hipGetLastError();
}
// ev (measurement hook): ev[0], ev[1] before and ev[2] after the two k_czt launches (the spectrum is formed inside the first one)
static inline hipError_t czt_evaluate(DirectState& d, OceanConsts C, const cf* h0, const cf* h0c, float t, float* dv, float* dn, float* dw,
                                      int white_stride, hipStream_t st, hipEvent_t* ev = nullptr) {
    CztState& z = d.czt;
    const int N = C.N;
    const unsigned nb = (unsigned)(((size_t)N * N + 127) / 128);
    if (z.table_length != C.length || z.table_unit_width != C.unit_width || z.table_gravity != C.gravity) {  // safety net only: the tables are uploaded at creation and at a
        hipError_t e = czt_upload_tables(z, N, C.unit_width, C.length, C.gravity, st);    // length change (direct_prepare_tables), never inside an enqueue
        if (e != hipSuccess) return e;
    }
    const CztPlan plan = czt_plan(z, N);
    const bool fused = plan == CZT_PLAN_TWO;
    if (ev) { hipEventRecord(ev[0], st); hipEventRecord(ev[1], st); }
    CztArgs A;
    A.w1 = z.w1; A.w2 = z.w2; A.Hh = z.Hh; A.TWf = z.TWf; A.TWi = z.TWi; A.Om = z.Om; A.K = z.K;
    A.nin = N + 1; A.nout = N;  // the packed planes live on the index set [0, N]^2 (czt_packed_value)
    if (plan == CZT_PLAN_ONE) {  // tiny grids: both axes and the assembly in one workgroup (k_czt_one)
        A.h0 = h0; A.h0c = h0c; A.t = t; A.C = C;
        if (ev) hipEventRecord(ev[2], st);
        return czt_launch_one(A, d.hds, dv, dn, dw, white_stride, st);
    }
    hipError_t e = hipSuccess;
    for (int pass = 0; pass < 2 && e == hipSuccess; pass++) {
        // along j: rows i = 0 .. N formed from the spectrum -> TT[p][b][i] (N rows of N + 1); along i: rows b -> O[p][a][b]
        A.in = z.TT;
        A.out = pass == 0 ? z.TT : z.O;
        A.h0 = pass == 0 ? h0 : nullptr; A.h0c = h0c; A.t = t; A.C = C;
        A.rows = pass == 0 ? N + 1 : N;
        A.in_ld = N + 1; A.in_plane = (long long)N * (N + 1);
        A.out_ld = pass == 0 ? N + 1 : N; A.out_plane = pass == 0 ? (long long)N * (N + 1) : (long long)N * N;
        if (pass == 1 && fused) {  // small grids: the second axis and the assembly in one launch, no plane O
            if (ev) hipEventRecord(ev[2], st);
            switch (z.M) {
                case 64: return czt_launch_rows_assemble<64>(A, d.hds, dv, dn, dw, white_stride, st);
                case 128: return czt_launch_rows_assemble<128>(A, d.hds, dv, dn, dw, white_stride, st);
                default: return czt_launch_rows_assemble<256>(A, d.hds, dv, dn, dw, white_stride, st);
            }
        }
        switch (z.M) {
            case 64: e = czt_launch<64>(A, st); break;
            case 128: e = czt_launch<128>(A, st); break;
            case 256: e = czt_launch<256>(A, st); break;
            case 512: e = czt_launch<512>(A, st); break;
            case 1024: e = czt_launch<1024>(A, st); break;
            case 2048: e = czt_launch<2048>(A, st); break;
            case 4096: e = czt_launch<4096>(A, st); break;
            default: return hipErrorInvalidValue;
        }
    }
    if (e != hipSuccess) return e;
    if (ev) hipEventRecord(ev[2], st);
    hipLaunchKernelGGL(k_czt_assemble_white, dim3(nb), dim3(128), 0, st, C, z.O, d.hds, dv, dn, dw, white_stride);
    return hipGetLastError();
}

static inline void direct_free(DirectState& d) {
    hipFree(d.A1); hipFree(d.T); hipFree(d.B1re); hipFree(d.B1im); hipFree(d.A2re); hipFree(d.A2im); hipFree(d.out); hipFree(d.hds);
    czt_free(d.czt);
    d = DirectState();
}
static inline int direct_alloc(DirectState& d, int N, hipStream_t st) {
    // The chirp-z form is the default wherever one workgroup holds the transform (N + 1 inputs, N outputs: 2N <= 4096): first hardware
    // run in round 4 -- 2.4x (N = 100) to 5.8x (N = 1000) faster than the GEMM form and an order of magnitude more accurate on
    // grids with large phases (profiles/r04a_bench_direct_*).  MW_DIRECT_CZT=0 selects the GEMM form (A/B, and the path of larger N).
    if (sw(SW_DIRECT_CZT) != 0 && czt_size(N) != 0) {
        if (czt_alloc(d.czt, N) != 0 || hipMalloc((void**)&d.hds, sizeof(cf) * (size_t)N * N) != hipSuccess) { direct_free(d); return 4; }
        d.N = N;
        d.use_czt = true;
        return 0;
    }
    d.Np = (N + 63) / 64 * 64;
    const size_t P2 = (size_t)d.Np * d.Np;
    struct { float** p; size_t n; } bufs[] = {{&d.A1, 10 * P2}, {&d.T, 10 * P2}, {&d.B1re, 2 * P2}, {&d.B1im, 2 * P2},
                                              {&d.A2re, 2 * P2}, {&d.A2im, 2 * P2}, {&d.out, 5 * P2}};
    for (auto& b : bufs) {
        if (hipMalloc((void**)b.p, sizeof(float) * b.n) != hipSuccess) { direct_free(d); return 4; }
        // the zero padding, ordered on the handle's stream: the table / spectrum kernels that fill the live part run there later
        // (a null-stream hipMemset is not ordered with a non-blocking stream)
        if (hipMemsetAsync(*b.p, 0, sizeof(float) * b.n, st) != hipSuccess) { direct_free(d); return 4; }
    }
    if (hipMalloc((void**)&d.hds, sizeof(cf) * (size_t)N * N) != hipSuccess) { direct_free(d); return 4; }
    d.N = N;
    return 0;
}
// tables that depend on (unit_width, length) but not on t: built when the handle is created and when its length changes
static inline hipError_t direct_prepare_tables(DirectState& d, int N, float unit_width, float length, float gravity, hipStream_t st) {
    if (d.use_czt) return czt_upload_tables(d.czt, N, unit_width, length, gravity, st);
    return hipSuccess;  // GEMM form: k_direct_tables runs on the stream, inside the first enqueue after a change (asynchronous)
}
// flop of one step as the GEMMs execute it (padded), and algorithmically (60 N^3)
static inline double direct_flops_padded(const DirectState& d) { return 60.0 * (double)d.Np * d.Np * d.Np; }

static inline hipError_t direct_gemm(const float* A, const float* B, float* C, int M, int Nc, int K, int lda, int ldb, int ldc,
                                     long long sA, long long sB, long long sC, int batch, hipStream_t st) {
    // the kernel has no bounds checks: every operand must be padded to whole tiles and 16-byte aligned rows
    if (M % MW_GEMM_BM || Nc % MW_GEMM_BN || K % MW_GEMM_BK || lda % 4 || ldb % 4 || batch < 1) return hipErrorInvalidValue;
    k_gemm_f32_mfma<<<dim3(Nc / MW_GEMM_BN, M / MW_GEMM_BM, batch), dim3(256), 0, st>>>(A, B, C, K, lda, ldb, ldc, sA, sB, sC);
    return hipGetLastError();
}

// ev (measurement hook, mw_ocean_profile_kernels): three events recorded before the spectrum kernel, before and after the GEMMs
static inline hipError_t direct_evaluate(DirectState& d, OceanConsts C, const cf* h0, const cf* h0c, float t, float* dv,
                                         float* dn, float* dw, int white_stride, hipStream_t st, hipEvent_t* ev = nullptr) {
    if (d.use_czt) return czt_evaluate(d, C, h0, h0c, t, dv, dn, dw, white_stride, st, ev);
    const int N = C.N, Np = d.Np;
    const unsigned nb = (unsigned)(((size_t)N * N + 127) / 128);
    const long long P2 = (long long)Np * Np;
    if (d.table_length != C.length || d.table_unit_width != C.unit_width) {  // E does not depend on t: once per handle / length
        hipLaunchKernelGGL(k_direct_tables, dim3(nb), dim3(128), 0, st, N, Np, C.length, C.unit_width, d.B1re, d.B1im, d.A2re, d.A2im);
        d.table_length = C.length;
        d.table_unit_width = C.unit_width;
    }
    if (ev) hipEventRecord(ev[0], st);
    hipLaunchKernelGGL(k_direct_spec, dim3(nb), dim3(128), 0, st, C, Np, h0, h0c, t, d.A1);
    if (ev) hipEventRecord(ev[1], st);
    hipError_t e;
    // step 1: Tr_f = A1_f B1re, Ti_f = A1_f B1im   (5 fields per launch; T_f = [Tr_f ; Ti_f])
    if ((e = direct_gemm(d.A1, d.B1re, d.T, Np, Np, 2 * Np, 2 * Np, Np, Np, 2 * P2, 0, 2 * P2, 5, st)) != hipSuccess) return e;
    if ((e = direct_gemm(d.A1, d.B1im, d.T + P2, Np, Np, 2 * Np, 2 * Np, Np, Np, 2 * P2, 0, 2 * P2, 5, st)) != hipSuccess) return e;
    // step 2: H = A2re T_0;  Dx, Dz, Sx, Sz = A2im T_1..4
    if ((e = direct_gemm(d.A2re, d.T, d.out, Np, Np, 2 * Np, 2 * Np, Np, Np, 0, 0, 0, 1, st)) != hipSuccess) return e;
    if ((e = direct_gemm(d.A2im, d.T + 2 * P2, d.out + P2, Np, Np, 2 * Np, 2 * Np, Np, Np, 0, 2 * P2, P2, 4, st)) != hipSuccess) return e;
    if (ev) hipEventRecord(ev[2], st);
    hipLaunchKernelGGL(k_direct_assemble, dim3(nb), dim3(128), 0, st, C, Np, d.out, d.hds, dv, dn);
    hipLaunchKernelGGL(k_direct_white, dim3(nb), dim3(128), 0, st, N, d.hds, dn, dw, white_stride);
    return hipGetLastError();
}
#endif

}  // namespace mw
