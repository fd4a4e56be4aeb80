// mistral_water.hip -- libmistral_water.so: gfx950 kernels + the C ABI of include/mistral_water.h.
//
// HIP-only product path: there is no CPU fallback anywhere in this file.  The CPU oracle lives in
// /oracle and is never linked here.
#include <hip/hip_runtime.h>

#include <dlfcn.h>  // tiles.inc binds RCCL at run time (dlopen / dlsym)

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/mistral_water.h"
#include "../../include/mistral_water_hooks.h"
#include "fftmesh_kernels.h"
#include "ocean_renderer_device.h"
#include "direct_kernels.h"
#include "gerstner_kernels.h"
#include "pond_kernels.h"

#ifndef MW_LATENCY_PLAN
#define MW_LATENCY_PLAN 1  // single-step enqueues at 512^2 / 1024^2 (mw_frame_plan_n): k_pass1<.., FS> + k_pass2_frame (launch_pass*_n)
#endif
#ifndef MW_LATENCY_PF
#define MW_LATENCY_PF 2  // prefetch level of the frame plan's pass 2 (k_pass2_hs<.., VT = 1, PF>)
#endif
#ifndef MW_WAVES_P1
#define MW_WAVES_P1 6  // min waves per SIMD the register allocator must leave room for (measured best)
#endif
#ifndef MW_WAVES_P2
#define MW_WAVES_P2 6
#endif

using namespace mw;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static mw_status fail(mw_status s, const std::string& m) {
    g_err = m;
    return s;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(MW_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
    } while (0)

// ------------------------------------------------------------------------------------------------
// __global__ wrappers: FFTMesh semantics
// ------------------------------------------------------------------------------------------------
__global__ void k_spectrum(int N, float length, float wind_x, float wind_y, float amplitude, float gravity,
                           uint64_t seed, cf* h0, cf* h0c) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    spectrum_element(N, length, wind_x, wind_y, amplitude, gravity, seed, idx / N, idx % N, h0, h0c);
}

__global__ void k_rest_mesh(int N, float unit_width, float* vertices, float* normals, float* uvs, int32_t* indices) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    rest_mesh_element(N, unit_width, idx / N, idx % N, vertices, normals, uvs, indices);
}

__global__ void k_prep(int N, float length, float gravity, const cf* h0, const cf* h0c, const cf* Wpre, f4* PQt,
                       f4* dPQ_i0, f4* dPQ_j0, float* Om) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    // idx enumerates the TRANSPOSED array so that the writes are the coalesced side
    prep_element(N, length, gravity, idx % N, idx / N, h0, h0c, Wpre, PQt, dPQ_i0, dPQ_j0, Om);
}

__global__ void k_omega_t(int N, float length, float gravity, float t, float* out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    out[idx] = omega_t_f32(N, length, gravity, idx / N, idx % N, t);
}

#if defined(MW_TIMING) && !defined(MW_LAB)
#error "MW_TIMING (cycle stamps inside the pass kernels) is a lab build option: add -DMW_LAB (tools/build_variant.sh does)"
#endif
#ifdef MW_TIMING
#ifndef MW_STAMP_STEP
#define MW_STAMP_STEP 3
#endif
__device__ long long g_stamps[2][64][16][32];  // [kernel][block slot][wave][stamp]
#define MW_STAMP(K, id)                                                                               \
    do {                                                                                              \
        if ((blockIdx.x % 37) == 5 && blockIdx.x / 37 < 64 && step == MW_STAMP_STEP && (threadIdx.x & 63) == 0 && threadIdx.x < 1024) \
            g_stamps[K][blockIdx.x / 37][threadIdx.x >> 6][id] = __builtin_readcyclecounter();        \
    } while (0)
// the constant 100-MHz clock beside the cycle counter (slots 30 / 31: start / end of a kernel): calibrates cycles against kernel time
#define MW_STAMP_RT(K, id)                                                                            \
    do {                                                                                              \
        if ((blockIdx.x % 37) == 5 && blockIdx.x / 37 < 64 && step == MW_STAMP_STEP && (threadIdx.x & 63) == 0 && threadIdx.x < 1024) \
            g_stamps[K][blockIdx.x / 37][threadIdx.x >> 6][id] = __builtin_amdgcn_s_memrealtime();    \
    } while (0)
// where the 4-wave workgroups of a pass-1 launch ran: HW_ID / XCC_ID of EVERY workgroup b < 768, parked in the unused wave slots 4..15
#define MW_STAMP_HWID(K)                                                                              \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < 768 && step == MW_STAMP_STEP) {                          \
            g_stamps[K][blockIdx.x % 64][4 + blockIdx.x / 64][0] = __builtin_amdgcn_s_getreg(63492);   \
            g_stamps[K][blockIdx.x % 64][4 + blockIdx.x / 64][1] = __builtin_amdgcn_s_getreg(63508);   \
            g_stamps[K][blockIdx.x % 64][4 + blockIdx.x / 64][2] = __builtin_amdgcn_s_memrealtime();    \
        }                                                                                             \
    } while (0)
#define MW_STAMP_HWID_END(K)                                                                          \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < 768 && step == MW_STAMP_STEP)                            \
            g_stamps[K][blockIdx.x % 64][4 + blockIdx.x / 64][3] = __builtin_amdgcn_s_memrealtime();    \
    } while (0)
#elif defined(MW_SCHED_FENCE)
#define MW_STAMP(K, id) __builtin_amdgcn_sched_barrier(0)
#define MW_STAMP_RT(K, id) do { } while (0)
#define MW_STAMP_HWID(K) do { } while (0)
#define MW_STAMP_HWID_END(K) do { } while (0)
#else
#define MW_STAMP(K, id) do { } while (0)
#define MW_STAMP_RT(K, id) do { } while (0)
#define MW_STAMP_HWID(K) do { } while (0)
#define MW_STAMP_HWID_END(K) do { } while (0)
#endif

// LDS layout of both pass kernels: [twiddle tables, if small] [NBUF sets of exchange buffers].  With
// NBUF == 2 the sets are used ping-pong (every store goes to the set the previous load did NOT read), so
// one barrier per exchange suffices; with NBUF == 1 a second (WAR) barrier follows every load.
// VT = virtual threads per lane (see k_pass2_hs): the phase functions are written for 4*T virtual threads (4 spectrum
// columns x T); a workgroup of 4*T/VT lanes runs virtual threads tid, tid + NT, ... of every phase back to back.
// issue priority (s_setprio, 0..3) of the row groups by field once the loads are out: the slope groups -- the longest fetch, then the
// normals to store -- ahead of displacement and halo row, the height groups (which only wait for hds after their transform) last.
// Measured on top of the wave-level exchanges: pass 2 of a lone step 17.1 -> 15.6 us (the reverse order 16.6; profiles/r04_ab_notes.md).
#ifndef MW_FRAME_PRIO_S
#define MW_FRAME_PRIO_S 3
#endif
#ifndef MW_FRAME_PRIO_D
#define MW_FRAME_PRIO_D 2
#endif
#ifndef MW_FRAME_PRIO_X
#define MW_FRAME_PRIO_X 2
#endif
#ifndef MW_FRAME_PRIO_H
#define MW_FRAME_PRIO_H 1
#endif
__device__ __forceinline__ void mw_setprio(int p) {  // the builtin wants a literal
    switch (p) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}
#ifndef MW_P1_FRAME_WAVE_SYNC
#define MW_P1_FRAME_WAVE_SYNC 1  // single-step plan: a column's exchanges stay inside its own wave up to the last one
#endif
#ifndef MW_P1_FRAME_PRIO
#define MW_P1_FRAME_PRIO 0       // single-step plan: issue priority by field (1: f = 0 highest; 2: f = 1 highest; 3: f = 2 highest)
#endif
// FS = the single-step (frame-at-a-time) instantiation: one FIELD per workgroup (A.field_split says which grid decodes it)
template <int N, int P, int VT, bool FS = false>
__global__ __launch_bounds__((P1Geom<N, P>::NTHREADS / VT))
__attribute__((amdgpu_waves_per_eu(VT > 1 ? P1Geom<N, P>::NTHREADS / VT / 256 : (P == 8 ? MW_WAVES_P1 : 4)))) void k_pass1(P1Args A, StepTimes times) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = P1Geom<N, P>;
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int T = FftGeom<N, P>::T, NT = G::NTHREADS / VT;
    static_assert(G::NTHREADS % VT == 0 && (VT == 1 || NT % T == 0), "a lane's virtual threads must belong to whole columns");
    const int tid = threadIdx.x;
    int jb = blockIdx.x, step = blockIdx.y;
    // Frame-at-a-time plan (A.field_split, single-step enqueues): grid (column jobs, 3) -- one FIELD per workgroup instead of
    // the three one after the other, each workgroup re-forming the (cheap) animated spectrum.  A step is 257 workgroups at
    // 1024^2 where the device has 1024 slots, so its latency is that of ONE workgroup: a third of the work each cuts it
    // accordingly.  The arithmetic of a field does not depend on which workgroup runs it: same bits as the batched plan.
    int f_lo = 0, f_hi = 3;
    if constexpr (FS) {
        if (A.field_split == 2) {  // 1-D grid over the list of active (column job, field) pairs (p1_frame_jobs)
            const int job = A.jobs[blockIdx.x];
            if (job < 0) return;
            jb = job & 0xffff;
            f_lo = job >> 16;
        } else {
            f_lo = (int)blockIdx.y;
        }
        f_hi = f_lo + 1;
        step = 0;
        if (!p1_field_active(N, jb, f_lo, G::CW)) return;  // block-uniform, before any barrier
        if (MW_P1_FRAME_PRIO) mw_setprio(3 - (f_lo + 4 - MW_P1_FRAME_PRIO) % 3);
    } else if (A.tgroup > 0 && !p1_block_map((int)blockIdx.x, G::GRID_X, A.nsteps, A.tgroup, &jb, &step)) return;
    // Up to its last exchange a column's buffer is written and read by the column's own T threads: where those are one wave (the
    // single-step plan at T == 64) the exchanges need that wave's LDS operations in order and no workgroup barrier -- the four columns
    // drift apart; the last exchange feeds the column-interleaved final pass and keeps the barrier.
    constexpr bool WS = FS && MW_P1_FRAME_WAVE_SYNC && VT == 1 && T == 64;
    auto col_sync = [&](bool whole_group) {
        if (WS && !whole_group) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    (void)col_sync;
    const float t = times.t[step];
    TwStage<N, P, NT> tws;
    if (TwGeom<N, P>::LDS_ALL) tws.load(lds, A.TW, tid);  // in LDS behind the spectrum requests, visible after the first barrier
    const Twiddles tw = TwGeom<N, P>::view(A.TW, lds);
    cf* set0 = lds + G::TW_LDS;
    int cur = 0;  // set the next store goes to
    P1State<P> st[VT];
    cf x[VT][P];
#define MW_VT(h) for (int h = 0; h < VT; h++)
#define MW_BUF(h) (set0 + cur * G::SETSTRIDE + ((tid + (h) * NT) / T) * G::BUFSTRIDE)
#define MW_U(h) ((tid + (h) * NT) % T)
    MW_STAMP(0, 0);
    MW_STAMP_RT(0, 30);
    if constexpr (FS) MW_STAMP_HWID(0);
#pragma unroll
    MW_VT(h) p1_animate<N, P>(A, jb, tid + h * NT, t, st[h]);
    if (TwGeom<N, P>::LDS_ALL) tws.store(lds, tid);
    MW_STAMP(0, 1);
#pragma unroll
    for (int f = 0; f < 3; f++) {
        if (f < f_lo || f >= f_hi) continue;              // block-uniform: the frame-at-a-time plan runs one field per workgroup
        if (!p1_field_active(N, jb, f, G::CW)) continue;  // block-uniform: height needs columns j <= N/2 only
#pragma unroll
        MW_VT(h) p1_build<N, P>(A, jb, tid + h * NT, f, st[h], x[h]);
        if (G::NBUF == 1 && f) __syncthreads();
        MW_STAMP(0, 2 + 8 * f);
#pragma unroll
        MW_VT(h) stage0_store<N, P, +1>(x[h], MW_U(h), MW_BUF(h));
        MW_STAMP(0, 3 + 8 * f);
        __syncthreads();  // the first barrier of the kernel also publishes the staged twiddle tables: always the whole group
#pragma unroll
        for (int s = 1; s < p1_mid_passes<N, P>(); s++) {
#pragma unroll
            MW_VT(h) load_slots<N, P>(x[h], MW_U(h), MW_BUF(h), s - 1);
            if (s == 1) MW_STAMP(0, 4 + 8 * f);
            if (G::NBUF == 1) col_sync(false); else cur ^= 1;
            if (s == 1) MW_STAMP(0, 5 + 8 * f);
#pragma unroll
            MW_VT(h) stage_store<N, P, +1>(x[h], MW_U(h), MW_BUF(h), tw, s);
            if (s == 1) MW_STAMP(0, 6 + 8 * f);
            col_sync(s == p1_mid_passes<N, P>() - 1);
        }
        MW_STAMP(0, 7 + 8 * f);
#pragma unroll
        MW_VT(h) p1_finish<N, P>(A, tw, jb, step, tid + h * NT, f, x[h], set0 + cur * G::SETSTRIDE);
        if (G::NBUF == 2) cur ^= 1;
        MW_STAMP(0, 8 + 8 * f);
    }
    MW_STAMP(0, 26);
    MW_STAMP_RT(0, 31);
    if constexpr (FS) MW_STAMP_HWID_END(0);
#undef MW_VT
#undef MW_BUF
#undef MW_U
}

#ifndef MW_XCD_GROUP
#define MW_XCD_GROUP 8  // adjacent row blocks kept on one XCD (1 = plain round-robin); 8-32: pass 2 -3 % at steady clocks
#endif
template <int NBLK>
__device__ __forceinline__ int p2_row_block(int b) {
    constexpr int XG = MW_XCD_GROUP;
    const int xcd = b % 8, cidx = b / 8;
    return (XG > 1 && NBLK % (8 * XG) == 0) ? (cidx / XG) * (8 * XG) + xcd * XG + (cidx % XG) : b;
}

template <int N, int P, int R2, bool DUMP = false>
__global__ __launch_bounds__((P2Geom<N, P, R2>::NTHREADS)) __attribute__((amdgpu_waves_per_eu(P == 8 ? MW_WAVES_P2 : 3))) void k_pass2(
    P2Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = P2Geom<N, P, R2>;
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int T = FftGeom<N, P>::T;
    const int tid = threadIdx.x, step = blockIdx.y;
    // XCD-aware row-block mapping: the dispatcher places block b on XCD b % 8 (speed only, never correctness); giving
    // each XCD a contiguous range of row blocks makes a block's halo row (the first row of the NEXT block) a hit in
    // the same XCD's L2 instead of a second 128-B line fill across the fabric.
    const int ab = p2_row_block<N / R2>((int)blockIdx.x);
    const int g = tid / T;
    TwStage<N, P, G::NTHREADS> tws;
    if (TwGeom<N, P>::LDS_ALL) tws.load(lds, A.TW, tid);
    const Twiddles tw = TwGeom<N, P>::view(A.TW, lds);
    cf* set0 = lds + G::TW_LDS;
    float* noise_lds = reinterpret_cast<float*>(lds + G::NOISE_OFF);
    int cur = 0;
    P2State<P> st;
    cf x[P];
#ifdef MW_P2_PREFETCH
    cf xn[P];
    if (p2_active<N, P, R2>(ab, tid, p2_field(0))) p2_fetch<N, P, R2>(A, ab, step, tid, p2_field(0), xn);
#endif
    MW_STAMP(1, 0);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int f = p2_field(k);
        const bool active = p2_active<N, P, R2>(ab, tid, f);
        if (G::NBUF == 1 && k) __syncthreads();
        MW_STAMP(1, 1 + 8 * k);
#ifdef MW_P2_PREFETCH
        if (active) {
#pragma unroll
            for (int q = 0; q < P; q++) x[q] = xn[q];
            p2_stage0<N, P, R2>(tid, x, set0 + cur * G::SETSTRIDE);
        }
        if (k < 2 && p2_active<N, P, R2>(ab, tid, p2_field(k + 1)))  // next field's rows fly during this field's passes
            p2_fetch<N, P, R2>(A, ab, step, tid, p2_field(k + 1), xn);
#else
        if (active) p2_load<N, P, R2>(A, ab, step, tid, f, x, set0 + cur * G::SETSTRIDE);
#endif
        if (k == 0 && TwGeom<N, P>::LDS_ALL) tws.store(lds, tid);  // published by the barrier below
        MW_STAMP(1, 2 + 8 * k);
        __syncthreads();
#pragma unroll
        for (int s = 1; s < FftGeom<N, P>::S; s++) {
            constexpr bool LIR = LastStays<N, P>::value && G::NBUF == 1;
            const bool in_regs = LIR && s == FftGeom<N, P>::S - 1;  // the last pass writes nothing to LDS: no barrier on either side of it
            if (active) p2_mid_load<N, P, R2>(tid, s, x, set0 + cur * G::SETSTRIDE);
            if (!in_regs) { if (G::NBUF == 1) __syncthreads(); else cur ^= 1; }
            if (active) p2_mid_store<N, P, R2>(tw, tid, s, x, set0 + cur * G::SETSTRIDE);
            if (!in_regs) __syncthreads();
        }
        MW_STAMP(1, 6 + 8 * k);
        if (active) p2_finish<N, P, R2>(A, tw, ab, step, tid, f, x, st, set0 + cur * G::SETSTRIDE, noise_lds);
        if (G::NBUF == 2) cur ^= 1;
        MW_STAMP(1, 7 + 8 * k);
    }
    if (G::NBUF == 1) __syncthreads();
    MW_STAMP(1, 25);
    if (p2_active<N, P, R2>(ab, tid, 1)) p2_publish_hds<N, P, R2>(tid, st, set0 + cur * G::SETSTRIDE);
    __syncthreads();
    if constexpr (DUMP) p2_dump_hds<N, P, R2>(A, ab, step, tid, G::NTHREADS, set0 + cur * G::SETSTRIDE);  // test hook
    MW_STAMP(1, 26);
    if (g < R2) p2_epilogue<N, P, R2>(A, ab, step, tid, st, set0 + cur * G::SETSTRIDE, noise_lds);
    MW_STAMP(1, 27);
}

// Pass 2, sequential-halo variant (large N): R2 row groups, no halo group.  Height and displacement fields first; then
// the vertices leave, every row publishes hds, rows 0..R2-2 form 1 - J, group 0 transforms the halo row in buffer 0, row R2-1 follows;
// the slope field comes last and its final pass writes normals and whitecap together.
//
// VT = "virtual threads" per lane: the phase functions are written for R2*T virtual threads; a workgroup of R2*T/VT
// lanes runs virtual threads tid, tid + NT, ... of every phase back to back.  With VT = 2 a 4096-point, 4-row block is 8
// waves instead of 16: each lane owns 2 x 16 points, the register budget doubles to 256 (the 16-wave form spilled 29
// dwords = 14 B of scratch traffic per grid point at its 128), the two independent rows of a lane give the scheduler two
// instruction streams to interleave, and every barrier joins half as many waves.
//
// mw_fresh (opt-in, -DMW_FRESH): every phase derives its lane indices from an opaque copy of the thread index, so that the
// compiler cannot hoist the index arithmetic of all phases to the top of the kernel; it removes the spills of the 16-wave
// form but the asm statements are scheduling barriers and the kernel gets 10-15 % slower: off.
__device__ __forceinline__ int mw_fresh(int v) {
#ifdef MW_FRESH
    asm volatile("" : "+v"(v));
#endif
    return v;
}
#ifndef MW_HS_JAC_FROM_LDS
#define MW_HS_JAC_FROM_LDS 99  // points per thread from which the Jacobian re-reads its own row from LDS (d[] dead): off
#endif
// minimum waves per SIMD the register allocator must leave room for: as many workgroups per CU as the LDS admits (at most 2)
constexpr int hs_min_waves(int nthreads, int lds_bytes) {
    const int wgs = (2 * lds_bytes <= 160 * 1024) ? 2 : 1;
    const int w = nthreads / 64 * wgs / 4;
    return w < 1 ? 1 : (w > 8 ? 8 : w);
}
template <int N, int P, int R2, int VT, bool DUMP = false, int PF = 0>
__global__ __launch_bounds__((P2Geom<N, P, R2, true>::NTHREADS / VT))
__attribute__((amdgpu_waves_per_eu(hs_min_waves(P2Geom<N, P, R2, true>::NTHREADS / VT, P2Geom<N, P, R2, true>::LDS_BYTES)))) void k_pass2_hs(P2Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = P2Geom<N, P, R2, true>;
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int T = FftGeom<N, P>::T, NT = G::NTHREADS / VT;  // NT lanes, each running VT virtual threads
    static_assert(G::NTHREADS % VT == 0 && NT % T == 0, "a lane's virtual threads must belong to distinct whole row groups");
    static_assert(T % 64 == 0, "row groups must be whole waves: the row group of a lane is treated as wave-uniform (512^2 at 16 points fails parity)");
    const int tid0 = threadIdx.x, step = blockIdx.y;
#ifdef MW_HS_NO_XCD_MAP
    const int ab = blockIdx.x;
#else
    const int ab = p2_row_block<N / R2>((int)blockIdx.x);  // neighbouring row blocks (halo rows, shared 128-B lines) on one XCD
#endif
    const int g0 = wave_uniform<true>(tid0 / T);  // row group of virtual thread 0; virtual thread h is in group g0 + h * NT / T
    TwStage<N, P, NT> tws;
    if (TwGeom<N, P>::LDS_ALL) tws.load(lds, A.TW, tid0);
    const Twiddles tw = TwGeom<N, P>::view(A.TW, lds);
    cf* set0 = lds + G::TW_LDS;
    P2StateHS<P> st[VT];
    cf x[VT][P];
#ifndef MW_HS_HALO_EARLY
#define MW_HS_HALO_EARLY 1
#endif
#ifndef MW_HS_HALO_EARLY_2048
#define MW_HS_HALO_EARLY_2048 1  // fits (251 VGPRs) since the radix-8 final pass forms its twiddles by powers (MW_TF_POWERS_MIN_RL)
#endif
#ifndef MW_HS_HALO_EARLY_4096
#define MW_HS_HALO_EARLY_4096 0  // 4096^2 prefetches the displacement rows during the height field instead (PF = 1): the two together spill
#endif
    // PF = 2 (the frame-at-a-time plan, one wave per row): EVERY exchange-buffer load of the workgroup -- height, displacement, the
    // stored half of the slope field, the halo row -- is requested before the first transform.  A single-step launch is one
    // workgroup per CU: nothing else hides a load.  Measured (profiles/r04_ab_notes.md): pass 2 of a lone step 19.5 -> 18.4 us.
    constexpr bool HALO_EARLY =
        ((N == 2048 ? (MW_HS_HALO_EARLY_2048 != 0) : (N >= 4096 ? (MW_HS_HALO_EARLY_4096 != 0) : (MW_HS_HALO_EARLY != 0))) && VT >= 2) || PF == 2;
    cf xh[HALO_EARLY ? P : 1];  // halo row data parked in registers across the displacement transform
    const int tid = tid0;  // MW_STAMP
    MW_STAMP(1, 0);
#define MW_VT(h) for (int h = 0; h < VT; h++)
#define MW_VTID(h) (mw_fresh(tid0) + (h) * NT)
    // PF = 1 (software prefetch; the 4096^2 plan of rounds 2-3, an A/B option since round 4: MW_PF_4096): the displacement field's exchange-buffer rows are
    // requested while the height field -- whose phase holds nothing but x -- is transformed, into a second register set.
    // Three things make the loads really asynchronous: the height field's own loads are issued first (scheduling fence;
    // vmcnt is in-order), the Nyquist-column term is added at use (p2_fetch's nyq), and no pass of the transform reads
    // global memory (TwGeom::PW_CF).  Measured: pass 2 -1.5 % at 4096^2 (round 2; with KeepT1 the plan without it is 1 % ahead), +1 % at 1024^2 and 2048^2.  Prefetching the
    // slope rows during the displacement transform as well (all of them: 17 spilled dwords; one virtual thread's: 243
    // VGPRs) made the kernel 4-11 % slower: removed.
    static_assert(PF == 0 || PF == 1 || PF == 2, "prefetch level");
    static_assert(PF != 2 || P2SlopeParts<N, P>::value, "PF = 2 parks the stored half of the slope field");
    constexpr bool SPARTS = P2SlopeParts<N, P>::value;
    // EARLY_SLOPES (experiment, off; profiles/r03_ab_notes.md): the slope field's stored half requested BEFORE the vertex stores
    // -- gfx950 counts stores in vmcnt, in order with loads -- and parked in x while the Jacobians are formed.  Measured: 4096^2
    // pass 2 7 % SLOWER (6173 -> 6629 us per 32 steps, 5 spilled dwords), 1024^2 40 % slower (256 VGPRs: one wave per SIMD):
    // the drain of the stores is not what the slope loads wait for.
#ifndef MW_EARLY_SLOPES_MIN_N
#define MW_EARLY_SLOPES_MIN_N 8192
#endif
    constexpr bool EARLY_SLOPES = SPARTS && N >= MW_EARLY_SLOPES_MIN_N;
    cf xn[PF ? VT : 1][PF ? P : 1], xn_nyq[PF ? VT : 1], xh_nyq = mk(0.f, 0.f);
    cf xs[PF == 2 ? VT : 1][PF == 2 ? P : 1];  // PF = 2: the stored half of the slope field, parked from the start
    constexpr bool KEEP = KeepT1<N, P>::value && P2SlopeParts<N, P>::value;
    cf t1m[KEEP ? VT : 1][KEEP ? P / 2 : 1];   // raw mirrored height-row values, from the height fetch to the slope assembly (KeepT1)
    (void)xn; (void)xn_nyq; (void)xs; (void)t1m;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int f = p2_hs_field(k);
        if (k != 0 || MW_TW_STAGE != 2) __syncthreads();  // the previous phase's LDS reads are done (k = 0: nothing to wait for -- the twiddle tables
                                                          // go to LDS behind the first requests below and are published by the barrier after stage 0)
        MW_STAMP(1, 1 + 8 * k);
        if (PF >= 1 && k == 1) {  // compile-time: k is unrolled
            if constexpr (PF != 0) {
#pragma unroll
                MW_VT(h) {
#pragma unroll
                    for (int q = 0; q < P; q++) x[h][q] = xn[h][q];
                    int r1, u1;
                    p2_load_map<N, P, R2>(MW_VTID(h), &r1, &u1);
                    if (u1 == 0) x[h][0] = x[h][0] + xn_nyq[h];
                }
            }
        } else if (f == 2 && SPARTS) {  // the slope half of every virtual thread in flight, then height rows + stage 0 one at a time
            if constexpr (PF == 2) {
#pragma unroll
                MW_VT(h)
#pragma unroll
                for (int q = 0; q < P; q++) x[h][q] = xs[h][q];
            } else if constexpr (!EARLY_SLOPES) {
#pragma unroll
                MW_VT(h) p2_fetch<N, P, R2, 1>(A, ab, step, MW_VTID(h), f, x[h]);
            }
        } else {
#pragma unroll
            MW_VT(h) p2_fetch<N, P, R2>(A, ab, step, MW_VTID(h), f, x[h], nullptr, (KEEP && f == 0) ? t1m[h] : nullptr);
        }
        if constexpr (PF != 0) {
            if (k == 0) {
                mw_sched_fence();
#pragma unroll
                MW_VT(h) p2_fetch<N, P, R2>(A, ab, step, MW_VTID(h), p2_hs_field(1), xn[h], &xn_nyq[h]);
                if constexpr (PF == 2) {
#pragma unroll
                    MW_VT(h) p2_fetch<N, P, R2, 1>(A, ab, step, MW_VTID(h), 2, xs[h]);
                }
            }
        }
        // The halo row's lines are the next row block's own lines: fetched while that block (same XCD, same phase) loads
        // them too, they are L2 hits; fetched two phases later they have left the L2 and cost a second 128-B fill per
        // 32-B piece (measured +6 B per grid point).  Needs 2P spare VGPRs across the displacement transform: VT >= 2.
        if constexpr (HALO_EARLY)
            if (k == (PF >= 1 ? 0 : 1) && g0 == 0 && ab * R2 + R2 < N)
                p2_hs_halo_fetch<N, P, R2>(A, ab, step, mw_fresh(tid0) % T, xh, PF >= 1 ? &xh_nyq : nullptr);
        if (k == 0 && TwGeom<N, P>::LDS_ALL) tws.store(lds, tid0);
        if (f == 2 && SPARTS) {
#pragma unroll
            MW_VT(h) {
                p2_fetch<N, P, R2, 2>(A, ab, step, MW_VTID(h), f, x[h], nullptr, KEEP ? t1m[h] : nullptr);
                p2_stage0<N, P, R2>(MW_VTID(h), x[h], set0);
#ifndef MW_SLOPE_STAGE_FENCE
#define MW_SLOPE_STAGE_FENCE 1
#endif
                if (MW_SLOPE_STAGE_FENCE) mw_sched_fence();
            }
        } else {
#pragma unroll
            MW_VT(h) p2_stage0<N, P, R2>(MW_VTID(h), x[h], set0);
        }
        MW_STAMP(1, 2 + 8 * k);
        __syncthreads();
#pragma unroll
        for (int s = 1; s < FftGeom<N, P>::S; s++) {
            const bool in_regs = mw_pass_in_regs<N, P>(s);  // the last pass writes nothing to LDS (LastInRegs / LastInWave)
#pragma unroll
            MW_VT(h) p2_mid_load<N, P, R2>(MW_VTID(h), s, x[h], set0);
            if (!in_regs) __syncthreads();
#pragma unroll
            MW_VT(h) p2_mid_store<N, P, R2>(tw, MW_VTID(h), s, x[h], set0);
            if (!in_regs) __syncthreads();
        }
        MW_STAMP(1, 6 + 8 * k);
        if (f == 2) {
#pragma unroll
            MW_VT(h) p2_hs_finish_slopes<N, P, R2>(A, tw, ab, step, MW_VTID(h), x[h], st[h], set0);
            MW_STAMP(1, 7 + 8 * k);
            break;
        }
#pragma unroll
        MW_VT(h) p2_hs_finish<N, P, R2>(tw, ab, MW_VTID(h), f, x[h], st[h], set0);
        MW_STAMP(1, 7 + 8 * k);
        if (f != 1) continue;
        // ---- displacement done: vertices, halo row, Jacobian ----
        if constexpr (EARLY_SLOPES) {
#pragma unroll
            MW_VT(h) p2_fetch<N, P, R2, 1>(A, ab, step, MW_VTID(h), 2, x[h]);
            mw_sched_fence();
        }
#pragma unroll
        MW_VT(h) p2_vertices<N, P, R2>(A, ab, step, MW_VTID(h), st[h]);
        MW_STAMP(1, 24);
        __syncthreads();  // every final-pass read of the displacement buffers is done
#pragma unroll
        MW_VT(h) p2_publish_hds<N, P, R2>(MW_VTID(h), st[h], set0);  // every row into its own buffer, plain index b
        __syncthreads();
        if constexpr (DUMP) p2_dump_hds<N, P, R2>(A, ab, step, tid0, NT, set0);  // test hook
        const bool has_halo = (ab * R2 + R2 < N);  // block-uniform
        // Rows 0..R2-2 have their (a+1) neighbour published already: they form 1 - J now.  Buffer 0 (row 0's copy) is
        // then free for the halo row's transform; row R2-1 waits for it and works from its own published copy, so that
        // nobody's d is live across the halo transform (P = 16: the transform alone takes ~100 VGPRs).
#pragma unroll
        MW_VT(h) {
            const int g = g0 + h * (NT / T);
            if (g != R2 - 1) {
                if (P >= MW_HS_JAC_FROM_LDS)
                    p2_hs_jacobian_lds<N, P, R2>(ab, MW_VTID(h), st[h], set0 + g * G::BUFSTRIDE, set0 + (g + 1) * G::BUFSTRIDE);
                else
                    p2_hs_jacobian<N, P, R2>(ab, MW_VTID(h), st[h], set0 + g * G::BUFSTRIDE, set0 + (g + 1) * G::BUFSTRIDE);
            }
        }
        __syncthreads();  // group 0 no longer reads its own row
        MW_STAMP(1, 25);
        if (has_halo) {  // group 0 = virtual thread 0 of the lanes below T
            const int u = mw_fresh(tid0) % T;
            cf xq[P];  // the halo row in registers of its own: the allocator no longer ties it to x[0] (248 -> 219 VGPRs at 1024^2)
            if (g0 == 0) {
                if constexpr (HALO_EARLY) {
#pragma unroll
                    for (int q = 0; q < P; q++) xq[q] = xh[q];
                    if (PF >= 1 && u == 0) xq[0] = xq[0] + xh_nyq;
                } else {
                    p2_hs_halo_fetch<N, P, R2>(A, ab, step, u, xq);
                }
                stage0_store<N, P, +1>(xq, u, set0);
            }
            __syncthreads();
#pragma unroll
            for (int s = 1; s < FftGeom<N, P>::S; s++) {
                const bool in_regs = mw_pass_in_regs<N, P>(s);
                if (g0 == 0) load_slots<N, P>(xq, u, set0, s - 1);
                if (!in_regs) __syncthreads();
                if (g0 == 0) { if (in_regs) stage_last_regs<N, P, +1>(xq, u, tw, s); else stage_store<N, P, +1>(xq, u, set0, tw, s); }  // g0 is wave-uniform: whole waves
                if (!in_regs) __syncthreads();
            }
            if (g0 == 0) {
                p2_last_load<N, P>(xq, u, set0);
                final_stage<N, P, +1>(xq, u, tw.TF);
            }
            __syncthreads();
            if (g0 == 0) p2_hs_halo_publish<N, P, R2>(ab, u, xq, set0);
            __syncthreads();
        }
        MW_STAMP(1, 26);
        if (g0 + (VT - 1) * (NT / T) == R2 - 1)  // the lane whose LAST virtual thread owns the block's last row
            p2_hs_jacobian_lds<N, P, R2>(ab, MW_VTID(VT - 1), st[VT - 1], set0 + (R2 - 1) * G::BUFSTRIDE, set0);
        MW_STAMP(1, 27);
    }
#undef MW_VT
#undef MW_VTID
}

// Pass 2 of a single-step enqueue (the frame-at-a-time plan, P2FrameGeom in fftmesh_kernels.h): 3 R2 + 1 row groups transform the
// three fields of the block's rows and the halo row at the same time; the latency of the workgroup -- which IS the latency of the
// step, 256 workgroups on 256 CUs -- is one transform instead of four.
#ifndef MW_FRAME_WAVE_SYNC
#define MW_FRAME_WAVE_SYNC 1  // the middle passes of a row stay inside its own wave: no workgroup barrier between them
#endif
template <int N, int P, int R2>
__global__ __launch_bounds__((P2FrameGeom<N, P, R2>::NTHREADS)) void k_pass2_frame(P2Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = P2FrameGeom<N, P, R2>;
    static_assert(G::OK, "frame variant: geometry");
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int T = G::T;
    const int tid = threadIdx.x, step = blockIdx.y;
    const int ab = p2_row_block<N / R2>((int)blockIdx.x);
    const int fg = wave_uniform<true>(tid / G::FT);  // 0 height, 1 displacement, 2 slopes, 3 the halo row (displacement of row a0 + R2)
    const int tl = tid - fg * G::FT;
    const bool has_halo = (ab * R2 + R2 < N);        // block-uniform
    const bool row = fg < 3, halo = (fg == 3) && has_halo;
    cf* set0 = lds + G::TW_LDS;
    cf* set_d = set0 + G::SETSTRIDE;
    cf* hbuf = set0 + 3 * G::SETSTRIDE;
    cf* mine = set0 + fg * G::SETSTRIDE;  // fg == 3: the halo row's buffer
    cf x[P];
    MW_STAMP(1, 0);
    MW_STAMP_RT(1, 30);
    TwStage<N, P, G::NTHREADS> tws;
    if (TwGeom<N, P>::LDS_ALL) tws.load(lds, A.TW, tid);  // requested first (vmcnt is in order), written to LDS behind the row requests
    if (row) p2_fetch<N, P, R2>(A, ab, step, tl, fg, x);
    else if (halo) p2_hs_halo_fetch<N, P, R2>(A, ab, step, tl, x);
    if (TwGeom<N, P>::LDS_ALL) tws.store(lds, tid);
    const Twiddles tw = TwGeom<N, P>::view(A.TW, lds);
    if (row) p2_stage0<N, P, R2>(tl, x, mine);
    else if (halo) stage0_store<N, P, +1>(x, tl, mine);
    MW_STAMP(1, 1);
    __syncthreads();  // stage 0 was written in the row-interleaved mapping of the loads: every wave of a field into all of its rows
    // From here to the final pass a row buffer belongs to ONE wave (exact layouts: row-major mapping, T == 64): its exchanges need the
    // wave's own LDS operations in order, nothing else -- the row groups drift apart, and the first to finish starts its stores while
    // the others still transform.
    static_assert((T == 64 || T == 32) && G::FT % 64 == 0, "a wave holds whole row groups of one field");
    constexpr bool WSYNC = MW_FRAME_WAVE_SYNC && XLay<N, P>::EXACT;  // (the padded layouts' middle passes run row-interleaved: barriers)
    auto row_sync = [&]() {
        if constexpr (WSYNC) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    mw_setprio(fg == 0 ? MW_FRAME_PRIO_H : (fg == 1 ? MW_FRAME_PRIO_D : (fg == 2 ? MW_FRAME_PRIO_S : MW_FRAME_PRIO_X)));
#pragma unroll
    for (int s = 1; s < FftGeom<N, P>::S; s++) {
        const bool in_regs = mw_pass_in_regs<N, P>(s);
        if (row) p2_mid_load<N, P, R2>(tl, s, x, mine);
        else if (halo) load_slots<N, P>(x, tl, mine, s - 1);
        if (!in_regs) row_sync();
        if (row) p2_mid_store<N, P, R2>(tw, tl, s, x, mine);  // (fg is wave-uniform: the in-wave exchange of LastInWave runs in whole waves)
        else if (halo) { if (in_regs) stage_last_regs<N, P, +1>(x, tl, tw, s); else stage_store<N, P, +1>(x, tl, mine, tw, s); }
        if (!in_regs) row_sync();
    }
    MW_STAMP(1, 2);
    // the final pass reads a row buffer by its own row group alone, too: that group may write it again without a barrier
    cf* set_s = set0 + 2 * G::SETSTRIDE;
    if (row || halo) {
        p2_last_load<N, P>(x, tl % T, mine + (row ? tl / T : 0) * G::BUFSTRIDE);
        final_stage<N, P, +1>(x, tl % T, tw.TF);
    }
    MW_STAMP(1, 3);
    if (fg == 1) p2_frame_hds<N, P, R2>(ab, tl, x, set_d);
    else if (fg == 2) p2_frame_normals<N, P, R2>(A, ab, step, tl, x, set_s);
    else if (halo) p2_hs_halo_publish<N, P, R2>(ab, tl, x, hbuf);
    MW_STAMP(1, 4);
    __syncthreads();
    MW_STAMP(1, 5);
    if (fg == 0) p2_frame_vertices<N, P, R2>(A, ab, step, tl, x, set_d);
    else if (fg == 1) p2_frame_white<N, P, R2>(A, ab, step, tl, set_d, hbuf, set_s);
    MW_STAMP(1, 6);
    MW_STAMP_RT(1, 31);
}

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
struct mw_ocean {
    mw_params p;
    int N = 0;          // synthesis grid size
    int sem = 0;
    int device = 0;
    bool use_fft = false;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    int p1_tgroup = 8;  // time-steps of one pass-1 column job grouped on one XCD (p1_block_map): -25 % pass-1 time;
                        // env MW_P1_TGROUP overrides (0 = plain 2-D grid)
    float timer = 0.f;
    // FFTMesh state
    cf *h0 = nullptr, *h0c = nullptr;
    f4 *PQt = nullptr, *dPQ_i0 = nullptr, *dPQ_j0 = nullptr;
    float* Om = nullptr;
    cf *TW = nullptr, *TW2 = nullptr, *Wpre = nullptr;  // twiddle tables of pass 1 / pass 2
    int* p1_jobs = nullptr;  // single-step plan: pass-1 job list (p1_frame_jobs)
    int p1_njobs = 0;
    cf *E = nullptr, *Cj0 = nullptr;
    int e_cap = 0;  // steps the exchange buffer holds
    float *s_vert = nullptr, *s_norm = nullptr, *s_white = nullptr;  // 1-step scratch for the host API
    void* scratch = nullptr;  // grow-only device staging of the host-pointer entry points (rest mesh, RGBA targets, ...):
    size_t scratch_cap = 0;   // allocated once at the largest size asked for, not per call
    DirectState direct;
    // OceanRenderer state
    OrState orr;
};

static bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

template <typename T>
static mw_status dmalloc(T** p, size_t count) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return MW_OK;
}

// device staging for the host-pointer entry points: one grow-only buffer per handle
static mw_status scratch_reserve(mw_ocean* o, size_t bytes, void** out) {
    if (o->scratch_cap < bytes) {
        if (o->scratch) {
            HIP_TRY(hipStreamSynchronize(o->stream));
            HIP_TRY(hipFree(o->scratch));
            o->scratch = nullptr;
            o->scratch_cap = 0;
        }
        if (hipMalloc(&o->scratch, bytes) != hipSuccess) return fail(MW_ENOMEM, "hipMalloc of the host-API staging buffer failed");
        o->scratch_cap = bytes;
    }
    *out = o->scratch;
    return MW_OK;
}
static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// host-side geometry mirror of FftGeom<N,P> / Plan<N>
static int plan_points(int N, int pass) {
    switch (N) {
#define MW_PLAN_CASE(NN) case NN: return pass == 1 ? Plan<NN>::P1 : Plan<NN>::P2;
        MW_PLAN_CASE(64) MW_PLAN_CASE(128) MW_PLAN_CASE(256) MW_PLAN_CASE(512) MW_PLAN_CASE(1024) MW_PLAN_CASE(2048) MW_PLAN_CASE(4096)
#undef MW_PLAN_CASE
        default: return 16;
    }
}

// concatenated twiddle table [TS1 | TS2 | TS3 | TF] in the layout of TwGeom<N,P>
namespace mw {
int plan_points_host(int N) { return N >= 2048 ? 16 : MW_PT; }
std::vector<cf> build_twiddle_table(int N, int P, int sgn) { return build_twiddle_table_host(N, P, sgn); }
}  // namespace mw

static mw_status upload_twiddles(mw_ocean* o) {
    const int N = o->N;
    std::vector<cf> tab = build_twiddle_table(N, plan_points(N, 1), +1), tab2 = build_twiddle_table(N, plan_points(N, 2), +1);
    std::vector<cf> Wpre(2 * N);
    for (int m = 0; m < 2 * N; m++) {
        double a = M_PI * (double)m / (double)N;  // (-1)^m e^{i pi m/N}
        double sg = (m & 1) ? -1.0 : 1.0;
        Wpre[m] = mk((float)(sg * cos(a)), (float)(sg * sin(a)));
    }
    mw_status st;
    if ((st = dmalloc(&o->TW, tab.size())) != MW_OK) return st;
    if ((st = dmalloc(&o->TW2, tab2.size())) != MW_OK) return st;
    if ((st = dmalloc(&o->Wpre, 2 * N)) != MW_OK) return st;
    HIP_TRY(hipMemcpy(o->TW, tab.data(), sizeof(cf) * tab.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(o->TW2, tab2.data(), sizeof(cf) * tab2.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(o->Wpre, Wpre.data(), sizeof(cf) * 2 * N, hipMemcpyHostToDevice));
    if (mw_frame_plan_n(N) && MW_LATENCY_PLAN) {
        const std::vector<int> jobs = p1_frame_jobs(N, N >= MW_CW2_MIN_N ? 2 : 4);
        if ((st = dmalloc(&o->p1_jobs, jobs.size())) != MW_OK) return st;
        HIP_TRY(hipMemcpy(o->p1_jobs, jobs.data(), sizeof(int) * jobs.size(), hipMemcpyHostToDevice));
        o->p1_njobs = (int)jobs.size();
    }
    return MW_OK;
}

static OceanConsts consts_of(const mw_ocean* o) {
    OceanConsts c;
    c.N = o->N;
    c.length = o->p.length;
    c.gravity = o->p.gravity;
    c.unit_width = o->p.unit_width;
    c.choppiness = o->p.choppiness;
    return c;
}

// the frame-at-a-time plan (single-step enqueues at 1024^2): compiled in by MW_LATENCY_PLAN, switched at run time by the
// switch of the same name (mw_switches.h; 0 = the batched plan for every enqueue; A/B without a rebuild)
static bool latency_plan_on() { return MW_LATENCY_PLAN && sw(SW_LATENCY_PLAN) != 0; }

// MW_FRAME_KERNEL=0: pass 2 of a single step by the sequential-halo kernel with one wave per row (round 3's frame plan) instead of
// k_pass2_frame; MW_P1_FRAME_XCD=0: its pass 1 on the plain (column jobs, 3) grid.  Run-time A/B switches, same bits either way.
static bool frame_kernel_on() { return sw(SW_FRAME_KERNEL) != 0; }
static bool p1_frame_xcd_on() { return sw(SW_P1_FRAME_XCD) != 0; }

// ---- kernel dispatch over N ----------------------------------------------------------------------
template <int N>
static hipError_t launch_pass1_n(const P1Args& A, const StepTimes& tm, int nsteps, hipStream_t st) {
    constexpr int P = Plan<N>::P1, VT = Plan<N>::VT1;
    static AttrOnce attr;  // per device: the attribute belongs to the function on the current device
    {
        hipError_t e = attr.set(reinterpret_cast<const void*>(&k_pass1<N, P, VT>), P1Geom<N, P>::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    constexpr int NT = P1Geom<N, P>::NTHREADS / VT, LB = P1Geom<N, P>::LDS_BYTES, GX = P1Geom<N, P>::GRID_X;
    if constexpr (mw_frame_plan_n(N) && MW_LATENCY_PLAN) {
        if (A.field_split) {
            static AttrOnce attrf;
            hipError_t e = attrf.set(reinterpret_cast<const void*>(&k_pass1<N, P, VT, true>), LB);
            if (e != hipSuccess) return e;
            if (A.field_split == 2) k_pass1<N, P, VT, true><<<dim3(A.njobs), dim3(NT), LB, st>>>(A, tm);
            else k_pass1<N, P, VT, true><<<dim3(GX, 3), dim3(NT), LB, st>>>(A, tm);
            return hipGetLastError();
        }
    }
    if (A.tgroup > 0)
        k_pass1<N, P, VT><<<dim3(p1_grid_blocks(GX, nsteps, A.tgroup)), dim3(NT), LB, st>>>(A, tm);
    else
        k_pass1<N, P, VT><<<dim3(GX, nsteps), dim3(NT), LB, st>>>(A, tm);
    return hipGetLastError();
}
template <int N, bool DUMP>
static hipError_t launch_pass2_n(const P2Args& A, int nsteps, hipStream_t st) {
    constexpr int P = Plan<N>::P2, R2 = Plan<N>::R2;
    constexpr bool HS = Plan<N>::HS;
    constexpr int VT = HS ? Plan<N>::VT : 1, PF = HS ? Plan<N>::PF : 0;
    constexpr int NT = P2Geom<N, P, R2, HS>::NTHREADS / VT, LB = P2Geom<N, P, R2, HS>::LDS_BYTES;
    static AttrOnce attr;
    {
        const void* fn;
        if constexpr (HS) fn = reinterpret_cast<const void*>(&k_pass2_hs<N, P, R2, VT, DUMP, PF>);
        else fn = reinterpret_cast<const void*>(&k_pass2<N, P, R2, DUMP>);
        hipError_t e = attr.set(fn, LB);
        if (e != hipSuccess) return e;
    }
    // Frame-at-a-time plan (FFTMesh.Update, S/FFTMesh.cs:60-73: ONE step per call; 512^2 and 1024^2): a step cannot fill the device,
    // its latency is that of one workgroup.  k_pass2_frame transforms the three fields of a row block side by side; MW_FRAME_KERNEL=0
    // selects round 3's form at 1024^2 -- the sequential-halo kernel with one virtual thread per lane (one wave per row, every load up
    // front) -- and the batched kernel at 512^2.  The arithmetic of a row does not depend on which kernel runs it: same bits.
    if constexpr (mw_frame_plan_n(N) && !DUMP && MW_LATENCY_PLAN) {
        if (nsteps == 1 && latency_plan_on()) {
            constexpr int RF = mw_frame_r2(N);
            if constexpr (P2FrameGeom<N, P, RF>::OK) {
                if (frame_kernel_on()) {
                    static AttrOnce attrf;
                    constexpr int LBF = P2FrameGeom<N, P, RF>::LDS_BYTES;
                    hipError_t e = attrf.set(reinterpret_cast<const void*>(&k_pass2_frame<N, P, RF>), LBF);
                    if (e != hipSuccess) return e;
                    k_pass2_frame<N, P, RF><<<dim3(N / RF, 1), dim3(P2FrameGeom<N, P, RF>::NTHREADS), LBF, st>>>(A);
                    return hipGetLastError();
                }
            }
            if constexpr (HS && VT == 2) {
                static AttrOnce attr1;
                constexpr int PFL = MW_LATENCY_PF;  // 2: every load of the workgroup requested up front
                hipError_t e = attr1.set(reinterpret_cast<const void*>(&k_pass2_hs<N, P, R2, 1, false, PFL>), LB);
                if (e != hipSuccess) return e;
                k_pass2_hs<N, P, R2, 1, false, PFL><<<dim3(N / R2, 1), dim3(P2Geom<N, P, R2, true>::NTHREADS), LB, st>>>(A);
                return hipGetLastError();
            }
        }
    }
    if constexpr (HS)
        k_pass2_hs<N, P, R2, VT, DUMP, PF><<<dim3(N / R2, nsteps), dim3(NT), LB, st>>>(A);
    else
        k_pass2<N, P, R2, DUMP><<<dim3(N / R2, nsteps), dim3(NT), LB, st>>>(A);
    return hipGetLastError();
}

#define MW_DISPATCH_N(N_, CALL)                  \
    switch (N_) {                                \
        case 64: { constexpr int NN = 64; CALL; } break;     \
        case 128: { constexpr int NN = 128; CALL; } break;   \
        case 256: { constexpr int NN = 256; CALL; } break;   \
        case 512: { constexpr int NN = 512; CALL; } break;   \
        case 1024: { constexpr int NN = 1024; CALL; } break; \
        case 2048: { constexpr int NN = 2048; CALL; } break; \
        case 4096: { constexpr int NN = 4096; CALL; } break; \
        default: return fail(MW_EINVAL, "unsupported FFT size"); \
    }

// time-steps of one pass-1 column job issued back to back on one XCD (p1_block_map): the largest divisor of nsteps up to
// the handle's p1_tgroup (8), any divisor -- a 20-step enqueue groups by 5 --, 0 = plain 2-D grid
static int p1_time_group(const mw_ocean* o, int nsteps) {
    for (int g = o->p1_tgroup; g > 1; g--)
        if (nsteps % g == 0) return g;
    return 0;
}
static mw_status launch_pass1(mw_ocean* o, const StepTimes& tm, int nsteps, hipStream_t st) {
    P1Args A;
    A.PQt = o->PQt; A.dPQ_i0 = o->dPQ_i0; A.dPQ_j0 = o->dPQ_j0; A.Om = o->Om; A.TW = o->TW;
    A.E = o->E; A.Cj0 = o->Cj0;
    A.c = consts_of(o);
    A.nsteps = nsteps;
    A.tgroup = p1_time_group(o, nsteps);
    A.field_split = (latency_plan_on() && nsteps == 1 && mw_frame_plan_n(o->N)) ? ((p1_frame_xcd_on() && o->p1_jobs) ? 2 : 1) : 0;
    if (A.field_split == 2) { A.jobs = o->p1_jobs; A.njobs = o->p1_njobs; }
    hipError_t e = hipSuccess;
    MW_DISPATCH_N(o->N, e = launch_pass1_n<NN>(A, tm, nsteps, st));
    if (e != hipSuccess) return fail(MW_EDEVICE, std::string("pass1 launch: ") + hipGetErrorString(e));
    return MW_OK;
}
static mw_status launch_pass2(mw_ocean* o, int nsteps, float* dv, float* dn, float* dw, int white_stride, cf* hds_dump = nullptr) {
    P2Args A;
    A.E = o->E; A.Cj0 = o->Cj0; A.TW = o->TW2; A.vertices = dv; A.normals = dn; A.white = dw; A.white_stride = white_stride;
    A.hds_dump = hds_dump;
    A.c = consts_of(o);
    hipError_t e = hipSuccess;
    if (hds_dump) { MW_DISPATCH_N(o->N, (e = launch_pass2_n<NN, true>(A, nsteps, o->stream))); }
    else { MW_DISPATCH_N(o->N, (e = launch_pass2_n<NN, false>(A, nsteps, o->stream))); }
    if (e != hipSuccess) return fail(MW_EDEVICE, std::string("pass2 launch: ") + hipGetErrorString(e));
    return MW_OK;
}

static mw_status ensure_exchange(mw_ocean* o, int nsteps) {
    if (o->e_cap >= nsteps) return MW_OK;
    if (o->E) {
        HIP_TRY(hipStreamSynchronize(o->stream));
        HIP_TRY(hipFree(o->E));
        HIP_TRY(hipFree(o->Cj0));
        o->E = o->Cj0 = nullptr;
        o->e_cap = 0;
    }
    mw_status s = dmalloc(&o->E, (size_t)nsteps * 3 * o->N * o->N);
    if (s != MW_OK) return s;
    if ((s = dmalloc(&o->Cj0, (size_t)nsteps * 3 * o->N)) != MW_OK) return s;
    o->e_cap = nsteps;
    return MW_OK;
}

static mw_status run_prep(mw_ocean* o) {
    const int N = o->N;
    if (o->use_fft) {
        hipLaunchKernelGGL(k_prep, dim3((N * N + 255) / 256), dim3(256), 0, o->stream, N, o->p.length, o->p.gravity,
                           o->h0, o->h0c, o->Wpre, o->PQt, o->dPQ_i0, o->dPQ_j0, o->Om);
        HIP_TRY(hipGetLastError());
    }
    return MW_OK;
}

// FETCH_SIZE calibration streams (MI355X_MICROARCH.md "HBM": calibrate the counter on a known byte count in your own
// access width): every lane reads `width` bytes, lanes contiguous, `bytes` in total; one float per block is written.
template <typename V>
__global__ void k_dbg_stream(const V* __restrict__ src, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        V v = src[i];
        acc += reinterpret_cast<const float*>(&v)[0];
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int32_t mw_abi_version(void) { return MW_ABI_VERSION; }

// Identity of this build, fixed AT BUILD TIME: -DMW_BUILD_HASH = SHA-256 (first 16 hex digits) over the kernel sources, the two
// headers of the boundary and the compile flags, computed by the build recipe (mistral_water/_native.py::source_hash, the one
// place that knows them; tools/build_variant.sh calls the same function) + the tag the build gave it (-DMW_BUILD_TAG, "default"
// for the in-tree .so).  It names the code that is loaded, whatever happens to the file on disk afterwards.
#ifndef MW_BUILD_TAG
#define MW_BUILD_TAG "default"
#endif
// -DMW_LAB: a build for measurements (a knob overridden on the command line, cycle stamps, the plan switches taken from the environment).
// Its id says so, and bench.py and the tests refuse it unless told otherwise: a number or a green test always names a product build.
#ifdef MW_LAB
#define MW_BUILD_KIND "lab:"
#else
#define MW_BUILD_KIND ""
#endif
#ifndef MW_BUILD_HASH
#warning "MW_BUILD_HASH is not defined: build through mistral_water/_native.py::build_native or tools/build_variant.sh (mw_build_id() will say unhashed-build)"
#define MW_BUILD_HASH "unhashed-build"
#endif
const char* mw_build_id(void) { return MW_BUILD_HASH " " MW_BUILD_KIND MW_BUILD_TAG; }
const char* mw_last_error(void) { return g_err.c_str(); }

int32_t mw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void mw_params_default(mw_params* p, int32_t semantics) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->semantics = semantics;
    p->gravity = 9.81f;  // S/FFTMesh.cs:52
    p->seed = 1;
    if (semantics == MW_SEM_OCEANRENDERER) {  // S/OceanRenderer.cs:10-19
        p->mult = 2.f; p->unit_width = 1.f; p->resolution = 256; p->length = 256.f; p->choppiness = 1.5f;
        p->amplitude = 1.f; p->wind_x = 0.f; p->wind_y = 0.f; p->t_division = 1.f;
    } else {  // S/FFTMesh.cs:9-23
        p->choppiness = 1.f; p->t_division = 1.f; p->resolution = 50; p->unit_width = 1.f; p->length = 1.f;
        p->wind_x = 1.f; p->wind_y = 1.f; p->amplitude = 1.f; p->mult = 1.f;
    }
}

void mw_ocean_destroy(mw_ocean* o) {
    if (!o) return;
    hipSetDevice(o->device);
    if (hipStreamSynchronize(o->stream) != hipSuccess) (void)hipGetLastError();  // a dead caller stream has nothing pending
    hipFree(o->h0); hipFree(o->h0c); hipFree(o->PQt); hipFree(o->Om); hipFree(o->dPQ_i0); hipFree(o->dPQ_j0);
    hipFree(o->TW); hipFree(o->TW2); hipFree(o->Wpre); hipFree(o->p1_jobs); hipFree(o->E); hipFree(o->Cj0); hipFree(o->s_vert); hipFree(o->s_norm); hipFree(o->s_white); hipFree(o->scratch);
    direct_free(o->direct);
    or_free(o->orr);
    if (o->own_stream) hipStreamDestroy(o->own_stream);
    delete o;
}

static mw_status ocean_create_impl(const mw_params* params, int tiles, mw_ocean** out) {
    if (!params || !out) return fail(MW_EINVAL, "mw_ocean_create: NULL argument");
    *out = nullptr;
    if (tiles < 1 || tiles > 64) return fail(MW_EINVAL, "mw_ocean_create_batch: ntiles must be in [1,64]");
    if (tiles > 1 && params->semantics != MW_SEM_OCEANRENDERER)
        return fail(MW_EINVAL, "mw_ocean_create_batch: OceanRenderer semantics only (FFTMesh batches in time, mw_ocean_evaluate_device)");
    if (params->semantics != MW_SEM_FFTMESH && params->semantics != MW_SEM_OCEANRENDERER)
        return fail(MW_EINVAL, "mw_ocean_create: unknown semantics");
    if (params->resolution < 2) return fail(MW_EINVAL, "mw_ocean_create: resolution must be >= 2");
    if (!(params->length > 0.f) || !(params->gravity > 0.f))
        return fail(MW_EINVAL, "mw_ocean_create: length and gravity must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(MW_EDEVICE, "mw_ocean_create: no HIP device visible (this library has no CPU fallback)");
    if (params->device < 0 || params->device >= ndev) return fail(MW_EINVAL, "mw_ocean_create: bad device ordinal");
    HIP_TRY(hipSetDevice(params->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, params->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MW_EDEVICE, std::string("mw_ocean_create: device is ") + prop.gcnArchName +
                                    ", this library carries gfx950 (MI355X) code only");

    mw_ocean* o = new (std::nothrow) mw_ocean();
    if (!o) return fail(MW_ENOMEM, "mw_ocean_create: out of host memory");
    o->p = *params;
    o->sem = params->semantics;
    o->device = params->device;
    mw_status s = MW_OK;
    hipError_t he = hipStreamCreateWithFlags(&o->own_stream, hipStreamNonBlocking);
    if (he != hipSuccess) { delete o; return fail(MW_EDEVICE, "hipStreamCreate failed"); }
    o->stream = o->own_stream;
    if (sw(SW_P1_TGROUP) >= 0) o->p1_tgroup = sw(SW_P1_TGROUP);

    if (o->sem == MW_SEM_FFTMESH) {
        const int N = params->resolution;
        if (N > 4096) { mw_ocean_destroy(o); return fail(MW_EINVAL, "FFTMesh: resolution > 4096 unsupported"); }
        o->N = N;
        // FFT path precondition: power of two and unit_width == length / N exactly.
        o->use_fft = is_pow2(N) && N >= 64 && (params->unit_width * (float)N == params->length);
        const size_t NN = (size_t)N * N;
        if ((s = dmalloc(&o->h0, NN)) != MW_OK || (s = dmalloc(&o->h0c, NN)) != MW_OK ||
            (s = dmalloc(&o->s_vert, NN * 3)) != MW_OK || (s = dmalloc(&o->s_norm, NN * 3)) != MW_OK ||
            (s = dmalloc(&o->s_white, NN * 4)) != MW_OK) { mw_ocean_destroy(o); return s; }
        if (o->use_fft) {
            if ((s = dmalloc(&o->PQt, NN)) != MW_OK || (s = dmalloc(&o->Om, NN)) != MW_OK ||
                (s = dmalloc(&o->dPQ_i0, (size_t)N)) != MW_OK ||
                (s = dmalloc(&o->dPQ_j0, (size_t)N)) != MW_OK || (s = upload_twiddles(o)) != MW_OK) {
                mw_ocean_destroy(o); return s;
            }
        } else {
            if (direct_alloc(o->direct, N, o->stream) != 0) { mw_ocean_destroy(o); return fail(MW_ENOMEM, "direct path alloc failed"); }
            if (direct_prepare_tables(o->direct, N, params->unit_width, params->length, params->gravity, o->stream) != hipSuccess) {
                mw_ocean_destroy(o);
                return fail(MW_EDEVICE, "direct path: chirp tables could not be uploaded");
            }
        }
        hipLaunchKernelGGL(k_spectrum, dim3((unsigned)((NN + 255) / 256)), dim3(256), 0, o->stream, N, params->length,
                           params->wind_x, params->wind_y, params->amplitude, params->gravity, params->seed, o->h0, o->h0c);
        if (hipGetLastError() != hipSuccess) { mw_ocean_destroy(o); return fail(MW_EDEVICE, "k_spectrum launch failed"); }
        if ((s = run_prep(o)) != MW_OK) { mw_ocean_destroy(o); return s; }
    } else {
        const int M = params->resolution * 8;  // S/OceanRenderer.cs:136
        if (!is_pow2(M)) { mw_ocean_destroy(o); return fail(MW_ENOTPOW2, "OceanRenderer: 8*resolution must be a power of two"); }
        if (M < 64 || M > 4096) { mw_ocean_destroy(o); return fail(MW_EINVAL, "OceanRenderer: texture size must be in [64,4096]"); }
        o->N = M;
        if ((s = or_create(o->orr, *params, M, o->stream, tiles)) != MW_OK) {
            std::string m = or_last_error();
            mw_ocean_destroy(o);
            return fail(s, m);
        }
    }
    he = hipStreamSynchronize(o->stream);
    if (he != hipSuccess) { mw_ocean_destroy(o); return fail(MW_EDEVICE, std::string("create sync: ") + hipGetErrorString(he)); }
    *out = o;
    return MW_OK;
}

mw_status mw_ocean_create(const mw_params* params, mw_ocean** out) { return ocean_create_impl(params, 1, out); }
mw_status mw_ocean_create_batch(const mw_params* params, int32_t ntiles, mw_ocean** out) { return ocean_create_impl(params, ntiles, out); }
int32_t mw_ocean_batch_size(const mw_ocean* o) { return o ? (o->sem == MW_SEM_OCEANRENDERER ? o->orr.tiles : 1) : 0; }

// Drain the stream the handle is leaving.  A caller-owned stream may have been destroyed since (a garbage-collected
// torch.cuda.Stream): hipErrorInvalidHandle / ContextIsDestroyed then mean "nothing pending" -- the handle must still be able
// to leave the dead stream (and mw_ocean_destroy to finish).
static mw_status drain_stream(mw_ocean* o) {
    const hipError_t e = hipStreamSynchronize(o->stream);
    if (e == hipSuccess) return MW_OK;
    if (o->stream != o->own_stream &&
        (e == hipErrorInvalidHandle || e == hipErrorContextIsDestroyed || e == hipErrorInvalidResourceHandle)) {
        (void)hipGetLastError();  // clear the sticky error: it described the caller's stream, not this library's work
        return MW_OK;
    }
    return fail(MW_EDEVICE, std::string("hipStreamSynchronize(previous stream): ") + hipGetErrorString(e));
}
mw_status mw_ocean_set_stream(mw_ocean* o, void* hip_stream) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    HIP_TRY(hipSetDevice(o->device));
    mw_status s = drain_stream(o);
    if (s != MW_OK) return s;
    o->stream = reinterpret_cast<hipStream_t>(hip_stream);  // NULL = HIP's legacy default stream, like the pond entry points
    return MW_OK;
}
mw_status mw_ocean_use_own_stream(mw_ocean* o) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    HIP_TRY(hipSetDevice(o->device));
    mw_status s = drain_stream(o);
    if (s != MW_OK) return s;
    o->stream = o->own_stream;
    return MW_OK;
}
void* mw_ocean_get_stream(mw_ocean* o) { return o ? reinterpret_cast<void*>(o->stream) : nullptr; }
mw_status mw_ocean_synchronize(mw_ocean* o) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    HIP_TRY(hipSetDevice(o->device));
    HIP_TRY(hipStreamSynchronize(o->stream));
    return MW_OK;
}
mw_status mw_ocean_set_choppiness(mw_ocean* o, float c) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    o->p.choppiness = c;
    return MW_OK;
}

int32_t mw_ocean_grid_size(const mw_ocean* o) { return o ? o->N : 0; }
int64_t mw_ocean_index_count(const mw_ocean* o) {
    if (!o) return 0;
    int64_t n = o->p.resolution;  // the MESH is resolution^2 in both modes (S/OceanRenderer.cs:132-135)
    return (n - 1) * (n - 1) * 6;
}
int32_t mw_ocean_max_batch(const mw_ocean* o) { return (o && o->sem == MW_SEM_FFTMESH && o->use_fft) ? MW_MAX_BATCH : 1; }
float mw_ocean_timer(const mw_ocean* o) { return o ? o->timer : 0.f; }
mw_status mw_ocean_reset_timer(mw_ocean* o) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    o->timer = 0.f;
    return MW_OK;
}

mw_status mw_ocean_set_spectrum(mw_ocean* o, const float* h0_xy, const float* h0conj_xy) {
    if (!o || !h0_xy || !h0conj_xy) return fail(MW_EINVAL, "mw_ocean_set_spectrum: NULL argument");
    HIP_TRY(hipSetDevice(o->device));
    const int tiles = (o->sem == MW_SEM_OCEANRENDERER) ? o->orr.tiles : 1;
    const size_t bytes = sizeof(cf) * (size_t)o->N * o->N * tiles;
    if (o->sem == MW_SEM_OCEANRENDERER) {  // initialTexture.rg / .ba, texel (px,py) at py*M + px, tile-major
        void* buf = nullptr;
        mw_status s = scratch_reserve(o, 2 * align256(bytes), &buf);
        if (s != MW_OK) return s;
        cf *a = static_cast<cf*>(buf), *b = reinterpret_cast<cf*>(static_cast<char*>(buf) + align256(bytes));
        HIP_TRY(hipMemcpyAsync(a, h0_xy, bytes, hipMemcpyHostToDevice, o->stream));
        HIP_TRY(hipMemcpyAsync(b, h0conj_xy, bytes, hipMemcpyHostToDevice, o->stream));
        k_or_set_init<<<dim3((unsigned)(((size_t)o->N * o->N + 255) / 256), tiles), dim3(256), 0, o->stream>>>(o->N, a, b, o->orr.initT,
                                                                                                          o->orr.phaseT);
        k_or_prep<<<dim3((unsigned)(((size_t)o->N * o->N + 255) / 256), tiles), dim3(256), 0, o->stream>>>(o->N, o->orr.initT, o->orr.PQT);
        HIP_TRY(hipGetLastError());
        o->orr.phase_sym = true;  // a fresh spectrum restarts the phase at 0
        HIP_TRY(hipStreamSynchronize(o->stream));
        return MW_OK;
    }
    HIP_TRY(hipMemcpyAsync(o->h0, h0_xy, bytes, hipMemcpyHostToDevice, o->stream));
    HIP_TRY(hipMemcpyAsync(o->h0c, h0conj_xy, bytes, hipMemcpyHostToDevice, o->stream));
    mw_status s = run_prep(o);
    if (s != MW_OK) return s;
    HIP_TRY(hipStreamSynchronize(o->stream));
    return MW_OK;
}
mw_status mw_ocean_get_spectrum(mw_ocean* o, float* h0_xy, float* h0conj_xy) {
    if (!o || !h0_xy || !h0conj_xy) return fail(MW_EINVAL, "mw_ocean_get_spectrum: NULL argument");
    HIP_TRY(hipSetDevice(o->device));
    const int tiles = (o->sem == MW_SEM_OCEANRENDERER) ? o->orr.tiles : 1;
    const size_t bytes = sizeof(cf) * (size_t)o->N * o->N * tiles;
    if (o->sem == MW_SEM_OCEANRENDERER) {
        void* buf = nullptr;
        mw_status s = scratch_reserve(o, 2 * align256(bytes), &buf);
        if (s != MW_OK) return s;
        cf *a = static_cast<cf*>(buf), *b = reinterpret_cast<cf*>(static_cast<char*>(buf) + align256(bytes));
        k_or_get_init<<<dim3((unsigned)(((size_t)o->N * o->N + 255) / 256), tiles), dim3(256), 0, o->stream>>>(o->N, o->orr.initT, a, b);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h0_xy, a, bytes, hipMemcpyDeviceToHost, o->stream));
        HIP_TRY(hipMemcpyAsync(h0conj_xy, b, bytes, hipMemcpyDeviceToHost, o->stream));
        HIP_TRY(hipStreamSynchronize(o->stream));
        return MW_OK;
    }
    HIP_TRY(hipMemcpyAsync(h0_xy, o->h0, bytes, hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipMemcpyAsync(h0conj_xy, o->h0c, bytes, hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipStreamSynchronize(o->stream));
    return MW_OK;
}

mw_status mw_ocean_reinit_spectrum(mw_ocean* o, float length, float wind_x, float wind_y, float amplitude, uint64_t seed) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    if (!(length > 0.f)) return fail(MW_EINVAL, "mw_ocean_reinit_spectrum: length must be positive");
    HIP_TRY(hipSetDevice(o->device));
    if (o->sem == MW_SEM_OCEANRENDERER) {  // S/OceanRenderer.cs:98-109: RenderInitial() again, phase textures untouched
        mw_status s = or_reinit(o->orr, length, wind_x, wind_y, amplitude, seed, o->stream);
        if (s != MW_OK) return fail(s, or_last_error());
    } else {
        const int N = o->N;
        const bool fft = is_pow2(N) && N >= 64 && (o->p.unit_width * (float)N == length);
        if (fft != o->use_fft)
            return fail(MW_ESTATE, "mw_ocean_reinit_spectrum: the new length moves the grid between the FFT and the direct-sum path; "
                                   "create a new handle");
        const size_t NN = (size_t)N * N;
        // Transactional: the new spectrum is generated into buffers of its own and the derived tables (PQt, omega) are rebuilt
        // from there; the handle adopts the new buffers only once every step has succeeded (and frees the old ones), otherwise
        // it keeps the old spectrum and rebuilds the tables from it with the old length.  No staging in `scratch`: another host
        // entry point cannot clobber the new spectrum half-way.
        cf *n0 = nullptr, *n0c = nullptr;
        mw_status s = dmalloc(&n0, NN);
        if (s == MW_OK) s = dmalloc(&n0c, NN);
        if (s != MW_OK) { hipFree(n0); hipFree(n0c); return s; }
        hipLaunchKernelGGL(k_spectrum, dim3((unsigned)((NN + 255) / 256)), dim3(256), 0, o->stream, N, length, wind_x, wind_y,
                           amplitude, o->p.gravity, seed, n0, n0c);
        hipError_t e = hipGetLastError();
        const float old_length = o->p.length;
        cf *old0 = o->h0, *old0c = o->h0c;
        o->p.length = length;  // run_prep reads it (omega table, S/FFTMesh.cs:141-147)
        o->h0 = n0; o->h0c = n0c;
        if (e == hipSuccess) s = run_prep(o);
        if (s == MW_OK && e == hipSuccess) e = hipStreamSynchronize(o->stream);
        if (s == MW_OK && e == hipSuccess && !o->use_fft && length != old_length)  // chirp tables of the new length, here and not inside the next enqueue
            e = direct_prepare_tables(o->direct, N, o->p.unit_width, length, o->p.gravity, o->stream);
        if (s != MW_OK || e != hipSuccess) {
            o->h0 = old0; o->h0c = old0c;
            o->p.length = old_length;
            (void)run_prep(o);  // tables back to (old spectrum, old length); if the device is gone, the handle is too (MW_EDEVICE)
            (void)hipStreamSynchronize(o->stream);
            hipFree(n0); hipFree(n0c);
            return s != MW_OK ? s : fail(MW_EDEVICE, std::string("mw_ocean_reinit_spectrum: ") + hipGetErrorString(e));
        }
        hipFree(old0); hipFree(old0c);
    }
    o->p.length = length; o->p.wind_x = wind_x; o->p.wind_y = wind_y; o->p.amplitude = amplitude; o->p.seed = seed;
    HIP_TRY(hipStreamSynchronize(o->stream));
    return MW_OK;
}

// OceanRenderer: the length the normal pass uses (normalMat._Length, set once in SetParams, S/OceanRenderer.cs:163, and
// never again -- a later length change through mw_ocean_reinit_spectrum leaves it behind).  Part of the checkpoint:
// restoring (initialTexture, phase) into a fresh handle created with the CURRENT length must also restore this.
float mw_ocean_normal_length(const mw_ocean* o) {
    if (!o) return 0.f;
    return o->sem == MW_SEM_OCEANRENDERER ? o->orr.c.normal_length : o->p.length;
}
mw_status mw_ocean_set_normal_length(mw_ocean* o, float normal_length) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, "mw_ocean_set_normal_length: OceanRenderer semantics only");
    if (!(normal_length > 0.f)) return fail(MW_EINVAL, "mw_ocean_set_normal_length: must be positive");
    o->orr.c.normal_length = normal_length;
    return MW_OK;
}

static mw_status phase_copy(mw_ocean* o, float* host_out, const float* host_in, const char* who) {
    if (!o || (!host_out && !host_in)) return fail(MW_EINVAL, std::string(who) + ": NULL argument");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, std::string(who) + ": OceanRenderer semantics only (FFTMesh state is the timer)");
    HIP_TRY(hipSetDevice(o->device));
    const size_t MM = (size_t)o->N * o->N, bytes = MM * sizeof(float) * o->orr.tiles;
    void* buf = nullptr;
    mw_status s = scratch_reserve(o, bytes, &buf);
    if (s != MW_OK) return s;
    float* tmp = static_cast<float*>(buf);
    const dim3 grid((unsigned)((MM + 255) / 256), o->orr.tiles), block(256);
    if (host_out) {  // device [px][py] -> host texel order py*M + px
        k_or_phase_transpose<<<grid, block, 0, o->stream>>>(o->N, o->orr.phaseT, tmp);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host_out, tmp, bytes, hipMemcpyDeviceToHost, o->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(tmp, host_in, bytes, hipMemcpyHostToDevice, o->stream));
        k_or_phase_transpose<<<grid, block, 0, o->stream>>>(o->N, tmp, o->orr.phaseT);
        HIP_TRY(hipGetLastError());
        // The two-transform plan needs phase[p] == phase[m(p)], m the index mirror (csrc/ocean_renderer_kernels.h): true of every phase the library
        // produced (omega is mirror-symmetric bit for bit), checked for what a caller hands in -- anything else runs the three-transform plan
        bool sym = true;
        const int M = o->N;
        for (int t = 0; t < o->orr.tiles && sym; t++) {
            const float* ph = host_in + (size_t)t * MM;
            for (int py = 0; py < M && sym; py++) {
                const float* row = ph + (size_t)py * M;
                const float* mrow = ph + (size_t)((M - py) % M) * M;
                for (int px = 0; px < M; px++)
                    if (row[px] != mrow[(M - px) % M]) { sym = false; break; }
            }
        }
        o->orr.phase_sym = sym;
    }
    HIP_TRY(hipStreamSynchronize(o->stream));
    return MW_OK;
}
mw_status mw_ocean_get_phase(mw_ocean* o, float* phase) { return phase_copy(o, phase, nullptr, "mw_ocean_get_phase"); }
mw_status mw_ocean_set_phase(mw_ocean* o, const float* phase) { return phase_copy(o, nullptr, phase, "mw_ocean_set_phase"); }
mw_status mw_ocean_set_timer(mw_ocean* o, float timer) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    o->timer = timer;
    return MW_OK;
}

mw_status mw_ocean_rest_mesh(mw_ocean* o, float* vertices_xyz, float* normals_xyz, float* uvs_xy, int32_t* indices) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    HIP_TRY(hipSetDevice(o->device));
    const int N = o->p.resolution;  // mesh resolution (not the 8x texture size in OceanRenderer mode)
    const size_t NN = (size_t)N * N, nidx = (size_t)(N - 1) * (N - 1) * 6;
    const size_t bv = align256(NN * 3 * sizeof(float)), bu = align256(NN * 2 * sizeof(float)), bi = align256(nidx * sizeof(int32_t));
    void* buf = nullptr;
    mw_status s = scratch_reserve(o, 2 * bv + bu + bi, &buf);
    if (s != MW_OK) return s;
    char* base = static_cast<char*>(buf);
    float* dv = vertices_xyz ? reinterpret_cast<float*>(base) : nullptr;
    float* dn = normals_xyz ? reinterpret_cast<float*>(base + bv) : nullptr;
    float* du = uvs_xy ? reinterpret_cast<float*>(base + 2 * bv) : nullptr;
    int32_t* di = indices ? reinterpret_cast<int32_t*>(base + 2 * bv + bu) : nullptr;
    hipLaunchKernelGGL(k_rest_mesh, dim3((unsigned)((NN + 255) / 256)), dim3(256), 0, o->stream, N, o->p.unit_width, dv,
                       dn, du, di);
    HIP_TRY(hipGetLastError());
    if (dv) HIP_TRY(hipMemcpyAsync(vertices_xyz, dv, NN * 3 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (dn) HIP_TRY(hipMemcpyAsync(normals_xyz, dn, NN * 3 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (du) HIP_TRY(hipMemcpyAsync(uvs_xy, du, NN * 2 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (di) HIP_TRY(hipMemcpyAsync(indices, di, nidx * sizeof(int32_t), hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipStreamSynchronize(o->stream));
    return MW_OK;
}

mw_status mw_ocean_evaluate_device(mw_ocean* o, const float* t, int32_t nsteps, void* d_vertices, void* d_normals,
                                   void* d_white, uint32_t flags) {
    if (!o || !t || !d_vertices || !d_normals || !d_white) return fail(MW_EINVAL, "mw_ocean_evaluate_device: NULL argument");
    if (o->sem != MW_SEM_FFTMESH) return fail(MW_ESTATE, "mw_ocean_evaluate_device: FFTMesh semantics only");
    if (nsteps < 1 || nsteps > mw_ocean_max_batch(o)) return fail(MW_EINVAL, "mw_ocean_evaluate_device: nsteps out of range");
    HIP_TRY(hipSetDevice(o->device));
    const int white_stride = (flags & MW_OUT_COLOR_RGBA) ? 4 : 1;
    if (!o->use_fft) {
        return direct_evaluate(o->direct, consts_of(o), o->h0, o->h0c, t[0], (float*)d_vertices, (float*)d_normals,
                               (float*)d_white, white_stride, o->stream) == hipSuccess
                   ? MW_OK
                   : fail(MW_EDEVICE, "direct-sum kernels failed to launch");
    }
    mw_status s = ensure_exchange(o, nsteps);
    if (s != MW_OK) return s;
    StepTimes tm;
    for (int k = 0; k < nsteps; k++) tm.t[k] = t[k];
    if ((s = launch_pass1(o, tm, nsteps, o->stream)) != MW_OK) return s;
    return launch_pass2(o, nsteps, (float*)d_vertices, (float*)d_normals, (float*)d_white, white_stride);
}

mw_status mw_ocean_evaluate(mw_ocean* o, float t, float* vertices_xyz, float* normals_xyz, float* colors_rgba) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    if (o->sem != MW_SEM_FFTMESH) return fail(MW_ESTATE, "mw_ocean_evaluate: FFTMesh semantics only");
    mw_status s = mw_ocean_evaluate_device(o, &t, 1, o->s_vert, o->s_norm, o->s_white, MW_OUT_COLOR_RGBA);
    if (s != MW_OK) return s;
    const size_t NN = (size_t)o->N * o->N;
    if (vertices_xyz) HIP_TRY(hipMemcpyAsync(vertices_xyz, o->s_vert, NN * 3 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (normals_xyz) HIP_TRY(hipMemcpyAsync(normals_xyz, o->s_norm, NN * 3 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (colors_rgba) HIP_TRY(hipMemcpyAsync(colors_rgba, o->s_white, NN * 4 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipStreamSynchronize(o->stream));
    return MW_OK;
}

mw_status mw_ocean_update(mw_ocean* o, float delta_time, float* vertices_xyz, float* normals_xyz, float* colors_rgba) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    o->timer += delta_time / o->p.t_division;  // S/FFTMesh.cs:70
    return mw_ocean_evaluate(o, o->timer, vertices_xyz, normals_xyz, colors_rgba);
}

mw_status mw_ocean_generate_texture_device(mw_ocean* o, float delta_time, void* d_height, void* d_disp_xz,
                                           void* d_normal_xyz, void* d_white) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, "mw_ocean_generate_texture: OceanRenderer semantics only");
    HIP_TRY(hipSetDevice(o->device));
    o->orr.choppiness = o->p.choppiness;
    mw_status s = or_generate(o->orr, delta_time, (float*)d_height, (float*)d_disp_xz, (float*)d_normal_xyz, (float*)d_white,
                              o->stream);
    if (s != MW_OK) return fail(s, or_last_error());
    return MW_OK;
}

mw_status mw_ocean_generate_texture(mw_ocean* o, float delta_time, float* height, float* disp_xz, float* normal_xyz,
                                    float* white) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, "mw_ocean_generate_texture: OceanRenderer semantics only");
    mw_status s = mw_ocean_generate_texture_device(o, delta_time, nullptr, nullptr, nullptr, nullptr);
    if (s != MW_OK) return s;
    const size_t MM = (size_t)o->N * o->N * o->orr.tiles;
    if (height) HIP_TRY(hipMemcpyAsync(height, o->orr.out_height, MM * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (disp_xz) HIP_TRY(hipMemcpyAsync(disp_xz, o->orr.out_disp, MM * 2 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (normal_xyz) HIP_TRY(hipMemcpyAsync(normal_xyz, o->orr.out_normal, MM * 3 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (white) HIP_TRY(hipMemcpyAsync(white, o->orr.out_white, MM * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipStreamSynchronize(o->stream));
    return MW_OK;
}

mw_status mw_ocean_generate_texture_steps_device(mw_ocean* o, const float* delta_time, int32_t nframes, void* d_height,
                                                 void* d_disp_xz, void* d_normal_xyz, void* d_white) {
    if (!o || !delta_time) return fail(MW_EINVAL, "mw_ocean_generate_texture_steps: NULL argument");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, "mw_ocean_generate_texture_steps: OceanRenderer semantics only");
    if (nframes < 1 || nframes > mw_ocean_max_frames(o))
        return fail(o->orr.tiles != 1 ? MW_ESTATE : MW_EINVAL, o->orr.tiles != 1 ? "mw_ocean_generate_texture_steps: a batched handle (mw_ocean_create_batch) advances one frame per call"
                                                                                 : "mw_ocean_generate_texture_steps: nframes out of range (mw_ocean_max_frames)");
    HIP_TRY(hipSetDevice(o->device));
    o->orr.choppiness = o->p.choppiness;
    // one frame: the lone-frame plan of mw_ocean_generate_texture_device (latency-bound launch forms); its textures are the handle's own
    mw_status s = nframes == 1 ? or_generate(o->orr, delta_time[0], (float*)d_height, (float*)d_disp_xz, (float*)d_normal_xyz, (float*)d_white, o->stream)
                               : or_generate_steps(o->orr, delta_time, nframes, (float*)d_height, (float*)d_disp_xz, (float*)d_normal_xyz,
                                                   (float*)d_white, o->stream);
    if (s != MW_OK) return fail(s, or_last_error());
    o->orr.fr_have[0] = !d_height; o->orr.fr_have[1] = !d_disp_xz; o->orr.fr_have[2] = !d_normal_xyz; o->orr.fr_have[3] = !d_white;
    o->orr.frames_last = nframes;
    return MW_OK;
}
// host forms of the steps calls: the frames stay in the handle's frame buffers, then leave over PCIe into the caller's [nframes][...] arrays
static mw_status steps_to_host(mw_ocean* o, const float* delta_time, int32_t nframes, bool rgba, float* const host[4], const char* who) {
    if (!o || !delta_time) return fail(MW_EINVAL, std::string(who) + ": NULL argument");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, std::string(who) + ": OceanRenderer semantics only");
    HIP_TRY(hipSetDevice(o->device));
    const size_t MM = (size_t)o->N * o->N;
    if (!rgba) {
        mw_status s = mw_ocean_generate_texture_steps_device(o, delta_time, nframes, nullptr, nullptr, nullptr, nullptr);
        if (s != MW_OK) return s;
        void* dev[4] = {nullptr, nullptr, nullptr, nullptr};
        if ((s = mw_ocean_frame_textures(o, 0, &dev[0], &dev[1], &dev[2], &dev[3])) != MW_OK) return s;
        const size_t per[4] = {MM, 2 * MM, 3 * MM, MM};
        for (int k = 0; k < 4; k++)
            if (host[k]) HIP_TRY(hipMemcpyAsync(host[k], dev[k], per[k] * (size_t)nframes * sizeof(float), hipMemcpyDeviceToHost, o->stream));
        HIP_TRY(hipStreamSynchronize(o->stream));
        return MW_OK;
    }
    const size_t bytes = MM * 4 * sizeof(float) * (size_t)nframes, stride = align256(bytes);
    void* buf = nullptr;
    int wanted = 0;
    for (int k = 0; k < 4; k++) wanted += host[k] ? 1 : 0;
    mw_status s = scratch_reserve(o, (size_t)(wanted ? wanted : 1) * stride, &buf);
    if (s != MW_OK) return s;
    float* dev[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0, j = 0; k < 4; k++)
        if (host[k]) dev[k] = reinterpret_cast<float*>(static_cast<char*>(buf) + (size_t)(j++) * stride);
    if ((s = mw_ocean_generate_texture_steps_rgba_device(o, delta_time, nframes, dev[0], dev[1], dev[2], dev[3])) != MW_OK) return s;
    for (int k = 0; k < 4; k++)
        if (host[k]) HIP_TRY(hipMemcpyAsync(host[k], dev[k], bytes, hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipStreamSynchronize(o->stream));
    return MW_OK;
}
mw_status mw_ocean_generate_texture_steps(mw_ocean* o, const float* delta_time, int32_t nframes, float* height, float* disp_xz, float* normal_xyz,
                                          float* white) {
    float* const host[4] = {height, disp_xz, normal_xyz, white};
    return steps_to_host(o, delta_time, nframes, false, host, "mw_ocean_generate_texture_steps");
}
mw_status mw_ocean_generate_texture_steps_rgba(mw_ocean* o, const float* delta_time, int32_t nframes, float* height_rgba, float* disp_rgba,
                                               float* normal_rgba, float* white_rgba) {
    float* const host[4] = {height_rgba, disp_rgba, normal_rgba, white_rgba};
    return steps_to_host(o, delta_time, nframes, true, host, "mw_ocean_generate_texture_steps_rgba");
}
mw_status mw_ocean_advance_phase(mw_ocean* o, const float* delta_time, int32_t nframes) {
    if (!o || (!delta_time && nframes > 0)) return fail(MW_EINVAL, "mw_ocean_advance_phase: NULL argument");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, "mw_ocean_advance_phase: OceanRenderer semantics only (FFTMesh time is mw_ocean_set_timer)");
    if (nframes < 0) return fail(MW_EINVAL, "mw_ocean_advance_phase: nframes < 0");
    if (nframes == 0) return MW_OK;
    HIP_TRY(hipSetDevice(o->device));
    mw_status s = or_advance_phase(o->orr, delta_time, nframes, o->stream);
    if (s != MW_OK) return fail(s, or_last_error());
    return MW_OK;
}
int32_t mw_ocean_max_frames(const mw_ocean* o) { return (o && o->sem == MW_SEM_OCEANRENDERER && o->orr.tiles == 1) ? MW_OR_MAX_FRAMES : 0; }
mw_status mw_ocean_frame_textures(mw_ocean* o, int32_t frame, void** d_height, void** d_disp_xz, void** d_normal_xyz, void** d_white) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, "mw_ocean_frame_textures: OceanRenderer semantics only");
    const OrState& s = o->orr;
    if (s.frames_last < 1) return fail(MW_ESTATE, "mw_ocean_frame_textures: no mw_ocean_generate_texture_steps_device call yet");
    if (frame < 0 || frame >= s.frames_last) return fail(MW_EINVAL, "mw_ocean_frame_textures: frame out of range");
    const size_t off = (size_t)frame * s.M * s.M;
    if (s.frames_last == 1) {  // a one-frame call ran the lone-frame plan: the frame is the handle's latest
        if (d_height) *d_height = s.fr_have[0] ? s.out_height : nullptr;
        if (d_disp_xz) *d_disp_xz = s.fr_have[1] ? s.out_disp : nullptr;
        if (d_normal_xyz) *d_normal_xyz = s.fr_have[2] ? s.out_normal : nullptr;
        if (d_white) *d_white = s.fr_have[3] ? s.out_white : nullptr;
        return MW_OK;
    }
    if (d_height) *d_height = s.fr_have[0] ? s.fr_height + off : nullptr;
    if (d_disp_xz) *d_disp_xz = s.fr_have[1] ? s.fr_disp + off : nullptr;
    if (d_normal_xyz) *d_normal_xyz = s.fr_have[2] ? s.fr_normal + 3 * off : nullptr;
    if (d_white) *d_white = s.fr_have[3] ? s.fr_white + off : nullptr;
    return MW_OK;
}
mw_status mw_ocean_generate_texture_steps_rgba_device(mw_ocean* o, const float* delta_time, int32_t nframes, void* d_height_rgba,
                                                      void* d_disp_rgba, void* d_normal_rgba, void* d_white_rgba) {
    if (!o || !delta_time) return fail(MW_EINVAL, "mw_ocean_generate_texture_steps_rgba: NULL argument");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, "mw_ocean_generate_texture_steps_rgba: OceanRenderer semantics only");
    if (nframes < 1 || nframes > mw_ocean_max_frames(o))
        return fail(o->orr.tiles != 1 ? MW_ESTATE : MW_EINVAL, "mw_ocean_generate_texture_steps_rgba: nframes out of range, or a batched handle");
    HIP_TRY(hipSetDevice(o->device));
    o->orr.choppiness = o->p.choppiness;
    mw_status s = nframes == 1 ? or_generate_rgba(o->orr, delta_time[0], (f4*)d_height_rgba, (f4*)d_disp_rgba, (f4*)d_normal_rgba, (f4*)d_white_rgba, o->stream)
                               : or_generate_steps_rgba(o->orr, delta_time, nframes, (f4*)d_height_rgba, (f4*)d_disp_rgba, (f4*)d_normal_rgba,
                                                        (f4*)d_white_rgba, o->stream);
    if (s != MW_OK) return fail(s, or_last_error());
    for (int k = 0; k < 4; k++) o->orr.fr_have[k] = true;
    o->orr.frames_last = nframes;
    return MW_OK;
}

mw_status mw_host_register(void* ptr, size_t bytes) {
    if (!ptr || bytes == 0) return fail(MW_EINVAL, "mw_host_register: NULL pointer or zero size");
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return MW_OK;
}
mw_status mw_host_unregister(void* ptr) {
    if (!ptr) return fail(MW_EINVAL, "mw_host_unregister: NULL pointer");
    HIP_TRY(hipHostUnregister(ptr));
    return MW_OK;
}

mw_status mw_ocean_generate_texture_rgba_device(mw_ocean* o, float delta_time, void* d_height_rgba, void* d_disp_rgba,
                                                void* d_normal_rgba, void* d_white_rgba) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, "mw_ocean_generate_texture_rgba: OceanRenderer semantics only");
    HIP_TRY(hipSetDevice(o->device));
    o->orr.choppiness = o->p.choppiness;
    mw_status s = or_generate_rgba(o->orr, delta_time, (f4*)d_height_rgba, (f4*)d_disp_rgba, (f4*)d_normal_rgba,
                                   (f4*)d_white_rgba, o->stream);
    if (s != MW_OK) return fail(s, or_last_error());
    return MW_OK;
}

mw_status mw_ocean_generate_texture_rgba(mw_ocean* o, float delta_time, float* height_rgba, float* disp_rgba,
                                         float* normal_rgba, float* white_rgba) {
    if (!o) return fail(MW_EINVAL, "NULL handle");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, "mw_ocean_generate_texture_rgba: OceanRenderer semantics only");
    HIP_TRY(hipSetDevice(o->device));
    const size_t bytes = (size_t)o->N * o->N * 4 * sizeof(float) * o->orr.tiles, stride = align256(bytes);
    float* host[4] = {height_rgba, disp_rgba, normal_rgba, white_rgba};
    float* dev[4] = {nullptr, nullptr, nullptr, nullptr};
    void* buf = nullptr;
    mw_status s = scratch_reserve(o, 4 * stride, &buf);
    if (s != MW_OK) return s;
    for (int k = 0; k < 4; k++)
        if (host[k]) dev[k] = reinterpret_cast<float*>(static_cast<char*>(buf) + k * stride);
    if ((s = mw_ocean_generate_texture_rgba_device(o, delta_time, dev[0], dev[1], dev[2], dev[3])) != MW_OK) return s;
    for (int k = 0; k < 4; k++)
        if (host[k]) HIP_TRY(hipMemcpyAsync(host[k], dev[k], bytes, hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipStreamSynchronize(o->stream));
    return MW_OK;
}

mw_status mw_ocean_displace_mesh_device(mw_ocean* o, void* d_vertices_xyz, void* d_normals_xyz, void* d_colors) {
    if (!o || !d_vertices_xyz) return fail(MW_EINVAL, "mw_ocean_displace_mesh: NULL argument");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, "mw_ocean_displace_mesh: OceanRenderer semantics only");
    HIP_TRY(hipSetDevice(o->device));
    mw_status s = or_displace_mesh(o->orr, o->p.resolution, o->p.unit_width, (float*)d_vertices_xyz, (float*)d_normals_xyz,
                                   (float*)d_colors, o->stream);
    if (s != MW_OK) return fail(s, or_last_error());
    return MW_OK;
}

mw_status mw_ocean_displace_mesh(mw_ocean* o, float* vertices_xyz, float* normals_xyz, float* colors) {
    if (!o || !vertices_xyz) return fail(MW_EINVAL, "mw_ocean_displace_mesh: NULL argument");
    if (o->sem != MW_SEM_OCEANRENDERER) return fail(MW_ESTATE, "mw_ocean_displace_mesh: OceanRenderer semantics only");
    HIP_TRY(hipSetDevice(o->device));
    const size_t nv = (size_t)o->p.resolution * o->p.resolution * o->orr.tiles, b3 = align256(nv * 3 * sizeof(float));
    void* buf = nullptr;
    mw_status s = scratch_reserve(o, 2 * b3 + align256(nv * sizeof(float)), &buf);
    if (s != MW_OK) return s;
    char* base = static_cast<char*>(buf);
    float* dv = reinterpret_cast<float*>(base);
    float* dn = normals_xyz ? reinterpret_cast<float*>(base + b3) : nullptr;
    float* dc = colors ? reinterpret_cast<float*>(base + 2 * b3) : nullptr;
    if ((s = mw_ocean_displace_mesh_device(o, dv, dn, dc)) != MW_OK) return s;
    HIP_TRY(hipMemcpyAsync(vertices_xyz, dv, nv * 3 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (dn) HIP_TRY(hipMemcpyAsync(normals_xyz, dn, nv * 3 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (dc) HIP_TRY(hipMemcpyAsync(colors, dc, nv * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipStreamSynchronize(o->stream));
    return MW_OK;
}

// per-launch durations -> (mean, median, p10, p90, min, max), milliseconds
static void launch_stats(std::vector<float>& v, float* out6) {
    std::sort(v.begin(), v.end());
    const size_t n = v.size();
    double acc = 0.0;
    for (float x : v) acc += x;
    auto pct = [&](double q) { return v[(size_t)std::min<double>((double)n - 1.0, std::floor(q * (double)(n - 1) + 0.5))]; };
    out6[0] = (float)(acc / (double)n); out6[1] = pct(0.5); out6[2] = pct(0.1); out6[3] = pct(0.9); out6[4] = v.front(); out6[5] = v.back();
}
static mw_status profile_kernels_impl(mw_ocean* o, int32_t nsteps, int32_t iters, float* ms_out, float* stats_out, const char** names_out,
                                      int32_t* nkernels);
mw_status mw_ocean_profile_kernels(mw_ocean* o, int32_t nsteps, int32_t iters, float* ms_out, const char** names_out,
                                   int32_t* nkernels) {
    if (!ms_out) return fail(MW_EINVAL, "mw_ocean_profile_kernels: bad argument");
    return profile_kernels_impl(o, nsteps, iters, ms_out, nullptr, names_out, nkernels);
}
mw_status mw_ocean_profile_kernels_stats(mw_ocean* o, int32_t nsteps, int32_t iters, float* stats_out, const char** names_out,
                                         int32_t* nkernels) {
    if (!stats_out) return fail(MW_EINVAL, "mw_ocean_profile_kernels_stats: bad argument");
    float ms[4] = {0.f, 0.f, 0.f, 0.f};
    return profile_kernels_impl(o, nsteps, iters, ms, stats_out, names_out, nkernels);
}
static mw_status profile_kernels_impl(mw_ocean* o, int32_t nsteps, int32_t iters, float* ms_out, float* stats_out, const char** names_out,
                                      int32_t* nkernels) {
    if (!o || !ms_out || !nkernels || iters < 1) return fail(MW_EINVAL, "mw_ocean_profile_kernels: bad argument");
    if (nsteps < 1 || nsteps > MW_MAX_BATCH) return fail(MW_EINVAL, "nsteps out of range");
    HIP_TRY(hipSetDevice(o->device));
    HIP_TRY(hipStreamSynchronize(o->stream));
    if (o->sem == MW_SEM_OCEANRENDERER) {
        // nsteps frames per enqueue (1: the lone-frame plan), in situ: the launches of a call follow one another as in
        // mw_ocean_generate_texture[_steps]_device, a HIP event between every two; the frames stay in the handle.  The phase ADVANCES.
        if (nsteps > 1 && o->orr.tiles != 1) return fail(MW_ESTATE, "mw_ocean_profile_kernels: a batched handle advances one frame per call");
        static const char* rnames[4] = {"k_or_pass1 (dispersion + spectrum + transform along py)", "k_or_pass2 (transform along px, height / displacement)",
                                        "k_or_normal_white", "copies (k_or_copy_frame: the last frame becomes the handle's latest)"};
        static const char* snames[4] = {"k_or_pass1_steps (phase chain + spectra + transform along py, all frames)", rnames[1], rnames[2], rnames[3]};
        float dts[MW_OR_MAX_FRAMES];
        for (int k = 0; k < MW_OR_MAX_FRAMES; k++) dts[k] = 1.0f / 60.0f;
        o->orr.choppiness = o->p.choppiness;
        auto call = [&](hipEvent_t* ev) {
            return nsteps == 1 ? or_generate(o->orr, dts[0], nullptr, nullptr, nullptr, nullptr, o->stream, ev)
                               : or_generate_steps(o->orr, dts, nsteps, nullptr, nullptr, nullptr, nullptr, o->stream, ev);
        };
        mw_status s = MW_OK;
        {
            const auto t0 = std::chrono::steady_clock::now();
            do {
                for (int w = 0; w < 2 && s == MW_OK; w++) s = call(nullptr);
                hipStreamSynchronize(o->stream);
            } while (s == MW_OK && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.12);
        }
        if (s != MW_OK) return fail(s, or_last_error());
        // events of one call, per chunk j of frames (one chunk in the lone-frame plan): [3j] before its spectrum launch, [3j + 1] after it,
        // [3j + 2] after its pass 2, [3j + 3] after its normal / whitecap pass; [3 nch + 1] after the copies
        const int nch = nsteps == 1 ? 1 : or_steps_chunks(o->orr.M, nsteps);
        const size_t per_it = 2 + 3 * (size_t)nch;
        std::vector<hipEvent_t> ev(per_it * (size_t)iters);
        for (auto& e : ev) hipEventCreate(&e);
        for (int it = 0; it < iters && s == MW_OK; it++) s = call(&ev[per_it * (size_t)it]);
        hipStreamSynchronize(o->stream);
        std::vector<float> per[4];
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int it = 0; it < iters && s == MW_OK; it++) {
            hipEvent_t* e = &ev[per_it * (size_t)it];
            float m[4] = {0.f, 0.f, 0.f, 0.f}, x = 0.f;
            for (int j = 0; j < nch; j++)
                for (int k = 0; k < 3; k++) { hipEventElapsedTime(&x, e[3 * j + k], e[3 * j + k + 1]); m[k] += x; }
            hipEventElapsedTime(&m[3], e[3 * nch], e[3 * nch + 1]);
            for (int k = 0; k < 4; k++) { acc[k] += m[k]; per[k].push_back(m[k]); }
        }
        for (auto& e : ev) hipEventDestroy(e);
        if (s != MW_OK) return fail(s, or_last_error());
        for (int k = 0; k < 4; k++) {
            ms_out[k] = (float)(acc[k] / iters);
            if (names_out) names_out[k] = (nsteps == 1 ? rnames : snames)[k];
            if (stats_out) launch_stats(per[k], stats_out + 6 * k);
        }
        *nkernels = 4;
        o->orr.frames_last = nsteps;
        for (int k = 0; k < 4; k++) o->orr.fr_have[k] = true;
        return MW_OK;
    }
    if (!o->use_fft) {  // direct-sum path: kernel 0 = the four GEMM launches of one step, kernel 1 = spectrum + assembly + whitecap
        if (nsteps != 1) return fail(MW_EINVAL, "mw_ocean_profile_kernels: the direct-sum path evaluates one step per enqueue");
        static const char* gnames[2] = {"k_gemm_f32_mfma (4 launches: z sum, x sum)", "k_direct_spec + k_direct_assemble + k_direct_white"};
        static const char* znames[2] = {"k_czt (2 launches: spectrum + chirp-z along j, chirp-z along i)", "k_czt_assemble_white"};
        static const char* fnames[2] = {"k_czt (spectrum + chirp-z along j)", "k_czt_rows_assemble (chirp-z along i + vertices, normals, whitecap: one launch)"};
        static const char* onames[2] = {"(no separate launch)", "k_czt_one (both axes + vertices, normals, whitecap: one workgroup, one launch)"};
        const char* const* dnames = gnames;
        if (o->direct.use_czt) {  // the names follow the plan czt_evaluate runs (czt_plan: the one place that decides)
            const CztPlan plan = czt_plan(o->direct.czt, o->N);
            dnames = plan == CZT_PLAN_ONE ? onames : (plan == CZT_PLAN_TWO ? fnames : znames);
        }
        hipEvent_t ev[4];
        for (auto& e : ev) hipEventCreate(&e);
        hipError_t he = hipSuccess;
        for (int w = 0; w < 5 && he == hipSuccess; w++)
            he = direct_evaluate(o->direct, consts_of(o), o->h0, o->h0c, 1.0f, o->s_vert, o->s_norm, o->s_white, 1, o->stream);
        double acc[2] = {0.0, 0.0};
        std::vector<float> per[2];
        for (int it = 0; it < iters && he == hipSuccess; it++) {
            he = direct_evaluate(o->direct, consts_of(o), o->h0, o->h0c, 1.0f + (float)it / 60.f, o->s_vert, o->s_norm, o->s_white, 1, o->stream, ev);
            hipEventRecord(ev[3], o->stream);
            hipEventSynchronize(ev[3]);
            float a = 0.f, g = 0.f, b = 0.f;
            hipEventElapsedTime(&a, ev[0], ev[1]);
            hipEventElapsedTime(&g, ev[1], ev[2]);
            hipEventElapsedTime(&b, ev[2], ev[3]);
            acc[0] += g;
            acc[1] += a + b;
            per[0].push_back(g);
            per[1].push_back(a + b);
        }
        for (auto& e : ev) hipEventDestroy(e);
        if (he != hipSuccess) return fail(MW_EDEVICE, std::string("direct-sum profile: ") + hipGetErrorString(he));
        for (int k = 0; k < 2; k++) {
            ms_out[k] = (float)(acc[k] / iters);
            if (names_out) names_out[k] = dnames[k];
            if (stats_out) launch_stats(per[k], stats_out + 6 * k);
        }
        *nkernels = 2;
        return MW_OK;
    }
    mw_status s = ensure_exchange(o, nsteps);
    if (s != MW_OK) return s;
    const size_t NN = (size_t)o->N * o->N;
    float *dv = nullptr, *dn = nullptr, *dw = nullptr;
    if (nsteps == 1) { dv = o->s_vert; dn = o->s_norm; dw = o->s_white; }
    else {
        if ((s = dmalloc(&dv, NN * 3 * nsteps)) != MW_OK || (s = dmalloc(&dn, NN * 3 * nsteps)) != MW_OK ||
            (s = dmalloc(&dw, NN * nsteps)) != MW_OK) { hipFree(dv); hipFree(dn); hipFree(dw); return s; }
    }
    StepTimes tm;
    for (int k = 0; k < nsteps; k++) tm.t[k] = 1.0f + (float)k / 60.f;
    // In-situ timing: the two kernels alternate exactly as in mw_ocean_evaluate_device (pass 2 of a batch follows
    // its pass 1), with a HIP event between every launch on the launch stream.
    static const char* names[2] = {"k_pass1 (h~ + transform along i)", "k_pass2 (transform along j + epilogue)"};
    std::vector<hipEvent_t> ev(2 * iters + 1);
    for (auto& e : ev) hipEventCreate(&e);
    // warm-up: the allocations above idle the device for milliseconds and the first ~10 ms after idle run at reduced
    // clocks (bench.py preheats its timed region for the same reason): 120 ms of the same two kernels, at least 2 rounds
    {
        const auto t0 = std::chrono::steady_clock::now();
        int rounds = 0;
        do {
            for (int w = 0; w < 2 && s == MW_OK; w++) {
                s = launch_pass1(o, tm, nsteps, o->stream);
                if (s == MW_OK) s = launch_pass2(o, nsteps, dv, dn, dw, 1);
            }
            hipStreamSynchronize(o->stream);
            rounds++;
        } while (s == MW_OK && (rounds < 1 || std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.12));
    }
    hipEventRecord(ev[0], o->stream);
    for (int it = 0; it < iters && s == MW_OK; it++) {
        s = launch_pass1(o, tm, nsteps, o->stream);
        hipEventRecord(ev[2 * it + 1], o->stream);
        if (s == MW_OK) s = launch_pass2(o, nsteps, dv, dn, dw, 1);
        hipEventRecord(ev[2 * it + 2], o->stream);
    }
    hipEventSynchronize(ev[2 * iters]);
    double acc[2] = {0.0, 0.0};
    std::vector<float> per[2];
    for (int it = 0; it < iters && s == MW_OK; it++) {
        float m1 = 0.f, m2 = 0.f;
        hipEventElapsedTime(&m1, ev[2 * it], ev[2 * it + 1]);
        hipEventElapsedTime(&m2, ev[2 * it + 1], ev[2 * it + 2]);
        acc[0] += m1;
        acc[1] += m2;
        per[0].push_back(m1);
        per[1].push_back(m2);
    }
    for (int k = 0; k < 2; k++) {
        ms_out[k] = (float)(acc[k] / iters);
        if (names_out) names_out[k] = names[k];
        if (stats_out && s == MW_OK) launch_stats(per[k], stats_out + 6 * k);
    }
    for (auto& e : ev) hipEventDestroy(e);
    if (nsteps != 1) { hipFree(dv); hipFree(dn); hipFree(dw); }
    *nkernels = 2;
    return s;
}

// test hook: omega(i,j)*t exactly as the kernels form it (bit-exactness check vs the oracle)
mw_status mw_debug_omega_t(mw_ocean* o, float t, float* out_host) {
    if (!o || !out_host) return fail(MW_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(o->device));
    const int N = o->N;
    float* d = nullptr;
    mw_status s = dmalloc(&d, (size_t)N * N);
    if (s != MW_OK) return s;
    hipLaunchKernelGGL(k_omega_t, dim3((N * N + 255) / 256), dim3(256), 0, o->stream, N, o->p.length, o->p.gravity, t, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, d, sizeof(float) * N * N, hipMemcpyDeviceToHost, o->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(o->stream);
    hipFree(d);
    return e == hipSuccess ? MW_OK : fail(MW_EDEVICE, std::string("mw_debug_omega_t: ") + hipGetErrorString(e));
}

// test hook: one EvaluateWaves(t) that also returns hds = (d.x, d.z) exactly as the kernels hold it (S/FFTMesh.cs:247), so
// that the whitecap stage -- forward differences, the i = N-1 / j = N-1 edge rules (:258-274), the halo rows handed
// between workgroups -- can be compared BIT FOR BIT with the oracle's float32 whitecap of the same hds and normals
mw_status mw_debug_evaluate_hds(mw_ocean* o, float t, float* vertices_xyz, float* normals_xyz, float* colors_rgba, float* hds_xy) {
    if (!o || !hds_xy) return fail(MW_EINVAL, "mw_debug_evaluate_hds: NULL argument");
    if (o->sem != MW_SEM_FFTMESH) return fail(MW_ESTATE, "mw_debug_evaluate_hds: FFTMesh semantics only");
    HIP_TRY(hipSetDevice(o->device));
    const size_t NN = (size_t)o->N * o->N;
    cf* dh = nullptr;
    if (o->use_fft) {
        void* buf = nullptr;
        mw_status s = scratch_reserve(o, NN * sizeof(cf), &buf);
        if (s != MW_OK) return s;
        dh = static_cast<cf*>(buf);
        if ((s = ensure_exchange(o, 1)) != MW_OK) return s;
        StepTimes tm;
        tm.t[0] = t;
        if ((s = launch_pass1(o, tm, 1, o->stream)) != MW_OK) return s;
        if ((s = launch_pass2(o, 1, o->s_vert, o->s_norm, o->s_white, 4, dh)) != MW_OK) return s;
    } else {
        mw_status s = mw_ocean_evaluate_device(o, &t, 1, o->s_vert, o->s_norm, o->s_white, MW_OUT_COLOR_RGBA);
        if (s != MW_OK) return s;
        dh = o->direct.hds;
    }
    if (vertices_xyz) HIP_TRY(hipMemcpyAsync(vertices_xyz, o->s_vert, NN * 3 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (normals_xyz) HIP_TRY(hipMemcpyAsync(normals_xyz, o->s_norm, NN * 3 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (colors_rgba) HIP_TRY(hipMemcpyAsync(colors_rgba, o->s_white, NN * 4 * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipMemcpyAsync(hds_xy, dh, NN * sizeof(cf), hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipStreamSynchronize(o->stream));
    return MW_OK;
}

// measurement hook: the pass-1 time group an enqueue of nsteps uses (bench.py prints what it timed)
int32_t mw_debug_pass1_time_group(mw_ocean* o, int32_t nsteps) {
    return (o && o->sem == MW_SEM_FFTMESH && o->use_fft && nsteps >= 1 && nsteps <= MW_MAX_BATCH) ? p1_time_group(o, nsteps) : 0;
}

// test hooks: the stored omega table and the device sincos
__global__ void k_dbg_sincos(const float* x, int n, float* sn, float* cs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sincos_f32(x[i], &sn[i], &cs[i]);
}
__global__ void k_dbg_sincos_fast(const float* x, int n, float* sn, float* cs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sincos_fast_f32(x[i], &sn[i], &cs[i]);
}
// test hook: ONE wave applies wave_transpose4 (v_permlane16_swap / v_permlane32_swap, mw_math.h) to 64 lanes x 16 complex slots
__global__ __launch_bounds__(64) void k_dbg_wave_transpose4(cf* io) {
    cf x[16];
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = io[threadIdx.x * 16 + r];
    wave_transpose4<16>(x);
#pragma unroll
    for (int r = 0; r < 16; r++) io[threadIdx.x * 16 + r] = x[r];
}
static mw_status debug_sincos(const float* x_host, int32_t n, float* s_host, float* c_host, bool fast);
mw_status mw_debug_sincos(const float* x_host, int32_t n, float* s_host, float* c_host) {
    return debug_sincos(x_host, n, s_host, c_host, false);
}
mw_status mw_debug_sincos_fast(const float* x_host, int32_t n, float* s_host, float* c_host) {
    return debug_sincos(x_host, n, s_host, c_host, true);
}
mw_status mw_debug_set_switch(const char* name, int32_t value) {
    const int k = switch_index(name);
    if (k < 0) return fail(MW_EINVAL, std::string("mw_debug_set_switch: unknown switch ") + (name ? name : "(null)"));
    switch_table().v[k].store(value);
    return MW_OK;
}
int32_t mw_debug_get_switch(const char* name) {
    const int k = switch_index(name);
    return k < 0 ? INT32_MIN : sw((Switch)k);
}
mw_status mw_debug_wave_transpose4(float* inout_host) {
    if (!inout_host) return fail(MW_EINVAL, "NULL argument");
    cf* d = nullptr;
    mw_status s = dmalloc(&d, 64 * 16);
    if (s != MW_OK) return s;
    hipError_t e = hipMemcpy(d, inout_host, sizeof(cf) * 64 * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_dbg_wave_transpose4, dim3(1), dim3(64), 0, 0, d); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpy(inout_host, d, sizeof(cf) * 64 * 16, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail(MW_EDEVICE, std::string("mw_debug_wave_transpose4: ") + hipGetErrorString(e));
    return MW_OK;
}
static mw_status debug_sincos(const float* x_host, int32_t n, float* s_host, float* c_host, bool fast) {
    if (!x_host || !s_host || !c_host || n < 1) return fail(MW_EINVAL, "mw_debug_sincos: bad argument");
    float *dx = nullptr, *ds = nullptr, *dc = nullptr;
    hipError_t e = hipMalloc((void**)&dx, 4 * (size_t)n);
    if (e == hipSuccess) e = hipMalloc((void**)&ds, 4 * (size_t)n);
    if (e == hipSuccess) e = hipMalloc((void**)&dc, 4 * (size_t)n);
    if (e == hipSuccess) e = hipMemcpy(dx, x_host, 4 * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (fast) hipLaunchKernelGGL(k_dbg_sincos_fast, dim3((n + 255) / 256), dim3(256), 0, 0, dx, n, ds, dc);
        else hipLaunchKernelGGL(k_dbg_sincos, dim3((n + 255) / 256), dim3(256), 0, 0, dx, n, ds, dc);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(s_host, ds, 4 * (size_t)n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(c_host, dc, 4 * (size_t)n, hipMemcpyDeviceToHost);
    hipFree(dx); hipFree(ds); hipFree(dc);
    return e == hipSuccess ? MW_OK : fail(MW_EDEVICE, std::string("mw_debug_sincos: ") + hipGetErrorString(e));
}
mw_status mw_debug_stream_read(int64_t bytes, int32_t width, int32_t iters) {
    void* buf = nullptr;
    float* sink = nullptr;
    HIP_TRY(hipMalloc(&buf, (size_t)bytes));
    HIP_TRY(hipMalloc((void**)&sink, 4 * 4096));
    HIP_TRY(hipMemset(buf, 0, (size_t)bytes));
    for (int it = 0; it < iters; it++) {
        if (width == 4) k_dbg_stream<float><<<2048, 256>>>((const float*)buf, (size_t)bytes / 4, sink);
        else if (width == 8) k_dbg_stream<cf><<<2048, 256>>>((const cf*)buf, (size_t)bytes / 8, sink);
        else k_dbg_stream<f4><<<2048, 256>>>((const f4*)buf, (size_t)bytes / 16, sink);
    }
    hipError_t e = hipDeviceSynchronize();
    hipFree(buf);
    hipFree(sink);
    return e == hipSuccess ? MW_OK : fail(MW_EDEVICE, "stream_read failed");
}

mw_status mw_debug_get_omega(mw_ocean* o, float* out_host) {  // [j][i] layout
    if (!o || !o->Om) return fail(MW_EINVAL, "no omega table");
    HIP_TRY(hipMemcpy(out_host, o->Om, sizeof(float) * o->N * o->N, hipMemcpyDeviceToHost));
    return MW_OK;
}

#ifdef MW_TIMING
mw_status mw_debug_get_stamps(long long* out_host) {
    HIP_TRY(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(long long) * 2 * 64 * 16 * 32));
    return MW_OK;
}
#endif

#include "tiles.inc"

// ---- pond -------------------------------------------------------------------------------------
}  // extern "C" (reopened below)
// Device staging of the handle-less host-pointer pond entry points: one grow-only buffer per device, held for the whole
// (synchronous) call -- no hipMalloc / hipFree per frame, like the handle entry points (scratch_reserve).
struct PondScratch {
    std::mutex mu[64];
    void* buf[64] = {};
    size_t cap[64] = {};
    struct Lock {
        PondScratch& s;
        int d;
        Lock(PondScratch& s_, int device) : s(s_), d(device & 63) { s.mu[d].lock(); }
        ~Lock() { s.mu[d].unlock(); }
        void* reserve(size_t bytes) {
            if (s.cap[d] < bytes) {
                if (s.buf[d]) { (void)hipDeviceSynchronize(); (void)hipFree(s.buf[d]); s.buf[d] = nullptr; s.cap[d] = 0; }
                if (hipMalloc(&s.buf[d], bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
                s.cap[d] = bytes;
            }
            return s.buf[d];
        }
    };
};
static PondScratch g_pond_scratch;
extern "C" {
mw_status mw_gerstner_displace_device(const void* d_pos_xyz, int64_t nverts, const float* waves, int32_t nwaves,
                                      float amplitude, float frequency, float steepness, float t, void* d_out_xyz,
                                      void* hip_stream) {
    if (!d_pos_xyz || !d_out_xyz || !waves) return fail(MW_EINVAL, "mw_gerstner_displace_device: NULL argument");
    if (nwaves < 1 || nwaves > MW_GERSTNER_MAX_WAVES) return fail(MW_EINVAL, "nwaves must be in [1,16]");
    if (nverts < 0) return fail(MW_EINVAL, "nverts < 0");
    if (nverts == 0) return MW_OK;
    hipError_t e = gerstner_launch((const float*)d_pos_xyz, nverts, waves, nwaves, amplitude, frequency, steepness, t,
                                   (float*)d_out_xyz, reinterpret_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail(MW_EDEVICE, std::string("gerstner launch: ") + hipGetErrorString(e));
    return MW_OK;
}

int32_t mw_gerstner_max_steps(int32_t nwaves) {
    if (nwaves != 4 && nwaves != 8) return 0;
    const int m = MW_GERSTNER_PHASES / nwaves;
    return m < 32 ? m : 32;
}

mw_status mw_gerstner_displace_steps_device(const void* d_pos_xyz, int64_t nverts, const float* waves, int32_t nwaves,
                                            float amplitude, float frequency, float steepness, const float* t,
                                            int32_t nsteps, void* d_out_xyz, void* hip_stream) {
    if (!d_pos_xyz || !d_out_xyz || !waves || !t) return fail(MW_EINVAL, "mw_gerstner_displace_steps_device: NULL argument");
    const int maxs = mw_gerstner_max_steps(nwaves);
    if (maxs == 0) return fail(MW_EINVAL, "mw_gerstner_displace_steps_device: nwaves must be 4 or 8");
    if (nsteps < 1 || nsteps > maxs) return fail(MW_EINVAL, "mw_gerstner_displace_steps_device: nsteps out of range");
    if (nverts < 0) return fail(MW_EINVAL, "nverts < 0");
    if (nverts == 0) return MW_OK;
    hipError_t e = gerstner_launch_steps((const float*)d_pos_xyz, nverts, waves, nwaves, amplitude, frequency, steepness, t, nsteps,
                                         (float*)d_out_xyz, reinterpret_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail(MW_EDEVICE, std::string("gerstner launch: ") + hipGetErrorString(e));
    return MW_OK;
}

mw_status mw_gerstner_displace(const float* pos_xyz, int64_t nverts, const float* waves, int32_t nwaves, float amplitude,
                               float frequency, float steepness, float t, float* out_xyz, int32_t device) {
    if (!pos_xyz || !out_xyz || !waves) return fail(MW_EINVAL, "mw_gerstner_displace: NULL argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(MW_EDEVICE, "mw_gerstner_displace: no HIP device visible (no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(MW_EINVAL, "bad device ordinal");
    if (nverts <= 0) return nverts == 0 ? MW_OK : fail(MW_EINVAL, "nverts < 0");
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = (size_t)nverts * 3 * sizeof(float), stride = align256(bytes);
    PondScratch::Lock lk(g_pond_scratch, device);
    char* base = static_cast<char*>(lk.reserve(2 * stride));
    if (!base) return fail(MW_ENOMEM, "mw_gerstner_displace: device staging buffer");
    float *dp = reinterpret_cast<float*>(base), *dq = reinterpret_cast<float*>(base + stride);
    HIP_TRY(hipMemcpy(dp, pos_xyz, bytes, hipMemcpyHostToDevice));
    mw_status s = mw_gerstner_displace_device(dp, nverts, waves, nwaves, amplitude, frequency, steepness, t, dq, nullptr);
    if (s != MW_OK) return s;
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipMemcpy(out_xyz, dq, bytes, hipMemcpyDeviceToHost));
    return MW_OK;
}

static mw_status pond_params_of(const mw_pond_params* p, PondParams* P, const char* who) {
    if (!p) return fail(MW_EINVAL, std::string(who) + ": NULL params");
    if (p->mode != MW_POND_WAVE && p->mode != MW_POND_GERSTNER && p->mode != MW_POND_GERSTNER_LEVEL_ONE)
        return fail(MW_EINVAL, std::string(who) + ": unknown displacement mode");
    P->mode = p->mode; P->amplitude = p->amplitude; P->frequency = p->frequency; P->speed = p->speed;
    P->steepness = p->steepness; P->smoothing = p->smoothing;
    for (int i = 0; i < 4; i++) { P->wspeed[i] = p->wspeed[i]; P->dir_ab[i] = p->dir_ab[i]; P->dir_cd[i] = p->dir_cd[i]; }
    return MW_OK;
}

mw_status mw_pond_displace_device(const mw_pond_params* p, const void* d_pos_xyz, int64_t nverts, float t, void* d_out_xyz,
                                  void* d_out_normal_xyz, void* hip_stream) {
    PondParams P;
    mw_status s = pond_params_of(p, &P, "mw_pond_displace_device");
    if (s != MW_OK) return s;
    if (nverts < 0) return fail(MW_EINVAL, "nverts < 0");
    if (nverts == 0) return MW_OK;
    if (!d_pos_xyz || !d_out_xyz) return fail(MW_EINVAL, "mw_pond_displace_device: NULL argument");
    hipError_t e = pond_launch(P, (const float*)d_pos_xyz, nverts, t, (float*)d_out_xyz, (float*)d_out_normal_xyz,
                               reinterpret_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail(MW_EDEVICE, std::string("pond launch: ") + hipGetErrorString(e));
    return MW_OK;
}

mw_status mw_pond_displace(const mw_pond_params* p, const float* pos_xyz, int64_t nverts, float t, float* out_xyz,
                           float* out_normal_xyz, int32_t device) {
    PondParams P;
    mw_status s = pond_params_of(p, &P, "mw_pond_displace");
    if (s != MW_OK) return s;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(MW_EDEVICE, "mw_pond_displace: no HIP device visible (no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(MW_EINVAL, "bad device ordinal");
    if (nverts <= 0) return nverts == 0 ? MW_OK : fail(MW_EINVAL, "nverts < 0");
    if (!pos_xyz || !out_xyz) return fail(MW_EINVAL, "mw_pond_displace: NULL argument");
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = (size_t)nverts * 3 * sizeof(float), stride = align256(bytes);
    PondScratch::Lock lk(g_pond_scratch, device);
    char* base = static_cast<char*>(lk.reserve((out_normal_xyz ? 3 : 2) * stride));
    if (!base) return fail(MW_ENOMEM, "mw_pond_displace: device staging buffer");
    float *dp = reinterpret_cast<float*>(base), *dq = reinterpret_cast<float*>(base + stride);
    float* dn = out_normal_xyz ? reinterpret_cast<float*>(base + 2 * stride) : nullptr;
    HIP_TRY(hipMemcpy(dp, pos_xyz, bytes, hipMemcpyHostToDevice));
    if ((s = mw_pond_displace_device(p, dp, nverts, t, dq, dn, nullptr)) != MW_OK) return s;
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipMemcpy(out_xyz, dq, bytes, hipMemcpyDeviceToHost));
    if (dn) HIP_TRY(hipMemcpy(out_normal_xyz, dn, bytes, hipMemcpyDeviceToHost));
    return MW_OK;
}

}  // extern "C"
