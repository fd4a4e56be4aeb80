// ocean_renderer_device.h -- __global__ wrappers and host-side state of MW_SEM_OCEANRENDERER (device build only).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mistral_water.h"
#include "ocean_renderer_kernels.h"
#include "mw_switches.h"

namespace mw {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device), safe from any number of host threads
struct AttrOnce {
    std::once_flag once[64];
    hipError_t res[64];
    hipError_t set(const void* fn, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const int d = dev & 63;
        std::call_once(once[d], [&] { res[d] = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
        return res[d];
    }
};

// `tiles` independent oceans (seed, seed + 1, ...) share one handle: every buffer below carries a leading tile axis and
// every kernel one more grid dimension, so a GenerateTexture() of all tiles is still three launches (mw_ocean_create_batch).
// A frame of ONE 1024^2 texture is latency-bound (three launches of 256-768 workgroups); the phase recurrence forbids
// batching in time, so the tile axis is what fills the machine.  omT (a function of k alone) is shared by the tiles.
struct OrState {
    int M = 0;
    int tiles = 1;
    OrConsts c{};
    float mult = 1.f, choppiness = 0.f;
    f4* initT = nullptr;
    f4* PQT = nullptr;     // [px][py] Hermitian parts (P, Q) of the initial spectrum: the packed plan (or_prep_element), rebuilt with initT
    bool phase_sym = true; // the phase texture equals its mirror image (true from creation on; mw_ocean_set_phase checks what it is given)
    float* omT = nullptr;  // [px][py] angular frequency (or_omega), fixed per handle
    float *phaseT = nullptr, *phaseT2 = nullptr;  // current phase / next phase (swapped after every frame)
    cf *TW = nullptr, *E = nullptr;
    float *out_height = nullptr, *out_disp_g = nullptr, *out_normal = nullptr, *out_white = nullptr;
    cf* out_disp_cf = nullptr;
    float* out_disp = nullptr;  // alias of out_disp_cf as floats (r, b)
    float *out_height_g = nullptr, *out_disp_a = nullptr;  // Im h, Im Dz: written only once the RGBA layout was asked for
    bool want_imag = false, have_imag = false, have_frame = false;
    // mw_ocean_generate_texture_steps_device: [frames_cap] frames of the exchange buffer and of the textures the caller did not ask for
    // (disp.g always: OceanNormal reads it, no entry point hands it out alone); grown on demand, never shrunk
    int frames_cap = 0, frames_last = 0;  // capacity / frames of the latest steps call
    cf* fr_E = nullptr;
    float *fr_height = nullptr, *fr_disp_g = nullptr, *fr_normal = nullptr, *fr_white = nullptr, *fr_height_g = nullptr, *fr_disp_a = nullptr;
    cf* fr_disp = nullptr;
    bool fr_have[4] = {false, false, false, false};  // which textures the latest steps call kept here
};
static thread_local std::string g_or_err;
static inline const char* or_last_error() { return g_or_err.c_str(); }

__global__ void k_or_omega(OrConsts c, float* omT) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= c.M * c.M) return;
    omT[idx] = or_omega(c, idx / c.M, idx % c.M);  // transposed enumeration [px][py]
}
__global__ void k_or_init(int M, float length, float wind_x, float wind_y, float amp, float gravity, uint64_t seed, f4* initT,
                          float* phaseT) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * M) return;
    const size_t toff = (size_t)blockIdx.y * M * M;  // tile blockIdx.y: seed + tile
    // idx enumerates the transposed array (px major) so the writes coalesce
    or_init_element(M, length, wind_x, wind_y, amp, gravity, seed + blockIdx.y, idx / M, idx % M, initT + toff,
                    phaseT ? phaseT + toff : nullptr);
}

// initialTexture <-> (h0, h0conj) arrays in the reference's texel order idx = py*M + px
__global__ void k_or_set_init(int M, const cf* h0, const cf* h0c, f4* initT, float* phaseT) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * M) return;
    const size_t toff = (size_t)blockIdx.y * M * M;
    h0 += toff; h0c += toff; initT += toff; phaseT += toff;
    const int px = idx / M, py = idx % M;  // transposed enumeration: coalesced writes
    const cf a = h0[(size_t)py * M + px], b = h0c[(size_t)py * M + px];
    f4 v; v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y;
    initT[idx] = v;
    phaseT[idx] = 0.f;  // RenderInitial does not touch the phase in the reference; a fresh spectrum restarts it here
}
__global__ void k_or_get_init(int M, const f4* initT, cf* h0, cf* h0c) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * M) return;
    const size_t toff = (size_t)blockIdx.y * M * M;
    initT += toff; h0 += toff; h0c += toff;
    const int py = idx / M, px = idx % M;
    const f4 v = initT[(size_t)px * M + py];
    h0[idx] = mk(v.x, v.y);
    h0c[idx] = mk(v.z, v.w);
}

__global__ void k_or_prep(int M, const f4* initT, f4* PQT) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * M) return;
    const size_t toff = (size_t)blockIdx.y * M * M;  // tile blockIdx.y
    or_prep_element(M, idx / M, idx % M, initT + toff, PQT + toff);
}
#ifndef MW_OR_P1_LONE_CHUNK
#define MW_OR_P1_LONE_CHUNK 4  // points whose loads the lone frame's spectrum workgroup requests together (1 / 2 / 4 / 8: 1024^2 34.4 / 34.1 / 34.0 /
#endif                         // 34.7 us per frame, 512^2 19.9 / 19.0 / 18.9 / 18.9; the three-transform plan keeps point by point)
#ifndef MW_OR_P2_LONE_SPLIT
#define MW_OR_P2_LONE_SPLIT 1  // lone frame, packed plan: pass 2 with one field per workgroup
#endif
#ifndef MW_OR_PACKED_MAX_M
#define MW_OR_PACKED_MAX_M 4096  // textures from this size up keep the three-transform plan
#endif
#ifndef MW_OR_PACKED
#define MW_OR_PACKED 1  // planar-texture calls with a mirror-symmetric phase run two transforms per frame (0: always three)
#endif

// the packed plan's spectrum kernel (lone frame: gridDim.y == 2, one field per workgroup; tiles / big textures: gridDim.y == 1, both fields)
template <int N, int P>
__global__ __launch_bounds__((OrP1Geom<N, P>::NTHREADS)) void k_or_pass1_packed(OrP1Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = OrP1Geom<N, P>;
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int T = FftGeom<N, P>::T;
    const int tid = threadIdx.x, jb = blockIdx.x;
    const int f0 = gridDim.y == 1 ? 0 : (int)blockIdx.y, f1 = gridDim.y == 1 ? 2 : f0 + 1;
    {   // tile blockIdx.z of a batched handle
        const size_t toff = (size_t)blockIdx.z * N * N;
        A.initT += toff; A.PQT += toff; A.phase_in += toff; A.phase_out += toff; A.E += 2 * toff;
    }
    const int w = tid / T, u = tid % T;
    TwStage<N, P, G::NTHREADS, false> tws;
    tws.load(lds, A.TW, tid);
    const Twiddles tw = TwGeom<N, P>::view(A.TW, lds);
    cf* set0 = lds + G::TW_LDS;
    cf h[P], hh[P], x[P];
#pragma unroll
    for (int q = 0; q < P; q++) h[q] = hh[q] = mk(0.f, 0.f);
    if (gridDim.y == 1) or_p1_animate_packed<N, P, MW_OR_P1_CHUNK>(A, jb, tid, true, true, true, h, hh);
    else or_p1_animate_packed<N, P, MW_OR_P1_LONE_CHUNK>(A, jb, tid, f0 == 0, f0 == 0, f0 == 1, h, hh);
    tws.store(lds, tid);
    for (int f = f0; f < f1; f++) {
        or_p1_build_packed<N, P>(A, jb, tid, f, h, hh, x);
        if (f != f0) __syncthreads();
        stage0_store<N, P, -1>(x, u, set0 + w * G::BUFSTRIDE);
        __syncthreads();
#pragma unroll
        for (int s = 1; s < FftGeom<N, P>::S; s++) {
            load_slots<N, P>(x, u, set0 + w * G::BUFSTRIDE, s - 1);
            __syncthreads();
            stage_store<N, P, -1, false>(x, u, set0 + w * G::BUFSTRIDE, tw, s);
            __syncthreads();
        }
        or_p1_finish<N, P>(A, tw, jb, tid, f, x, set0);
    }
}

template <int N, int P>
__global__ __launch_bounds__((OrP1Geom<N, P>::NTHREADS)) void k_or_pass1(OrP1Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = OrP1Geom<N, P>;
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int T = FftGeom<N, P>::T;
    // gridDim.y == 3: one field per block (a single texture is latency-bound: the three fields of a column job run as three
    // concurrent blocks, each recomputing the cheap h~); gridDim.y == 1: the block does the three fields one after the other
    // from ONE read of the spectrum and phase (a batched handle is bandwidth-bound: 24 instead of 41.5 B per texel read)
    const int tid = threadIdx.x, jb = blockIdx.x;
    const int f0 = gridDim.y == 1 ? 0 : (int)blockIdx.y, f1 = gridDim.y == 1 ? 3 : f0 + 1;
    {   // tile blockIdx.z of a batched handle
        const size_t toff = (size_t)blockIdx.z * N * N;
        A.initT += toff; A.phase_in += toff; A.phase_out += toff; A.E += 3 * toff;
    }
    const int w = tid / T, u = tid % T;
    TwStage<N, P, G::NTHREADS, false> tws;
    tws.load(lds, A.TW, tid);
    const Twiddles tw = TwGeom<N, P>::view(A.TW, lds);
    cf* set0 = lds + G::TW_LDS;
    cf h[P], x[P];
    if (gridDim.y == 1) or_p1_animate<N, P, MW_OR_P1_CHUNK>(A, jb, tid, f0 == 0, h);  // one workgroup per column job: bandwidth-bound forms
    else or_p1_animate<N, P, 1>(A, jb, tid, f0 == 0, h);                                // one field per workgroup: the lone frame
    tws.store(lds, tid);  // behind the spectrum / phase requests; published by the first barrier
    for (int f = f0; f < f1; f++) {
        or_p1_build<N, P>(A, jb, tid, f, h, x);
        if (f != f0) __syncthreads();  // the previous field's final-pass reads of the buffers are done
        stage0_store<N, P, -1>(x, u, set0 + w * G::BUFSTRIDE);
        __syncthreads();
#pragma unroll
        for (int s = 1; s < FftGeom<N, P>::S; s++) {
            load_slots<N, P>(x, u, set0 + w * G::BUFSTRIDE, s - 1);
            __syncthreads();
            stage_store<N, P, -1, false>(x, u, set0 + w * G::BUFSTRIDE, tw, s);
            __syncthreads();
        }
        or_p1_finish<N, P>(A, tw, jb, tid, f, x, set0);
    }
}

// nframes consecutive frames of one ocean: grid (M/4 column jobs, 1, frame groups).  KEEP: the initial spectrum of the workgroup's
// points stays in registers over the frames of its group (else re-read per frame: an L2 hit after the first).
// (forcing 5 / 6 waves per SIMD with amdgpu_waves_per_eu: 44 / 108 B of scratch, 19.4 / 21.5 us per frame against 19.2: not kept)
template <int N, int P, bool KEEP, bool PACKED = false>
__global__ __launch_bounds__((OrP1Geom<N, P>::NTHREADS)) void k_or_pass1_steps(OrP1StepsArgs S) {
    constexpr int NF = PACKED ? 2 : 3;  // planes of the exchange buffer per frame
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = OrP1Geom<N, P>;
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int T = FftGeom<N, P>::T;
    const int tid = threadIdx.x, jb = blockIdx.x;
    const int k0 = (int)blockIdx.z * S.group, k1 = (k0 + S.group < S.nframes) ? k0 + S.group : S.nframes;
    OrP1Args A = S.a;
    const int w = tid / T, u = tid % T;
    TwStage<N, P, G::NTHREADS, false> tws;
    tws.load(lds, A.TW, tid);
    const Twiddles tw = TwGeom<N, P>::view(A.TW, lds);
    cf* set0 = lds + G::TW_LDS;
    float om[P], ph[P];
    f4 v[P], vn;
    const int fsel = PACKED ? (int)blockIdx.y : 0;  // packed plan: one field per workgroup (or_p1_steps_coeff)
    or_p1_steps_begin<N, P>(A, jb, tid, om, ph);
    float gain[PACKED ? P : 1];
    if constexpr (PACKED) {
        if (KEEP) or_p1_steps_coeff<N, P>(A, jb, tid, fsel, v, vn);
        or_p1_steps_gain<N, P>(A, jb, tid, fsel, gain);  // (the choppiness of this enqueue: A.c is fixed for its frames)
    }
    else { if (KEEP) or_p1_steps_spectrum<N, P>(A, jb, tid, v); }
    tws.store(lds, tid);  // behind the phase / spectrum requests; published by the first barrier
    for (int k = 0; k < k0; k++) or_p1_steps_advance<P>(om, ph, S.dt[k]);  // the chain over the frames of the groups before this one
    A.E += (size_t)NF * N * N * k0;
    for (int k = k0; k < k1; k++) {
        cf h[PACKED ? 1 : P], x[P];
        or_p1_steps_advance<P>(om, ph, S.dt[k]);
        if constexpr (PACKED) {
            if (!KEEP) or_p1_steps_coeff<N, P>(A, jb, tid, fsel, v, vn);
        } else {
            if (!KEEP) or_p1_steps_spectrum<N, P>(A, jb, tid, v);
            or_p1_steps_animate<P>(v, ph, h);
        }
        for (int f = PACKED ? fsel : 0; f < (PACKED ? fsel + 1 : 3); f++) {
            if constexpr (PACKED) or_p1_steps_build_split<N, P>(tid, f, gain, v, vn, ph, x);
            else or_p1_build<N, P>(A, jb, tid, f, h, x);
            if ((!PACKED && f != 0) || k != k0) __syncthreads();  // the previous field's final-pass reads of the buffers are done
            stage0_store<N, P, -1>(x, u, set0 + w * G::BUFSTRIDE);
            __syncthreads();
#pragma unroll
            for (int s = 1; s < FftGeom<N, P>::S; s++) {
                load_slots<N, P>(x, u, set0 + w * G::BUFSTRIDE, s - 1);
                __syncthreads();
                stage_store<N, P, -1, false>(x, u, set0 + w * G::BUFSTRIDE, tw, s);
                __syncthreads();
            }
            or_p1_finish<N, P>(A, tw, jb, tid, f, x, set0);
        }
        A.E += (size_t)NF * N * N;
    }
    if (k1 == S.nframes && fsel == 0) or_p1_steps_store_phase<N, P>(A, jb, tid, ph);  // the last group holds the phase after every frame
}

// the textures of frame `last` of a steps call become the handle's latest frame (mw_ocean_displace_mesh, the tile gather, ...): every array as
// float4 (M * M is a multiple of 4 and the frame offsets keep the 16-byte alignment), thread i copies element i of each array that has one
__global__ __launch_bounds__(256) void k_or_copy_frame(size_t MM, const float* h, const cf* d, const float* dg, const float* n, const float* w,
                                                       const float* hg, const float* da, float* oh, cf* od, float* odg, float* on, float* ow,
                                                       float* ohg, float* oda) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, q = MM / 4;
    auto cp = [i](const void* src, void* dst) { reinterpret_cast<f4*>(dst)[i] = reinterpret_cast<const f4*>(src)[i]; };
    if (i < 3 * q) cp(n, on);
    if (i < 2 * q) cp(d, od);
    if (i < q) {
        cp(h, oh); cp(dg, odg); cp(w, ow);
        if (hg) { cp(hg, ohg); cp(da, oda); }
    }
}

// workgroup b of nb (a multiple of 8, else the identity) -> index in an order that gives XCD b % 8 one contiguous eighth of the indices
__device__ __forceinline__ unsigned or_xcd_band(unsigned b, unsigned nb) { return (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b; }
template <int N, int P>
__global__ __launch_bounds__((OrP2Geom<N, P>::NTHREADS)) void k_or_pass2(OrP2Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = OrP2Geom<N, P>;
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int T = FftGeom<N, P>::T;
    // (row groups banded per XCD like the normal pass's rows, so that it would find these textures in its own L2: measured no gain, round 6)
    const int tid = threadIdx.x, ab = blockIdx.x;
    {   // tile blockIdx.z of a batched handle
        const size_t toff = (size_t)blockIdx.z * N * N;
        A.E += 3 * toff; A.height += toff; A.disp += toff; A.disp_g += toff;
        if (A.height_g) A.height_g += toff;
        if (A.disp_a) A.disp_a += toff;
    }
    TwStage<N, P, G::NTHREADS, false> tws;
    tws.load(lds, A.TW, tid);
    const Twiddles tw = TwGeom<N, P>::view(A.TW, lds);
    cf* set0 = lds + G::TW_LDS;
    cf x[P];
    float dx[P];
    // grid.y = 0: hx then hz (their real parts leave interleaved as displacementTexture.rb); grid.y = 1: h
    const int k0 = blockIdx.y == 0 ? 0 : 2, k1 = blockIdx.y == 0 ? 2 : 3;
    for (int k = k0; k < k1; k++) {
        const int f = or_p2_field(k);
        if (k != k0) __syncthreads();
        or_p2_load<N, P>(A, ab, tid, f, x, set0);
        if (k == k0) tws.store(lds, tid);  // behind the first row requests; published by the barrier below
        __syncthreads();
#pragma unroll
        for (int s = 1; s < FftGeom<N, P>::S; s++) {
            // exact LDS layouts: a wave stays inside one row; padded layout: the load-side (row-interleaved) mapping
            const int um = XLay<N, P>::EXACT ? tid % FftGeom<N, P>::T : tid >> 2, rm = XLay<N, P>::EXACT ? tid / FftGeom<N, P>::T : tid & 3;
            load_slots<N, P>(x, um, set0 + rm * G::BUFSTRIDE, s - 1);
            __syncthreads();
            stage_store<N, P, -1, false>(x, um, set0 + rm * G::BUFSTRIDE, tw, s);
            __syncthreads();
        }
        or_p2_finish<N, P>(A, tw, ab, tid, f, x, dx, set0);
    }
}

// the packed plan's second pass: plane 0 = hx (-> displacement.r, .g), plane 1 = G (-> height.r + i displacement.b); one workgroup per 4 rows
template <int N, int P>
__global__ __launch_bounds__((OrP2Geom<N, P>::NTHREADS)) void k_or_pass2_packed(OrP2Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = OrP2Geom<N, P>;
    cf* lds = reinterpret_cast<cf*>(smem);
    const int tid = threadIdx.x, ab = blockIdx.x;
    {   // tile / frame blockIdx.z
        const size_t toff = (size_t)blockIdx.z * N * N;
        A.E += 2 * toff; A.height += toff; A.disp += toff; A.disp_g += toff;
    }
    TwStage<N, P, G::NTHREADS, false> tws;
    tws.load(lds, A.TW, tid);
    const Twiddles tw = TwGeom<N, P>::view(A.TW, lds);
    cf* set0 = lds + G::TW_LDS;
    cf x[P];
    float dx[P];
    // gridDim.y == 2 (the lone frame: 256 row groups on 256 CUs, latency-bound): one field per workgroup, half the work on the critical path
    const bool split = gridDim.y == 2;
    const int k0 = split ? (int)blockIdx.y : 0, k1 = split ? k0 + 1 : 2;
    for (int k = k0; k < k1; k++) {
        if (k != k0) __syncthreads();
        or_p2_load<N, P>(A, ab, tid, k, x, set0);
        if (k == k0) tws.store(lds, tid);
        __syncthreads();
#pragma unroll
        for (int s = 1; s < FftGeom<N, P>::S; s++) {
            const int um = XLay<N, P>::EXACT ? tid % FftGeom<N, P>::T : tid >> 2, rm = XLay<N, P>::EXACT ? tid / FftGeom<N, P>::T : tid & 3;
            load_slots<N, P>(x, um, set0 + rm * G::BUFSTRIDE, s - 1);
            __syncthreads();
            stage_store<N, P, -1, false>(x, um, set0 + rm * G::BUFSTRIDE, tw, s);
            __syncthreads();
        }
        or_p2_finish<N, P>(A, tw, ab, tid, split ? 4 + k : (k == 0 ? 1 : 3), x, dx, set0);
    }
}

// F/OceanNormal.shader + F/WhiteCap.shader in one launch: WhiteCap reads _Bump at its own texel only (:38), so the thread
// that produced the normal goes straight on to the whitecap (its own global write is visible to itself).
#ifndef MW_OR_NW_BANDS
#define MW_OR_NW_BANDS 1  // one band of texel rows per XCD (0: rows round-robin over the XCDs, A/B)
#endif
#ifndef MW_OR_NW_LDS_TURN
#define MW_OR_NW_LDS_TURN 1  // a wave's normals turned through LDS so that every store instruction writes 1 KiB contiguous (0: straight from the lane)
#endif
#ifndef MW_OR_NW_QUAD
#define MW_OR_NW_QUAD 1  // four texels of a row per thread from 16-byte loads (0: one texel per thread, A/B)
#endif
template <bool NT>
__global__ __launch_bounds__(256) void k_or_normal_white(OrConsts c, const float* height, const cf* disp, const float* disp_g,
                                                         float* normal, float* white) {
    // XCD-aware: block b runs on XCD b % 8; give each XCD one contiguous band of texel rows, so that the +-1 and +-8 row
    // neighbours are hits in ITS L2 (round-robin rows made every XCD fetch its own copy: 29.5 B/texel for 16 needed)
    const unsigned blk = MW_OR_NW_BANDS ? or_xcd_band(blockIdx.x, gridDim.x) : blockIdx.x;
    int idx = blk * blockDim.x + threadIdx.x;
    {   // tile blockIdx.y of a batched handle
        const size_t toff = (size_t)blockIdx.y * c.M * c.M;
        height += toff; disp += toff; disp_g += toff; normal += 3 * toff; white += toff;
    }
#if MW_OR_NW_QUAD
    const int Q = c.M / 4;  // thread idx: texels 4 (idx % Q) .. + 3 of row idx / Q
    if (idx >= c.M * Q) return;  // whole waves (M^2 / 4 is a multiple of 64)
    float n[4][3], w[4];
    or_normal_white_quad_compute(c, 4 * (idx % Q), idx / Q, height, disp, disp_g, n, w);
    f4 o;
    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
    mw_store_stream<NT>(reinterpret_cast<f4*>(white) + idx, o);  // 64 lanes x 16 B = 1 KiB contiguous
#if MW_OR_NW_LDS_TURN
    // The normals: a lane owns 48 contiguous bytes (4 texels x 3), a wave 3 KiB (thread idx <-> texels 4 idx .. 4 idx + 3 of the row-major texture).
    // Stored straight from the lane every instruction writes 16-byte pieces 48 bytes apart -- a third of every line, left to the L2 to merge, and
    // slow with the non-temporal hint.  Turned through LDS inside the wave (as wave_store_3f4 does for the pond's vertices) every store
    // instruction writes 64 consecutive float4 = 1 KiB.
    __shared__ f4 tile[256 * 3];
    const int lane = threadIdx.x & 63;
    f4* wt = tile + (threadIdx.x - lane) * 3;
    f4 r0, r1, r2;
    r0.x = n[0][0]; r0.y = n[0][1]; r0.z = n[0][2]; r0.w = n[1][0];
    r1.x = n[1][1]; r1.y = n[1][2]; r1.z = n[2][0]; r1.w = n[2][1];
    r2.x = n[2][2]; r2.y = n[3][0]; r2.z = n[3][1]; r2.w = n[3][2];
    wt[lane * 3] = r0; wt[lane * 3 + 1] = r1; wt[lane * 3 + 2] = r2;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f4* dst = reinterpret_cast<f4*>(normal) + (size_t)3 * (idx - lane);
#pragma unroll
    for (int j = 0; j < 3; j++) mw_store_stream<NT>(&dst[j * 64 + lane], wt[j * 64 + lane]);
#else
    float* np_ = normal + (size_t)12 * idx;
    o.x = n[0][0]; o.y = n[0][1]; o.z = n[0][2]; o.w = n[1][0]; mw_store_stream<NT>(reinterpret_cast<f4*>(np_), o);
    o.x = n[1][1]; o.y = n[1][2]; o.z = n[2][0]; o.w = n[2][1]; mw_store_stream<NT>(reinterpret_cast<f4*>(np_ + 4), o);
    o.x = n[2][2]; o.y = n[3][0]; o.z = n[3][1]; o.w = n[3][2]; mw_store_stream<NT>(reinterpret_cast<f4*>(np_ + 8), o);
#endif
#else
    if (idx >= c.M * c.M) return;
    float nxz[2];
    or_normal_element<NT>(c, idx % c.M, idx / c.M, height, disp, disp_g, normal, nxz);
    or_white_element<NT>(c, idx % c.M, idx / c.M, disp, normal, white, nxz);
#endif
}
static inline unsigned or_nw_blocks(size_t MM) { return (unsigned)((MM / (MW_OR_NW_QUAD ? 4 : 1) + 255) / 256); }

__global__ void k_or_pack_rgba(int M, const float* height, const float* height_g, const cf* disp, const float* disp_g,
                               const float* disp_a, const float* normal, const float* white, f4* H, f4* D, f4* Nn, f4* W) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * M) return;
    const size_t toff = (size_t)blockIdx.y * M * M;  // tile blockIdx.y
    or_pack_rgba_element(idx, height + toff, height_g + toff, disp + toff, disp_g + toff, disp_a + toff, normal + 3 * toff, white + toff,
                         H ? H + toff : nullptr, D ? D + toff : nullptr, Nn ? Nn + toff : nullptr, W ? W + toff : nullptr);
}
__global__ void k_or_displace_mesh(int M, int res, float unit_width, const float* height, const cf* disp, const float* normal,
                                   const float* white, float* vert, float* nrm, float* col) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= res * res) return;
    const size_t toff = (size_t)blockIdx.y * M * M, voff = (size_t)blockIdx.y * res * res;  // tile blockIdx.y
    or_mesh_vertex(M, res, unit_width, idx / res, idx % res, height + toff, disp + toff, normal + 3 * toff, white + toff, vert + 3 * voff,
                   nrm ? nrm + 3 * voff : nullptr, col ? col + voff : nullptr);
}

std::vector<cf> build_twiddle_table(int N, int P, int sgn);  // mistral_water.hip
int plan_points_host(int N);

static inline void or_free(OrState& s) {
    hipFree(s.initT); hipFree(s.PQT); hipFree(s.phaseT); hipFree(s.phaseT2); hipFree(s.omT); hipFree(s.TW); hipFree(s.E); hipFree(s.out_height); hipFree(s.out_disp_cf);
    hipFree(s.out_disp_g); hipFree(s.out_normal); hipFree(s.out_white); hipFree(s.out_height_g); hipFree(s.out_disp_a);
    hipFree(s.fr_E); hipFree(s.fr_height); hipFree(s.fr_disp_g); hipFree(s.fr_normal); hipFree(s.fr_white); hipFree(s.fr_height_g);
    hipFree(s.fr_disp_a); hipFree(s.fr_disp);
    s = OrState();
}

static inline mw_status or_create(OrState& s, const mw_params& p, int M, hipStream_t st, int tiles = 1) {
    s.M = M;
    s.tiles = tiles;
    s.c.M = M; s.c.length = p.length; s.c.gravity = p.gravity; s.c.choppiness = p.choppiness; s.c.normal_length = p.length;
    s.mult = p.mult; s.choppiness = p.choppiness;
    const size_t MM = (size_t)M * M, TM = MM * (size_t)tiles;
    std::vector<cf> tab = build_twiddle_table(M, plan_points_host(M), -1);
#define OR_ALLOC(ptr, bytes) if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) { g_or_err = "OceanRenderer: hipMalloc failed"; return MW_ENOMEM; }
    OR_ALLOC(s.initT, sizeof(f4) * TM) OR_ALLOC(s.PQT, sizeof(f4) * TM) OR_ALLOC(s.phaseT, sizeof(float) * TM) OR_ALLOC(s.phaseT2, sizeof(float) * TM) OR_ALLOC(s.omT, sizeof(float) * MM) OR_ALLOC(s.TW, sizeof(cf) * tab.size())
    OR_ALLOC(s.E, sizeof(cf) * 3 * TM) OR_ALLOC(s.out_height, sizeof(float) * TM) OR_ALLOC(s.out_disp_cf, sizeof(cf) * TM)
    OR_ALLOC(s.out_disp_g, sizeof(float) * TM) OR_ALLOC(s.out_normal, sizeof(float) * 3 * TM) OR_ALLOC(s.out_white, sizeof(float) * TM)
#undef OR_ALLOC
    s.out_disp = reinterpret_cast<float*>(s.out_disp_cf);
    if (hipMemcpy(s.TW, tab.data(), sizeof(cf) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) { g_or_err = "twiddle upload failed"; return MW_EDEVICE; }
    k_or_init<<<dim3((unsigned)((MM + 255) / 256), tiles), dim3(256), 0, st>>>(M, p.length, p.wind_x, p.wind_y, p.amplitude / 10000.f,
                                                                              p.gravity, p.seed, s.initT, s.phaseT);
    k_or_omega<<<dim3((unsigned)((MM + 255) / 256)), dim3(256), 0, st>>>(s.c, s.omT);
    k_or_prep<<<dim3((unsigned)((MM + 255) / 256), tiles), dim3(256), 0, st>>>(M, s.initT, s.PQT);
    if (hipGetLastError() != hipSuccess) { g_or_err = "k_or_init launch failed"; return MW_EDEVICE; }
    return MW_OK;
}
// the planar-texture plan of a call: two transforms per frame where the identity holds (or_prep_element), else the shaders' three
// (4096^2 textures keep three: their 1024-thread P = 16 workgroups have 128 registers per lane, and two animated spectra at once spill --
// one GenerateTexture() 545 -> 637 us; 2048^2: 127 -> 110 us, 4 / 8 tiles per call 24.0 / 25.1 -> 24.3 / 23.3 us per tile-frame)
static inline bool or_use_packed(const OrState& s) {
    return MW_OR_PACKED != 0 && sw(SW_OR_PACKED) != 0 && !s.want_imag && s.phase_sym && s.M < MW_OR_PACKED_MAX_M;
}

// RenderInitial() after a parameter change (S/OceanRenderer.cs:98-109): initialTexture again from the new length / wind /
// amplitude, dispersion and spectrum passes on the new length (:94-97); the phase textures and the normal pass's length stay
static inline mw_status or_reinit(OrState& s, float length, float wind_x, float wind_y, float amplitude, uint64_t seed, hipStream_t st) {
    const size_t MM = (size_t)s.M * s.M;
    s.c.length = length;
    k_or_init<<<dim3((unsigned)((MM + 255) / 256), s.tiles), dim3(256), 0, st>>>(s.M, length, wind_x, wind_y, amplitude / 10000.f,
                                                                                s.c.gravity, seed, s.initT, nullptr);
    k_or_omega<<<dim3((unsigned)((MM + 255) / 256)), dim3(256), 0, st>>>(s.c, s.omT);
    k_or_prep<<<dim3((unsigned)((MM + 255) / 256), s.tiles), dim3(256), 0, st>>>(s.M, s.initT, s.PQT);
    if (hipGetLastError() != hipSuccess) { g_or_err = "k_or_init launch failed"; return MW_EDEVICE; }
    return MW_OK;
}
// phase texture <-> host order (texel (px,py) at py*M + px); the device keeps it transposed
__global__ void k_or_phase_transpose(int M, const float* src, float* dst) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * M) return;
    const size_t toff = (size_t)blockIdx.y * M * M;  // tile blockIdx.y
    (dst + toff)[idx] = (src + toff)[(size_t)(idx % M) * M + idx / M];
}

#ifndef MW_OR_STREAM_E_TILES
#define MW_OR_STREAM_E_TILES 2
#endif
#ifndef MW_OR_BIG_N
#define MW_OR_BIG_N 2048  // from this texture size up ONE tile is what several 1024^2 tiles are: it fills the device and its exchange buffer the caches
#endif
// a call is "bandwidth-bound" (three fields per pass-1 block from one read of the spectrum, streaming stores) when it carries several tiles or
// one big one; a single 1024^2 texture is latency-bound (one field per block, cacheable stores)
#ifndef MW_OR_STREAM_BIG_N
#define MW_OR_STREAM_BIG_N MW_OR_BIG_N
#endif
template <int N>
static inline bool or_call_is_big(const OrState& s) { return s.tiles >= MW_OR_STREAM_E_TILES || N >= MW_OR_STREAM_BIG_N; }
template <int N>
static hipError_t or_launch_passes(OrState& s, float dt, hipStream_t st, hipEvent_t* ev = nullptr) {
    constexpr int P = Plan<N>::P;
    static AttrOnce attr1, attr2;
    {
        hipError_t e = attr1.set(reinterpret_cast<const void*>(&k_or_pass1<N, P>), OrP1Geom<N, P>::LDS_BYTES);
        if (e != hipSuccess) return e;
        e = attr2.set(reinterpret_cast<const void*>(&k_or_pass2<N, P>), OrP2Geom<N, P>::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    OrP1Args A1;
    A1.initT = s.initT; A1.PQT = s.PQT; A1.omT = s.omT; A1.phase_in = s.phaseT; A1.phase_out = s.phaseT2; A1.TW = s.TW; A1.E = s.E; A1.c = s.c; A1.dt = dt;
    A1.stream_E = or_call_is_big<N>(s) ? 1 : 0;
    constexpr int NT1 = OrP1Geom<N, P>::NTHREADS, LB1 = OrP1Geom<N, P>::LDS_BYTES;
    constexpr int NT2 = OrP2Geom<N, P>::NTHREADS, LB2 = OrP2Geom<N, P>::LDS_BYTES;
    const bool packed = or_use_packed(s), all_fields = (s.tiles > 1 || N >= MW_OR_BIG_N);
    OrP2Args A2;
    A2.E = s.E; A2.TW = s.TW; A2.height = s.out_height; A2.disp = s.out_disp_cf; A2.disp_g = s.out_disp_g; A2.c = s.c;
    A2.height_g = s.want_imag ? s.out_height_g : nullptr;
    A2.disp_a = s.want_imag ? s.out_disp_a : nullptr;
    if constexpr (N < MW_OR_PACKED_MAX_M) if (packed) {
        static AttrOnce attr1p, attr2p;
        hipError_t e = attr1p.set(reinterpret_cast<const void*>(&k_or_pass1_packed<N, P>), LB1);
        if (e != hipSuccess) return e;
        e = attr2p.set(reinterpret_cast<const void*>(&k_or_pass2_packed<N, P>), LB2);
        if (e != hipSuccess) return e;
        if (ev) hipEventRecord(ev[0], st);
        k_or_pass1_packed<N, P><<<dim3(N / 4, all_fields ? 1 : 2, s.tiles), dim3(NT1), LB1, st>>>(A1);
        if (ev) hipEventRecord(ev[1], st);
        std::swap(s.phaseT, s.phaseT2);
        k_or_pass2_packed<N, P><<<dim3(N / 4, (all_fields || !MW_OR_P2_LONE_SPLIT) ? 1 : 2, s.tiles), dim3(NT2), LB2, st>>>(A2);
        if (ev) hipEventRecord(ev[2], st);
        return hipGetLastError();
    }
    if (ev) hipEventRecord(ev[0], st);
    k_or_pass1<N, P><<<dim3(N / 4, all_fields ? 1 : 3, s.tiles), dim3(NT1), LB1, st>>>(A1);
    if (ev) hipEventRecord(ev[1], st);
    std::swap(s.phaseT, s.phaseT2);
    k_or_pass2<N, P><<<dim3(N / 4, 2, s.tiles), dim3(NT2), LB2, st>>>(A2);
    if (ev) hipEventRecord(ev[2], st);
    return hipGetLastError();
}

// one GenerateTexture(): results land in s.out_*; optional device destinations receive copies
// ev (measurement hook, 5 events): recorded before pass 1 and after pass 1, pass 2, the normal / whitecap pass and the copies
static inline mw_status or_generate(OrState& s, float delta_time, float* d_height, float* d_disp, float* d_normal, float* d_white,
                                    hipStream_t st, hipEvent_t* ev = nullptr) {
    s.c.choppiness = s.choppiness;
    const float dt = delta_time * s.mult;  // S/OceanRenderer.cs:223
    if (s.want_imag && !s.out_height_g) {
        const size_t bytes = sizeof(float) * (size_t)s.M * s.M * s.tiles;
        if (hipMalloc((void**)&s.out_height_g, bytes) != hipSuccess || hipMalloc((void**)&s.out_disp_a, bytes) != hipSuccess) {
            g_or_err = "OceanRenderer: hipMalloc failed";
            return MW_ENOMEM;
        }
    }
    hipError_t e = hipSuccess;
    switch (s.M) {
        case 64: e = or_launch_passes<64>(s, dt, st, ev); break;
        case 128: e = or_launch_passes<128>(s, dt, st, ev); break;
        case 256: e = or_launch_passes<256>(s, dt, st, ev); break;
        case 512: e = or_launch_passes<512>(s, dt, st, ev); break;
        case 1024: e = or_launch_passes<1024>(s, dt, st, ev); break;
        case 2048: e = or_launch_passes<2048>(s, dt, st, ev); break;
        case 4096: e = or_launch_passes<4096>(s, dt, st, ev); break;
        default: g_or_err = "OceanRenderer: unsupported texture size"; return MW_EINVAL;
    }
    if (e != hipSuccess) { g_or_err = std::string("OceanRenderer pass launch: ") + hipGetErrorString(e); return MW_EDEVICE; }
    const size_t MM = (size_t)s.M * s.M, TM = MM * (size_t)s.tiles;
    const unsigned nb = or_nw_blocks(MM);
    if (s.tiles >= MW_OR_STREAM_E_TILES || s.M >= MW_OR_STREAM_BIG_N)
        k_or_normal_white<true><<<dim3(nb, s.tiles), dim3(256), 0, st>>>(s.c, s.out_height, s.out_disp_cf, s.out_disp_g, s.out_normal, s.out_white);
    else
        k_or_normal_white<false><<<dim3(nb, s.tiles), dim3(256), 0, st>>>(s.c, s.out_height, s.out_disp_cf, s.out_disp_g, s.out_normal, s.out_white);
    if (hipGetLastError() != hipSuccess) { g_or_err = "OceanRenderer normal/white launch failed"; return MW_EDEVICE; }
    if (ev) hipEventRecord(ev[3], st);
    s.have_frame = true;
    s.have_imag = s.want_imag;
    hipError_t ce = hipSuccess;
    if (d_height && ce == hipSuccess) ce = hipMemcpyAsync(d_height, s.out_height, TM * 4, hipMemcpyDeviceToDevice, st);
    if (d_disp && ce == hipSuccess) ce = hipMemcpyAsync(d_disp, s.out_disp_cf, TM * 8, hipMemcpyDeviceToDevice, st);
    if (d_normal && ce == hipSuccess) ce = hipMemcpyAsync(d_normal, s.out_normal, TM * 12, hipMemcpyDeviceToDevice, st);
    if (d_white && ce == hipSuccess) ce = hipMemcpyAsync(d_white, s.out_white, TM * 4, hipMemcpyDeviceToDevice, st);
    if (ce != hipSuccess) { g_or_err = std::string("OceanRenderer result copy: ") + hipGetErrorString(ce); return MW_EDEVICE; }
    if (ev) hipEventRecord(ev[4], st);
    return MW_OK;
}

// one GenerateTexture() delivered as the reference's four ARGBFloat render targets (any destination may be NULL)
static inline mw_status or_generate_rgba(OrState& s, float delta_time, f4* d_height, f4* d_disp, f4* d_normal, f4* d_white,
                                         hipStream_t st) {
    s.want_imag = true;  // for this call: Im h and Im Dz are channels of the targets, so it runs the three-transform plan (or_use_packed)
    mw_status r = or_generate(s, delta_time, nullptr, nullptr, nullptr, nullptr, st);
    s.want_imag = false;
    if (r != MW_OK) return r;
    const size_t MM = (size_t)s.M * s.M;
    k_or_pack_rgba<<<dim3((unsigned)((MM + 255) / 256), s.tiles), dim3(256), 0, st>>>(s.M, s.out_height, s.out_height_g, s.out_disp_cf,
                                                                              s.out_disp_g, s.out_disp_a, s.out_normal,
                                                                              s.out_white, d_height, d_disp, d_normal, d_white);
    if (hipGetLastError() != hipSuccess) { g_or_err = "k_or_pack_rgba launch failed"; return MW_EDEVICE; }
    return MW_OK;
}

// ---- nframes consecutive GenerateTexture() calls in one enqueue ------------------------------------------------------------------
// frame buffers of the handle: the exchange buffer and disp.g for `n` frames, plus every texture the caller gave no destination for
static inline mw_status or_frames_reserve(OrState& s, int n, const bool (&need)[4], bool imag) {
    const size_t MM = (size_t)s.M * s.M;
    if (n > s.frames_cap) {  // grow: drop everything (hipFree waits for the device), the arrays come back below at the new size
        hipFree(s.fr_E); hipFree(s.fr_height); hipFree(s.fr_disp_g); hipFree(s.fr_normal); hipFree(s.fr_white); hipFree(s.fr_height_g);
        hipFree(s.fr_disp_a); hipFree(s.fr_disp);
        s.fr_E = nullptr; s.fr_disp = nullptr;
        s.fr_height = s.fr_disp_g = s.fr_normal = s.fr_white = s.fr_height_g = s.fr_disp_a = nullptr;
        s.frames_cap = 0; s.frames_last = 0;
    }
    const size_t cap = (size_t)(s.frames_cap ? s.frames_cap : n);
#define OR_FR(ptr, cond, bytes) if ((cond) && !(ptr) && hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) { (void)hipGetLastError(); g_or_err = "OceanRenderer: hipMalloc of the frame buffers failed"; return MW_ENOMEM; }
    OR_FR(s.fr_E, true, sizeof(cf) * 3 * MM * cap) OR_FR(s.fr_disp_g, true, sizeof(float) * MM * cap)
    OR_FR(s.fr_height, need[0], sizeof(float) * MM * cap) OR_FR(s.fr_disp, need[1], sizeof(cf) * MM * cap)
    OR_FR(s.fr_normal, need[2], sizeof(float) * 3 * MM * cap) OR_FR(s.fr_white, need[3], sizeof(float) * MM * cap)
    OR_FR(s.fr_height_g, imag, sizeof(float) * MM * cap) OR_FR(s.fr_disp_a, imag, sizeof(float) * MM * cap)
#undef OR_FR
    s.frames_cap = (int)cap;
    return MW_OK;
}
#ifndef MW_OR_FRAME_GROUPS
#define MW_OR_FRAME_GROUPS 4  // frame groups of a steps call = workgroups per column job (see k_or_pass1_steps)
#endif
#ifndef MW_OR_FRAME_GROUPS_PACKED
#define MW_OR_FRAME_GROUPS_PACKED 2  // the packed plan has two workgroups per column job and group already (one per field): 1 / 2 / 3 / 4 groups measured
#endif                               // 18.9 / 18.6 / 19.8 / 19.3 us per frame at 32 frames per enqueue
#ifndef MW_OR_STEPS_KEEP
#define MW_OR_STEPS_KEEP 1
#endif
#ifndef MW_OR_STEPS_CHUNK
#define MW_OR_STEPS_CHUNK 8  // frames per pass-2 / normal-pass launch pair at 1024^2 (scaled with the texture area)
#endif
static inline int or_steps_chunk(int M) {
    const long long c = (long long)MW_OR_STEPS_CHUNK * 1024 * 1024 / ((long long)M * M);
    return c < 1 ? 1 : (c > MW_OR_MAX_FRAMES ? MW_OR_MAX_FRAMES : (int)c);
}
static inline int or_steps_chunks(int M, int n) { const int c = or_steps_chunk(M); return (n + c - 1) / c; }
#ifndef MW_OR_STEPS_NW_NT
#define MW_OR_STEPS_NW_NT 1  // normal / whitecap textures of a steps call leave with non-temporal stores
#endif
#ifndef MW_OR_STEPS_MAX_N
#define MW_OR_STEPS_MAX_N 2048  // above: the 1024-thread P = 16 workgroup has 128 VGPRs per lane, no room for a chain in registers
#endif
#ifndef MW_OR_STEPS_KEEP_PACKED
#define MW_OR_STEPS_KEEP_PACKED 1  // packed plan: (h0, h0c) AND (P, Q) of a workgroup's points stay in registers over its frames
#endif
template <int N, bool PACKED>
static hipError_t or_launch_steps(OrState& s, const float* dt, int n, const OrP2Args& A2, float* f_n, float* f_w, hipStream_t st, hipEvent_t* ev = nullptr) {
    constexpr int P = Plan<N>::P, NF = PACKED ? 2 : 3;
    constexpr bool KEEP = (PACKED ? MW_OR_STEPS_KEEP_PACKED : MW_OR_STEPS_KEEP) != 0 && P <= 8;  // P = 16: no registers to spare
    static AttrOnce attr1, attr2;
    {
        hipError_t e;
        if constexpr (PACKED) e = attr2.set(reinterpret_cast<const void*>(&k_or_pass2_packed<N, P>), OrP2Geom<N, P>::LDS_BYTES);
        else e = attr2.set(reinterpret_cast<const void*>(&k_or_pass2<N, P>), OrP2Geom<N, P>::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    constexpr int NT1 = OrP1Geom<N, P>::NTHREADS, LB1 = OrP1Geom<N, P>::LDS_BYTES;
    if (ev) hipEventRecord(ev[0], st);
    constexpr int NT2 = OrP2Geom<N, P>::NTHREADS, LB2 = OrP2Geom<N, P>::LDS_BYTES;
    const size_t MM = (size_t)N * N;
    const int chunk = or_steps_chunk(N);
    // pass 2 and the normal / whitecap pass of the frames [c0, c0 + cn)
    auto rest = [&](int c0, int cn, int j) {
        OrP2Args B2 = A2;
        const size_t off = MM * (size_t)c0;
        B2.E += NF * off; B2.height += off; B2.disp += off; B2.disp_g += off;
        if (B2.height_g) B2.height_g += off;
        if (B2.disp_a) B2.disp_a += off;
        if constexpr (PACKED) k_or_pass2_packed<N, P><<<dim3(N / 4, 1, cn), dim3(NT2), LB2, st>>>(B2);
        else k_or_pass2<N, P><<<dim3(N / 4, 2, cn), dim3(NT2), LB2, st>>>(B2);
        if (ev) hipEventRecord(ev[2 + 3 * j], st);
        k_or_normal_white<MW_OR_STEPS_NW_NT != 0><<<dim3(or_nw_blocks(MM), cn), dim3(256), 0, st>>>(s.c, B2.height, B2.disp, B2.disp_g, f_n + 3 * off, f_w + off);
        if (ev) hipEventRecord(ev[3 + 3 * j], st);
    };
    // Pass 2 and the normal / whitecap pass alternate over chunks of frames: the height / displacement textures a chunk writes (16 B per texel
    // and frame) are still in the 256-MB Infinity Cache when the normal pass reads them back.  Behind 32 frames of pass-2 output (512 MB at
    // 1024^2) every read of the normal pass went to HBM: 9.1 us per frame against 7.2 from the cache (chunks of 4 / 8 / 16 / 32 frames:
    // 21.9 / 21.8 / 23.2 / 23.3 us per frame).
    if constexpr (N <= MW_OR_STEPS_MAX_N) {
        hipError_t e = attr1.set(reinterpret_cast<const void*>(&k_or_pass1_steps<N, P, KEEP, PACKED>), LB1);
        if (e != hipSuccess) return e;
        OrP1StepsArgs S;
        S.a.initT = s.initT; S.a.PQT = s.PQT; S.a.omT = s.omT; S.a.phase_in = s.phaseT; S.a.phase_out = s.phaseT2; S.a.TW = s.TW; S.a.E = s.fr_E; S.a.c = s.c;
        S.a.dt = 0.f; S.a.stream_E = 1;
        for (int k = 0; k < MW_OR_MAX_FRAMES; k++) S.dt[k] = k < n ? dt[k] : 0.f;
        S.nframes = n;
        // ONE spectrum launch over all frames: per chunk (so that pass 2 would find the exchange buffer in the cache) it ran 195 -> 283 us per 32
        // frames and pass 2 no faster -- the exchange buffer is written with streaming stores and does not stay
        int groups = PACKED ? MW_OR_FRAME_GROUPS_PACKED : MW_OR_FRAME_GROUPS;
        if (groups > n) groups = n;
        S.group = (n + groups - 1) / groups;
        groups = (n + S.group - 1) / S.group;
        k_or_pass1_steps<N, P, KEEP, PACKED><<<dim3(N / 4, PACKED ? 2 : 1, groups), dim3(NT1), LB1, st>>>(S);
        for (int c0 = 0, j = 0; c0 < n; c0 += chunk, j++) {
            if (ev) hipEventRecord(ev[1 + 3 * j], st);
            rest(c0, (n - c0 < chunk) ? n - c0 : chunk, j);
        }
        std::swap(s.phaseT, s.phaseT2);
    } else {  // one bandwidth-bound spectrum launch per frame (each already fills the device)
        hipError_t e;
        if constexpr (PACKED) e = attr1.set(reinterpret_cast<const void*>(&k_or_pass1_packed<N, P>), LB1);
        else e = attr1.set(reinterpret_cast<const void*>(&k_or_pass1<N, P>), LB1);
        if (e != hipSuccess) return e;
        for (int c0 = 0, j = 0; c0 < n; c0 += chunk, j++) {
            const int cn = (n - c0 < chunk) ? n - c0 : chunk;
            for (int k = c0; k < c0 + cn; k++) {
                OrP1Args A1;
                A1.initT = s.initT; A1.PQT = s.PQT; A1.omT = s.omT; A1.phase_in = s.phaseT; A1.phase_out = s.phaseT2; A1.TW = s.TW; A1.c = s.c; A1.dt = dt[k];
                A1.E = s.fr_E + (size_t)NF * N * N * k;
                A1.stream_E = 1;
                if constexpr (PACKED) k_or_pass1_packed<N, P><<<dim3(N / 4, 1, 1), dim3(NT1), LB1, st>>>(A1);
                else k_or_pass1<N, P><<<dim3(N / 4, 1, 1), dim3(NT1), LB1, st>>>(A1);
                std::swap(s.phaseT, s.phaseT2);
            }
            if (ev) hipEventRecord(ev[1 + 3 * j], st);
            rest(c0, cn, j);
        }
    }
    return hipGetLastError();
}
template <int N>
static hipError_t or_launch_steps_plan(bool packed, OrState& s, const float* dt, int n, const OrP2Args& A2, float* f_n, float* f_w, hipStream_t st, hipEvent_t* ev) {
    if constexpr (N < MW_OR_PACKED_MAX_M) { if (packed) return or_launch_steps<N, true>(s, dt, n, A2, f_n, f_w, st, ev); }
    return or_launch_steps<N, false>(s, dt, n, A2, f_n, f_w, st, ev);
}
// frames k = 0 .. n-1 advance the phase by delta_time[k] * mult one after the other, exactly as n calls of or_generate would; device
// destinations are [n][M*M*...] (NULL: the frame stays in the handle's own frame buffers).  The handle's latest-frame textures
// (out_*) receive frame n-1.
static inline mw_status or_generate_steps(OrState& s, const float* delta_time, int n, float* d_height, float* d_disp, float* d_normal,
                                          float* d_white, hipStream_t st, hipEvent_t* ev = nullptr) {
    if (s.tiles != 1) { g_or_err = "generate_texture_steps: a batched handle (mw_ocean_create_batch) advances one frame per call"; return MW_ESTATE; }
    if (n < 1 || n > MW_OR_MAX_FRAMES) { g_or_err = "generate_texture_steps: nframes out of range"; return MW_EINVAL; }
    s.c.choppiness = s.choppiness;
    const size_t MM = (size_t)s.M * s.M;
    if (s.want_imag && !s.out_height_g) {
        if (hipMalloc((void**)&s.out_height_g, sizeof(float) * MM) != hipSuccess || hipMalloc((void**)&s.out_disp_a, sizeof(float) * MM) != hipSuccess) {
            g_or_err = "OceanRenderer: hipMalloc failed";
            return MW_ENOMEM;
        }
    }
    const bool need[4] = {!d_height, !d_disp, !d_normal, !d_white};
    mw_status r = or_frames_reserve(s, n, need, s.want_imag);
    if (r != MW_OK) return r;
    float dt[MW_OR_MAX_FRAMES];
    for (int k = 0; k < n; k++) dt[k] = delta_time[k] * s.mult;  // S/OceanRenderer.cs:223
    float* const f_h = d_height ? d_height : s.fr_height;
    cf* const f_d = d_disp ? reinterpret_cast<cf*>(d_disp) : s.fr_disp;
    float* const f_n = d_normal ? d_normal : s.fr_normal;
    float* const f_w = d_white ? d_white : s.fr_white;
    OrP2Args A2;
    A2.E = s.fr_E; A2.TW = s.TW; A2.height = f_h; A2.disp = f_d; A2.disp_g = s.fr_disp_g; A2.c = s.c;
    A2.height_g = s.want_imag ? s.fr_height_g : nullptr;
    A2.disp_a = s.want_imag ? s.fr_disp_a : nullptr;
    hipError_t e = hipSuccess;
    const bool packed = or_use_packed(s);
    switch (s.M) {
        case 64: e = or_launch_steps_plan<64>(packed, s, dt, n, A2, f_n, f_w, st, ev); break;
        case 128: e = or_launch_steps_plan<128>(packed, s, dt, n, A2, f_n, f_w, st, ev); break;
        case 256: e = or_launch_steps_plan<256>(packed, s, dt, n, A2, f_n, f_w, st, ev); break;
        case 512: e = or_launch_steps_plan<512>(packed, s, dt, n, A2, f_n, f_w, st, ev); break;
        case 1024: e = or_launch_steps_plan<1024>(packed, s, dt, n, A2, f_n, f_w, st, ev); break;
        case 2048: e = or_launch_steps_plan<2048>(packed, s, dt, n, A2, f_n, f_w, st, ev); break;
        case 4096: e = or_launch_steps_plan<4096>(packed, s, dt, n, A2, f_n, f_w, st, ev); break;
        default: g_or_err = "OceanRenderer: unsupported texture size"; return MW_EINVAL;
    }
    if (e != hipSuccess) { g_or_err = std::string("OceanRenderer steps launch: ") + hipGetErrorString(e); return MW_EDEVICE; }
    const size_t last = (size_t)(n - 1) * MM;
    k_or_copy_frame<<<dim3((unsigned)((3 * MM / 4 + 255) / 256)), dim3(256), 0, st>>>(MM, f_h + last, f_d + last, s.fr_disp_g + last, f_n + 3 * last, f_w + last,
                                                    s.want_imag ? s.fr_height_g + last : nullptr, s.want_imag ? s.fr_disp_a + last : nullptr,
                                                    s.out_height, s.out_disp_cf, s.out_disp_g, s.out_normal, s.out_white, s.out_height_g, s.out_disp_a);
    if (hipGetLastError() != hipSuccess) { g_or_err = "OceanRenderer normal/white launch failed"; return MW_EDEVICE; }
    if (ev) hipEventRecord(ev[1 + 3 * or_steps_chunks(s.M, n)], st);
    s.have_frame = true;
    s.have_imag = s.want_imag;
    return MW_OK;
}
// the same n frames delivered as the four ARGBFloat render targets, [n][M*M*4] each (any destination may be NULL)
static inline mw_status or_generate_steps_rgba(OrState& s, const float* delta_time, int n, f4* d_height, f4* d_disp, f4* d_normal, f4* d_white,
                                               hipStream_t st) {
    s.want_imag = true;
    mw_status r = or_generate_steps(s, delta_time, n, nullptr, nullptr, nullptr, nullptr, st);
    s.want_imag = false;
    if (r != MW_OK) return r;
    const size_t MM = (size_t)s.M * s.M;
    k_or_pack_rgba<<<dim3((unsigned)((MM + 255) / 256), n), dim3(256), 0, st>>>(s.M, s.fr_height, s.fr_height_g, s.fr_disp, s.fr_disp_g, s.fr_disp_a,
                                                                           s.fr_normal, s.fr_white, d_height, d_disp, d_normal, d_white);
    if (hipGetLastError() != hipSuccess) { g_or_err = "k_or_pack_rgba launch failed"; return MW_EDEVICE; }
    return MW_OK;
}

// The Dispersion pass alone (F/Dispersion.shader:32-41) for n frames: the phase texture after n GenerateTexture() calls with these delta times,
// no textures produced -- the same or_phase_step chain per texel, so a handle advanced this way continues bit for bit like one that rendered the
// frames.  How a rank of a multi-GPU job seeks to ITS block of a frame sequence (time-steps shard across devices; in this
// semantics only the phase links them), and how a recorder skips frames.
struct OrAdvanceArgs {
    float dt[MW_OR_MAX_FRAMES];
    int n;
};
__global__ __launch_bounds__(256) void k_or_advance(size_t MM, const float* omT, float* phase, OrAdvanceArgs A) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= MM) return;
    const float om = omT[i];
    float ph = phase[i + (size_t)blockIdx.y * MM];  // tile blockIdx.y
    for (int k = 0; k < A.n; k++) ph = or_phase_step(om, ph, A.dt[k]);
    phase[i + (size_t)blockIdx.y * MM] = ph;
}
static inline mw_status or_advance_phase(OrState& s, const float* delta_time, int n, hipStream_t st) {
    const size_t MM = (size_t)s.M * s.M;
    for (int k0 = 0; k0 < n; k0 += MW_OR_MAX_FRAMES) {
        OrAdvanceArgs A;
        A.n = (n - k0 < MW_OR_MAX_FRAMES) ? n - k0 : MW_OR_MAX_FRAMES;
        for (int k = 0; k < MW_OR_MAX_FRAMES; k++) A.dt[k] = k < A.n ? delta_time[k0 + k] * s.mult : 0.f;  // S/OceanRenderer.cs:223
        k_or_advance<<<dim3((unsigned)((MM + 255) / 256), s.tiles), dim3(256), 0, st>>>(MM, s.omT, s.phaseT, A);
    }
    if (hipGetLastError() != hipSuccess) { g_or_err = "k_or_advance launch failed"; return MW_EDEVICE; }
    return MW_OK;
}

// the ocean material's vertex stage on the res x res mesh, from the textures of the latest GenerateTexture()
static inline mw_status or_displace_mesh(OrState& s, int res, float unit_width, float* d_vert, float* d_nrm, float* d_col,
                                         hipStream_t st) {
    if (!s.have_frame) { g_or_err = "displace_mesh: no GenerateTexture() yet"; return MW_ESTATE; }
    const int nv = res * res;
    k_or_displace_mesh<<<dim3((unsigned)((nv + 255) / 256), s.tiles), dim3(256), 0, st>>>(s.M, res, unit_width, s.out_height,
                                                                                 s.out_disp_cf, s.out_normal, s.out_white,
                                                                                 d_vert, d_nrm, d_col);
    if (hipGetLastError() != hipSuccess) { g_or_err = "k_or_displace_mesh launch failed"; return MW_EDEVICE; }
    return MW_OK;
}

}  // namespace mw
