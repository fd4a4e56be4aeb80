// fftmesh_kernels.h -- thread-level bodies of the FFTMesh-semantics kernels (MW_SEM_FFTMESH).
//
// Replaces the O(N^4) hot loop of the reference's CPU path
//   S/FFTMesh.cs:178-190 (htilde), :192-220 (Displacement), :224-280 (EvaluateWaves)
// by two LDS-staged Stockham passes over a Hermitian-packed spectrum (DESIGN.md sections 3-5).
//
// Each kernel is written as a sequence of *phases* separated by workgroup barriers.  A phase is
// an MW_HD function of (block index, thread index, per-thread register state, LDS pointer), so
// the same code is launched by fftmesh.hip on the GPU and stepped in lock-step by the host
// emulation in tests/emul (test infrastructure only).
#pragma once
#include "mw_math.h"

namespace mw {

struct alignas(16) f4 {
    float x, y, z, w;
};

struct OceanConsts {
    int N;
    float length, gravity, unit_width, choppiness;
};

#define MW_MAX_BATCH 32
struct StepTimes {
    float t[MW_MAX_BATCH];
};

// Exchange-buffer geometry.  E[step][f] is stored as [j / CW][a][j % CW]: CW spectrum columns x 16/CW rows make one 128-B
// line, so a pass-1 workgroup that owns CW columns writes whole lines and a pass-2 workgroup that owns 16/CW * k rows reads
// whole (or, for fewer rows, contiguous fractions of) lines.  CW = 4 everywhere except 4096^2, where four 4096-point columns
// need 139 KiB of LDS (one workgroup per CU); with CW = 2 pass 1 is two 70-KiB workgroups per CU (MW_CW2_MIN_N, A/B in
// DESIGN.md section 6).
#ifndef MW_BUF_PAD
#define MW_BUF_PAD 4  // cf entries between the row / column exchange buffers of a workgroup (2: -4 % at 4096^2; 8, 12: neutral)
#endif
#ifndef MW_CW2_MIN_N
#define MW_CW2_MIN_N 8192  // grids from this size up use CW = 2 (8192 = none)
#endif
template <int N> struct Exch { static constexpr int CW = (N >= MW_CW2_MIN_N) ? 2 : 4; };

#ifndef MW_FAST_SINCOS
#define MW_FAST_SINCOS 1  // hardware v_sin/v_cos after an exact reduction (1.8e-7 abs; +1 % at 1024^2, -4 % OceanRenderer frame time)
#endif
MW_HD void mw_sincos(float x, float* s, float* c) {
    if (MW_FAST_SINCOS) sincos_fast_f32(x, s, c); else sincos_f32(x, s, c);
}
// hardware sine/cosine after an exact reduction (1.8e-7 absolute on the device): the VALU-bound pond kernels
MW_HD void mw_sincos_fast(float x, float* s, float* c) { sincos_fast_f32(x, s, c); }
// streaming (write-once) stores: MW_NT_STORES marks them non-temporal so that they do not displace reusable lines
#ifndef MW_NT_STORES
#define MW_NT_STORES 1  // measured: pass 1 -10 % (exchange-buffer stores), pass 2 -3 % (results)
#endif
// Results (vertices, normals, whitecap): non-temporal from 512^2 up (512^2 +5 % step, 1024^2 +2 %, and with the round-2
// kernels 2048^2 +3 %, 4096^2 +2 % in pass 2); slower at 256^2, where everything is cache-resident (-8 % pass 2).
#ifndef MW_NT_RESULTS_MIN_N
#define MW_NT_RESULTS_MIN_N 512
#endif
MW_HD constexpr bool mw_nt_results(int N) { return N >= MW_NT_RESULTS_MIN_N; }
// exchange buffer: non-temporal from 512^2 up (pass 1 -10 %); at 256^2 the whole batch's buffer is cache-resident and
// pass 2 reads it back 12 % slower if it was streamed out
#ifndef MW_NT_EXCHANGE_MIN_N
#define MW_NT_EXCHANGE_MIN_N 512
#endif
MW_HD constexpr bool mw_nt_exchange(int N) { return N >= MW_NT_EXCHANGE_MIN_N; }
template <bool NT = true>
MW_HD void mw_store_stream(float* p, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (MW_NT_STORES && NT) { __builtin_nontemporal_store(v, p); return; }
#endif
    *p = v;
}
template <bool NT = true>
MW_HD void mw_store_stream(cf* p, cf v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (MW_NT_STORES && NT) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        f2v t; t.x = v.x; t.y = v.y;
        __builtin_nontemporal_store(t, reinterpret_cast<f2v*>(p));
        return;
    }
#endif
    *p = v;
}
// A value that is the same in every lane of a wave (e.g. tid / T when T is a multiple of 64): telling the compiler
// lets it keep the value -- and every pointer derived from it -- in SGPRs, so global accesses use the
// `saddr + 32-bit voffset` form instead of one 64-bit VGPR address pair per access.
template <bool NT = true>
MW_HD void mw_store_stream(f4* p, f4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (MW_NT_STORES && NT) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p));
        return;
    }
#endif
    *p = v;
}
#ifndef MW_NT_LOADS
#define MW_NT_LOADS 0
#endif
// MW_NT_LOADS: 1 = every exchange-buffer load non-temporal (-9 % at 1024^2), 2 = only the loads flagged `last_use`
MW_HD cf mw_load_stream(const cf* p, bool last_use = false) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (MW_NT_LOADS == 1 || (MW_NT_LOADS == 2 && last_use)) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        const f2v t = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(p));
        return mk(t.x, t.y);
    }
#endif
    return *p;
}
template <bool WAVE_UNIFORM>
MW_HD int wave_uniform(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return WAVE_UNIFORM ? __builtin_amdgcn_readfirstlane(v) : v;
#else
    return v;
#endif
}
// scheduling fence: keeps the compiler from hoisting every LDS read of an unrolled loop above the arithmetic of its
// first iterations (all reads in flight at once = all their destination registers live at once)
MW_HD void mw_sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}
MW_HD float mw_rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __frsqrt_rn(x);
#else
    return 1.0f / sqrtf(x);
#endif
}
// multipliers only (tolerance-level, not index work): k = (i - N/2) * 2 pi / L, S/FFTMesh.cs:201
MW_HD float wave_k_fast(int N, float kscale, int i) { return (float)(i - N / 2) * kscale; }

// =============================== init-time element kernels ==================================

// S/FFTMesh.cs:149-166 Phillips(n,m) in float32 (device libm; tolerance-checked vs the oracle)
MW_HD float phillips_f32(int N, float length, float wind_x, float wind_y, float amplitude, float gravity, int n, int m) {
    float kx = sdiv((float)(2 * n - N), length) * MW_PI_F;
    float kz = sdiv((float)(2 * m - N), length) * MW_PI_F;
    float k_length = sqrtf(kx * kx + kz * kz);
    if (k_length < MW_EPS_F) return 0.0f;
    float k2 = k_length * k_length, k4 = k2 * k2;
    float knx = 0.f, knz = 0.f;
    if (k_length > 1e-5f) { knx = kx / k_length; knz = kz / k_length; }
    float w_length = sqrtf(wind_x * wind_x + wind_y * wind_y);
    float wnx = 0.f, wny = 0.f;
    if (w_length > 1e-5f) { wnx = wind_x / w_length; wny = wind_y / w_length; }
    float kDotW = knx * wnx + knz * wny;
    float l = w_length * w_length / gravity;
    float l2 = l * l;
    float damping = 0.001f;  // :163
    float L2 = l2 * damping * damping;
    return amplitude * expf(-1.f / (k2 * l2)) / k4 * (kDotW * kDotW) * expf(-k2 * L2);
}

// S/FFTMesh.cs:168-176 htilde0 with the library RNG; :114-116 fill order (4 draws / point)
MW_HD void spectrum_element(int N, float length, float wind_x, float wind_y, float amplitude, float gravity,
                            uint64_t seed, int i, int j, cf* h0, cf* h0c) {
    uint64_t idx = (uint64_t)i * N + j;
    float ph[2] = {phillips_f32(N, length, wind_x, wind_y, amplitude, gravity, i, j),
                   phillips_f32(N, length, wind_x, wind_y, amplitude, gravity, N - i, N - j)};
    cf out[2];
    for (int d = 0; d < 2; d++) {
        float z1 = uniform01(seed, 4 * idx + 2 * d), z2 = uniform01(seed, 4 * idx + 2 * d + 1);
        float rad = sqrtf(-2.f * logf(z1));
        float s, c;
        mw_sincos(2.0f * MW_PI_F * z2, &s, &c);
        float sc = sqrtf(ph[d] / 2.f);
        out[d] = mk(rad * c * sc, rad * s * sc);
    }
    h0[idx] = out[0];
    h0c[idx] = mk(out[1].x, -out[1].y);  // :116
}

// S/FFTMesh.cs:101-139 GenerateMesh, one vertex (and its up-to-two triangles) per call
MW_HD void rest_mesh_element(int N, float unit_width, int i, int j, float* vertices, float* normals, float* uvs,
                             int32_t* indices) {
    int cur = i * N + j;
    if (vertices) {
        vertices[3 * cur + 0] = rest_coord(N, unit_width, i);
        vertices[3 * cur + 1] = 0.f;
        vertices[3 * cur + 2] = rest_coord(N, unit_width, j);
    }
    if (normals) { normals[3 * cur] = 0.f; normals[3 * cur + 1] = 1.f; normals[3 * cur + 2] = 0.f; }
    if (uvs) {
        uvs[2 * cur + 0] = sdiv(smul((float)i, 1.0f), (float)(N - 1));  // :117
        uvs[2 * cur + 1] = sdiv(smul((float)j, 1.0f), (float)(N - 1));
    }
    if (!indices || j == N - 1) return;  // :118
    // the reference appends sequentially; closed form of indiceCount at (i,j):
    //   row 0 and row N-1 emit 3 indices per cell, interior rows 6.
    int64_t base;
    if (i == 0) base = 3 * (int64_t)j;
    else base = 3 * (int64_t)(N - 1) + (int64_t)(i - 1) * 6 * (N - 1) + (i == N - 1 ? 3 : 6) * (int64_t)j;
    if (i != N - 1) {  // :120-125
        indices[base++] = cur; indices[base++] = cur + 1; indices[base++] = cur + N;
    }
    if (i != 0) {  // :126-131
        indices[base++] = cur; indices[base++] = cur - N + 1; indices[base++] = cur + 1;
    }
}

// Hermitian packing tables.  k = (i,j), mk = mirror, s = -1 when exactly one
// index is the Nyquist index 0.  Real output fields only need
//   Hh(k,t) = 1/2 [h~(k,t) + s conj h~(mk,t)] = P e^{i w t} + Q e^{-i w t}
// with P = 1/2 (h0(k) + s conj h0c(mk)), Q = 1/2 (h0c(k) + s conj h0(mk)); on the two Nyquist lines
// the odd multipliers need Ha = Hh + D, D = dP e^{iwt} + dQ e^{-iwt}, dP = -s conj h0c(mk), dQ = -s conj h0(mk).
// PQt is stored TRANSPOSED ([j][i]) so that pass 1 (transform along i) reads contiguous rows, and is
// pre-multiplied by the transform's separable pre-twiddle pre(i+j) = (-1)^(i+j) e^{i pi (i+j)/N}
//  so the per-step kernel never touches it.  Om[j][i] = Dispersion(i,j), strict f32.
MW_HD f4 f4_times(f4 v, cf w) {  // both complex halves times w
    f4 r;
    r.x = v.x * w.x - v.y * w.y; r.y = v.x * w.y + v.y * w.x;
    r.z = v.z * w.x - v.w * w.y; r.w = v.z * w.y + v.w * w.x;
    return r;
}
MW_HD void prep_element(int N, float length, float gravity, int i, int j, const cf* h0, const cf* h0c, const cf* Wpre,
                        f4* PQt, f4* dPQ_i0, f4* dPQ_j0, float* Om) {
    int mi = (N - i) & (N - 1), mj = (N - j) & (N - 1);
    float s = ((i == 0) != (j == 0)) ? -1.f : 1.f;
    cf a = h0[(size_t)i * N + j], b = h0c[(size_t)i * N + j];
    cf am = h0[(size_t)mi * N + mj], bm = h0c[(size_t)mi * N + mj];
    f4 pq;
    pq.x = 0.5f * (a.x + s * bm.x);
    pq.y = 0.5f * (a.y - s * bm.y);
    pq.z = 0.5f * (b.x + s * am.x);
    pq.w = 0.5f * (b.y - s * am.y);
    const cf pre = Wpre[i + j];
    PQt[(size_t)j * N + i] = f4_times(pq, pre);
    Om[(size_t)j * N + i] = omega_f32(N, length, gravity, i, j);
    f4 d;
    d.x = -s * bm.x; d.y = s * bm.y; d.z = -s * am.x; d.w = s * am.y;
    d = f4_times(d, pre);
    if (i == 0) dPQ_i0[j] = d;
    if (j == 0) dPQ_j0[i] = d;
}

// =============================== pass 1: transform along i ===================================
// grid (N/4 + 1, nsteps); block 4*T threads (T = N/P).  Block jb < N/4 owns spectrum columns
// j = 4 jb .. 4 jb + 3.  The extra block jb == N/4 is the *Nyquist-column job*: it transforms the
// correction column cz(i,0) * D'(i,0) through the very same code path and writes it
// to Cj0, which pass 2 adds to element j = 0 of every row.  The i = 0 correction is one element per
// column: thread u == 0 carries it in `dl0` (zero for every other thread, so no branch).
#ifndef MW_NOISE_LDS
#define MW_NOISE_LDS 1
#endif
#ifndef MW_DBUF
#define MW_DBUF 0  // ping-pong exchange sets: measured no gain at 2x the LDS
#endif
struct P1Args {
    const f4* PQt;     // [j][i] (P,Q) * pre
    const f4* dPQ_i0;  // [j]    (dP,dQ) * pre on the row i = 0
    const f4* dPQ_j0;  // [i]    (dP,dQ) * pre on the column j = 0
    const float* Om;   // omega(i,j) at [j][i]
    const cf* TW;      // concatenated twiddle tables (TwGeom)
    cf* E;             // exchange buffer [step][3][N/4][N][4]
    cf* Cj0;           // [step][3][N]  transform of the j = 0 correction column
    OceanConsts c;
    int tgroup;        // > 0: 1-D grid, `tgroup` time-steps of one column job kept on one XCD (p1_block_map)
    int nsteps;
    int field_split = 0;  // single-step enqueues, ONE field per workgroup: 1 = grid (column jobs, 3); 2 = 1-D grid over `jobs`
    const int* jobs = nullptr;  // field_split == 2: workgroup id -> (field << 16 | column job), -1 = none (p1_frame_jobs)
    int njobs = 0;
};

// Pass-1 block -> (column job, time-step).  All time-steps of a batch read the same PQt/Om rows, so the tgroup
// steps of one column job are issued back to back on ONE XCD (the dispatcher places block b on XCD b % 8): the
// first of them pulls the rows into that XCD's L2 and the others hit there instead of crossing the fabric again.
// Speed only -- any bijection is correct.  Returns false for the padding blocks of the rounded-up grid.
MW_HD bool p1_block_map(int bid, int gx, int nsteps, int tgroup, int* jb, int* step) {
    const int xcd = bid % 8, slot = bid / 8;
    const int gg = (slot / tgroup) * 8 + xcd, m = slot % tgroup;
    if (gg >= gx * (nsteps / tgroup)) return false;
    *jb = gg % gx;
    *step = (gg / gx) * tgroup + m;
    return true;
}
MW_HD int p1_grid_blocks(int gx, int nsteps, int tgroup) {
    const int groups = gx * (nsteps / tgroup);
    return ((groups + 7) / 8) * 8 * tgroup;
}

template <int N, int P>
struct P1Geom {
    static constexpr int T = FftGeom<N, P>::T;
    static constexpr int CW = Exch<N>::CW;  // spectrum columns per workgroup
    static constexpr int NTHREADS = CW * T;
    static constexpr int BUFSTRIDE = FftGeom<N, P>::LBUF + (XLay<N, P>::EXACT ? XLay<N, P>::PAD_RD4 : MW_BUF_PAD);
    static constexpr int TW_LDS = (TwGeom<N, P>::LDS_ALL + 1) & ~1;  // cf units, 16-B aligned
    static constexpr int SETSTRIDE = CW * BUFSTRIDE;
    // 2: ping-pong exchange buffers, one barrier per exchange (when both sets fit a 100 KiB budget)
    static constexpr int NBUF = (MW_DBUF && (TW_LDS + 2 * SETSTRIDE) * 8 <= 100 * 1024) ? 2 : 1;
    static constexpr int LDS_BYTES = (TW_LDS + NBUF * SETSTRIDE) * (int)sizeof(cf);
    static constexpr int GRID_X = N / CW + 1;
    static_assert(NTHREADS <= 1024, "workgroup too large: use P = 16 for this N");
};

template <int P>
struct P1State {
    cf hh[P];  // Hh(k,t)*pre for a regular column; D'(i,0) for the Nyquist-column job
    cf dl0;    // D'(0,j) in thread u == 0 of a regular column, else 0
};

// multipliers of the packed fields: Z_f = (cx + cz) * Hh   (cx acts on the kx-odd part, cz on kz-odd)
//   f=0: H                       Z1 = Hh
//   f=1: Dx + i Dz               cx = -i kx/|k|, cz = -kz/|k|      (S/FFTMesh.cs:215, with its z sign)
//   f=2: Sx + i Sz (slopes)      cx = -i kx,     cz = +kz          (S/FFTMesh.cs:212)
MW_HD void field_coeffs(int f, float kx, float kz, cf* cx, cf* cz) {
    if (f == 1) {
        const float k2 = kx * kx + kz * kz;
        float ux = 0.f, uz = 0.f;
        if (!(k2 < MW_EPS_F * MW_EPS_F)) {  // |k| < EPSILON -> skipped, :213
            const float inv = mw_rsqrt(k2);
            ux = kx * inv;
            uz = kz * inv;
        }
        *cx = mk(0.f, -ux);
        *cz = mk(-uz, 0.f);
    } else {
        *cx = mk(0.f, -kx);
        *cz = mk(kz, 0.f);
    }
}

// Which fields a pass-1 block transforms.  The height field (f = 0) transforms to a REAL output, so after the
// first 1-D transform its rows are Hermitian in j: T(a, N-j) = conj T(a, j).  Only columns j <= N/2 are
// transformed and stored (blocks jb <= N/8); pass 2 rebuilds the other half by conjugation (p2_load).  The
// Nyquist-column job (jb == N/4) has no height term at all.
//
// The slope field (f = 2) is half-stored too, MW_SPLIT_SLOPES: Z3 = cx Ha_x + cz Ha_z with cz = kz(j) constant along the
// pass-1 direction, so its first transform is  T3(a,j) = G(a,j) + kz(j) T1(a,j) [+ the Nyquist-column term C3(a) at j = 0],
// G = the transform of the cx part alone.  G belongs to the REAL output Sx, i.e. it is Hermitian in j like the height
// rows T1, and kz(N - j) = -kz(j).  Pass 1 transforms and stores the slope field for the columns j <= N/2 only; pass 2 forms
//   mode 1 (stores G):   T3(a,j) = G'(a,j) + kz(j) T1'(a,j)            -- every element needs its height-row element again
//   mode 2 (stores T3):  T3(a,j) = T3(a,j)                              for j <= N/2
//                        T3(a,j) = conj(T3(a,m) - 2 kz(m) T1(a,m)), m = N - j   -- only the mirrored half re-reads T1
// 2.0 instead of 2.5 complex fields cross the exchange buffer and pass 1 transforms a fifth less.
// Mode 2's T1(a,m) values are exactly what the HEIGHT fetch loaded into the mirrored slots of the same lane: since round 4 they are
// kept in registers until the slope assembly (KeepT1 below) instead of being fetched a second time (4 B per grid point from beyond
// L2; up to round 3 both modes paid it and mode 1 was 0.8 % ahead at 4096^2).  Mode 2 is the plan at every size.
#ifndef MW_SPLIT_SLOPES
#define MW_SPLIT_SLOPES 2
#endif
#ifndef MW_SPLIT_SLOPES_4096
#define MW_SPLIT_SLOPES_4096 MW_SPLIT_SLOPES  // round 4: mode 2 everywhere -- with KeepT1 its height-row re-read is gone (4096^2: 275.0 -> 265.0 us
#endif                                        // per step over 512 steps; before KeepT1 mode 1, the G form, was 0.8 % ahead there)
MW_HD constexpr int mw_split_slopes(int N) { return N >= 4096 ? MW_SPLIT_SLOPES_4096 : MW_SPLIT_SLOPES; }
#ifndef MW_SLOPE_FENCE_Q
#define MW_SLOPE_FENCE_Q 8
#endif
#ifndef MW_KEEP_T1
#define MW_KEEP_T1 1
#endif
#ifndef MW_KEEP_T1_MAX_N
#define MW_KEEP_T1_MAX_N 4096
#endif
MW_HD bool p1_field_active(int N, int jb, int f, int cw = 4) {
    if (f == 1) return true;
    if (f == 2 && (!mw_split_slopes(N) || jb == N / cw)) return true;  // whole field / the Nyquist-column job keeps its cz term
    return jb <= N / (2 * cw);
}

// animated packed spectrum, P points per thread (column job w, i = u + T q)
template <int N, int P>
MW_HD void p1_animate(const P1Args& A, int jb, int tid, float t, P1State<P>& st) {
    constexpr int T = FftGeom<N, P>::T;
    constexpr int CW = Exch<N>::CW;
    const int w = wave_uniform<(T % 64) == 0>(tid / T), u = tid % T;
    const bool fix = (jb == N / CW);
    const int j = fix ? 0 : CW * jb + w;
    const f4* __restrict__ pqrow = fix ? A.dPQ_j0 : A.PQt + (size_t)j * N;  // wave-uniform row base
    const float* __restrict__ omrow = A.Om + (size_t)j * N;
    const bool live = !fix || w == 0;
#pragma unroll
    for (int q = 0; q < P; q++) {
        f4 pq = (pqrow + T * q)[(unsigned)u];
        float s, c;
        mw_sincos(smul((omrow + T * q)[(unsigned)u], t), &s, &c);  // omega*t: one f32 multiply, S/FFTMesh.cs:183
        cf h = animate(pq.x, pq.y, pq.z, pq.w, c, s);
        st.hh[q] = live ? h : mk(0.f, 0.f);
    }
    st.dl0 = mk(0.f, 0.f);
    if (u == 0 && !fix) {  // element i = 0 of this column
        f4 d = A.dPQ_i0[j];
        float s, c;
        mw_sincos(smul(A.Om[(size_t)j * N], t), &s, &c);
        st.dl0 = animate(d.x, d.y, d.z, d.w, c, s);
    }
}

template <int N, int P>
MW_HD void p1_build(const P1Args& A, int jb, int tid, int f, const P1State<P>& st, cf (&x)[P]) {
    constexpr int T = FftGeom<N, P>::T;
    constexpr int CW = Exch<N>::CW;
    const int w = tid / T, u = tid % T;
    const bool fix = (jb == N / CW);
    const int j = fix ? 0 : CW * jb + w;
    if (f == 0) {
#pragma unroll
        for (int q = 0; q < P; q++) x[q] = st.hh[q];
        return;
    }
    const float kscale = 2.0f * MW_PI_F / A.c.length;
    const float kz = wave_k_fast(N, kscale, j);
    const float fx = fix ? 0.f : 1.f;  // the Nyquist-column job keeps only the cz part
#pragma unroll
    for (int q = 0; q < P; q++) {
        cf cx, cz;
        field_coeffs(f, wave_k_fast(N, kscale, u + T * q), kz, &cx, &cz);
        if (mw_split_slopes(N) == 1 && f == 2) x[q] = cmul(fix ? cz : cx, st.hh[q]);  // regular column: G only (see p1_field_active)
        else x[q] = cmul(cscale(cx, fx) + cz, st.hh[q]);
        if (q == 0) x[0] = x[0] + cmul(cx, st.dl0);  // i = 0 correction (dl0 == 0 unless u == 0)
    }
}

}  // namespace mw
#include <vector>
namespace mw {
// Job list of the single-step plan's pass 1: exactly the ACTIVE (column job, field) pairs, one workgroup each, dealt so that the
// fields of one column job sit in consecutive slots of ONE XCD (the dispatcher places workgroup b on XCD b % 8): they re-form the
// same animated spectrum from the same rows at the same time, and one L2 fetches those rows once.  Every XCD gets the column jobs
// jb = x (mod 8); three-field jobs first, the single-field ones (jb > N / (2 CW): displacement only) after; lists padded with -1.
// An exact list instead of a (column jobs, 3) grid with early exits: the dispatcher then loads the CUs evenly (2 workgroups each
// at 1024^2) -- with the exits some CUs ran three and the step waited for those (profiles/r04_ab_notes.md).
inline std::vector<int> p1_frame_jobs(int N, int cw) {
    const int gx = N / cw + 1;
    std::vector<int> per[8];
    for (int pass = 0; pass < 2; pass++)
        for (int jb = 0; jb < gx; jb++) {
            int nf = 0;
            for (int f = 0; f < 3; f++) nf += p1_field_active(N, jb, f, cw) ? 1 : 0;
            if ((nf == 3) != (pass == 0)) continue;
            for (int f = 0; f < 3; f++)
                if (p1_field_active(N, jb, f, cw)) per[jb % 8].push_back((f << 16) | jb);
        }
    size_t len = 0;
    for (auto& v : per) len = v.size() > len ? v.size() : len;
    std::vector<int> out(8 * len, -1);
    for (int x = 0; x < 8; x++)
        for (size_t k = 0; k < per[x].size(); k++) out[8 * k + x] = per[x][k];
    return out;
}

// radix-P passes the pass-1 loop runs between stage 0 and p1_finish (all of them, unless p1_finish runs the last one itself)
template <int N, int P>
MW_HD constexpr int p1_mid_passes() { return FftGeom<N, P>::S - (LastInRegs<N, P>::value ? 1 : 0); }
// final pass in the column-interleaved mapping + coalesced store of the exchange buffer
template <int N, int P>
MW_HD void p1_finish(const P1Args& A, const Twiddles& tw, int jb, int step, int tid, int f, cf (&x)[P], const cf* lds) {
    constexpr int T = FftGeom<N, P>::T;
    constexpr int CW = Exch<N>::CW;
    const int w2 = tid % CW, u2 = tid / CW;
    if (LastInRegs<N, P>::value) {
        // N = P^S (4096 at 16 points): the LAST radix-P pass itself runs in the column-interleaved mapping -- it reads the previous
        // exchange (identity layout; the four buffers 8 entries apart mod 32: conflict-free) and its results, element u2 + T q in slot
        // q, go straight to the exchange buffer in whole lines.  One LDS round trip and two barriers per field less than
        // stage_store + load_last (the caller's pass loop stops one pass early: p1_mid_passes).
        load_slots<N, P>(x, u2, lds + w2 * P1Geom<N, P>::BUFSTRIDE, FftGeom<N, P>::S - 2);
        stage_regs<N, P, +1>(x, u2, tw, FftGeom<N, P>::S - 1);
    } else {
        load_last<N, P>(x, u2, lds + w2 * P1Geom<N, P>::BUFSTRIDE);
        final_stage<N, P, +1>(x, u2, tw.TF);
    }
    if (jb == N / CW) {
        if (w2 == 0) {
            cf* C = A.Cj0 + ((size_t)step * 3 + f) * N;
#pragma unroll
            for (int q = 0; q < P; q++) C[u2 + T * q] = x[q];
        }
        return;
    }
    cf* Ef = A.E + ((size_t)step * 3 + f) * N * N + (size_t)jb * N * CW;  // block-uniform
    const unsigned voff = (unsigned)(u2 * CW + w2);
#pragma unroll
    for (int q = 0; q < P; q++) mw_store_stream<mw_nt_exchange(N)>(&(Ef + (size_t)T * q * CW)[voff], x[q]);
}

// =============================== pass 2: transform along j + epilogue ========================
// grid (N/R2, nsteps); block (R2+1)*T threads: groups 0..R2-1 own rows a0..a0+R2-1, group R2 is the
// halo row a0+R2 (displacement field only) that the forward-difference Jacobian needs (S/FFTMesh.cs:262).
struct P2Args {
    const cf* E;
    const cf* Cj0;    // [step][3][N]
    const cf* TW;     // concatenated twiddle tables (TwGeom)
    float* vertices;  // [step][N*N*3]
    float* normals;   // [step][N*N*3]
    float* white;     // [step][N*N*white_stride]
    int white_stride; // 1 (scalar) or 4 (Unity Color, S/FFTMesh.cs:274)
    OceanConsts c;
    cf* hds_dump = nullptr;  // test hook (mw_debug_evaluate_hds): hds = (d.x, d.z) of S/FFTMesh.cs:247 as [step][a*N + b], else NULL
};

template <int N, int P>
struct P2Buf { static constexpr int BUFSTRIDE = FftGeom<N, P>::LBUF + (XLay<N, P>::EXACT ? XLay<N, P>::PAD_WR4 : MW_BUF_PAD); };  // one row's exchange buffer, cf units

// HS = "sequential halo" variant for large N (Plan<N>::HS): no halo thread group and no halo buffer -- the halo row is
// transformed by group 0 AFTER the displacement field, in buffer 0, once rows 0..R2-2 have formed their Jacobians from
// the published rows.  R2 full rows then fit where R2+1 did not (N = 4096: 4 rows = 136 KiB), so a workgroup consumes
// whole 128-B exchange chunks.
template <int N, int P, int R2, bool HS = false>
struct P2Geom {
    static constexpr int T = FftGeom<N, P>::T;
    static constexpr int NGROUPS = HS ? R2 : R2 + 1;
    static constexpr int NTHREADS = NGROUPS * T;
    static constexpr int BUFSTRIDE = P2Buf<N, P>::BUFSTRIDE;
    static constexpr int TW_LDS = (TwGeom<N, P>::LDS_ALL + 1) & ~1;
    static constexpr int SETSTRIDE = NGROUPS * BUFSTRIDE;
    static constexpr int NBUF = (!HS && MW_DBUF && (TW_LDS + 2 * SETSTRIDE) * 8 <= 100 * 1024) ? 2 : 1;
    // the whitecap noise term |0.3 n.xz| waits in LDS (R2*N floats) from the slope field to the epilogue instead of
    // in P registers per thread: that is what lets pass 2 fit 80 VGPRs without scratch spills
    static constexpr bool NOISE_REG = HS || !MW_NOISE_LDS;
    static constexpr int NOISE_OFF = TW_LDS + NBUF * SETSTRIDE;  // cf units
    static constexpr int NOISE_CF = NOISE_REG ? 0 : R2 * N / 2;
#ifndef MW_P2_LDS_EXTRA
#define MW_P2_LDS_EXTRA 0  // occupancy experiments: bytes of unused LDS per pass-2 workgroup
#endif
    static constexpr int LDS_BYTES = (NOISE_OFF + NOISE_CF) * (int)sizeof(cf) + MW_P2_LDS_EXTRA;
    static_assert(NTHREADS <= 1024, "workgroup too large");
};

template <int P, bool NOISE_REG_ = !MW_NOISE_LDS>
struct P2State {
    static constexpr bool NOISE_REG = NOISE_REG_;
    float noise[NOISE_REG_ ? P : 1];  // |0.3 n.xz| per slot (S/FFTMesh.cs:269-270) when it is not parked in LDS
    float h[P];      // height
    cf d[P];         // hds = (d.x, d.z), un-scaled by choppiness (S/FFTMesh.cs:247)
};

// field processing order in pass 2: slopes, height, displacement (displacement last: its LDS buffers
// are then recycled as the hds neighbour exchange)
MW_HD int p2_field(int k) { return k == 0 ? 2 : (k == 1 ? 0 : 1); }

template <int N, int P, int R2>
MW_HD bool p2_active(int ab, int tid, int f) {
    constexpr int T = FftGeom<N, P>::T;
    const int g = tid / T;
    return g < R2 || (f == 1 && ab * R2 + R2 < N);
}

// (row slot, thread-in-row) of the load-side mapping: rows interleaved in consecutive lanes so that
// 4*R2 consecutive lanes read one contiguous R2*32-byte piece of the exchange buffer
template <int N, int P, int R2>
MW_HD void p2_load_map(int tid, int* r1, int* u1) {
    constexpr int T = FftGeom<N, P>::T;
    if (tid < R2 * T) { *r1 = tid % R2; *u1 = tid / R2; }
    else { *r1 = R2; *u1 = tid - R2 * T; }
}

// global loads of one field's row data (row-interleaved mapping).  Split from the stage-0 pass so that the kernel
// can issue field k+1's loads before it starts field k's exchanges (software prefetch: twice the bytes in flight).
// PART (half-stored slope field only, mw_split_slopes): 0 = the whole load; 1 = the stored half alone, raw (G in mode 1, T3 in
// mode 2); 2 = the assembly on top of part 1 -- the height-row loads, the kz T1 term and the conjugation of the mirrored
// slots.  The sequential-halo kernel issues part 1 of all its virtual threads, then part 2 + stage 0 one virtual thread at a
// time: the height rows' registers are then live for one virtual thread only.
// nyq != nullptr: the Nyquist-column term is returned there instead of being added to x[0] (a prefetch must not consume any
// of its loads: the add's s_waitcnt would wait for all of them, vmcnt being in-order); the caller adds it when it uses x.
// keep (KeepT1, mode 2 of the half-stored slope field): the height fetch (f = 0) leaves the RAW values of its mirrored slots q >= P/2
// there -- element (a, N - j) of the height rows, before the conjugation --, and the slope assembly (f = 2, PART 0 / 2) takes them from
// there instead of fetching the same exchange-buffer words a second time (4 B per grid point that had left the L2 by then).
template <int N, int P>
struct KeepT1 {
    static constexpr bool value = MW_KEEP_T1 && mw_split_slopes(N) == 2 && N <= MW_KEEP_T1_MAX_N && (FftGeom<N, P>::T % Exch<N>::CW == 0) &&
                                  ((N / 2) % FftGeom<N, P>::T == 0);
};
template <int N, int P, int R2, int PART = 0>
MW_HD void p2_fetch(const P2Args& A, int ab, int step, int tid, int f, cf (&x)[P], cf* nyq = nullptr, cf* keep = nullptr) {
    constexpr int T = FftGeom<N, P>::T;
    int r1, u1;
    p2_load_map<N, P, R2>(tid, &r1, &u1);
    const int row = ab * R2 + r1;
    constexpr int CW = Exch<N>::CW;
    const cf* Ef = A.E + ((size_t)step * 3 + f) * N * N + (size_t)ab * R2 * CW;  // block-uniform base (first row of the block)
    // per-lane 32-bit offset of element j = u1 (slot q adds the uniform T*q/CW chunks of N*CW)
    const unsigned voff = (unsigned)(((u1 / CW) * N + r1) * CW + (u1 % CW));
    // Half-stored fields (height; with MW_SPLIT_SLOPES also G): element j > N/2 is conj of element m = N - j.  With T a
    // multiple of CW, slot q is mirrored for every lane (T q > N/2), for none (T (q + 1) <= N/2), or -- the one slot with
    // T q == N/2 -- for the lanes u1 > 0: the decision is compile-time and two per-lane offsets serve all slots,
    //   plain:    voff  + q (T/CW) N CW          mirrored (m = N - u1 - T q):   voffm - q (T/CW) N CW
    const bool half = (f == 0) || (mw_split_slopes(N) && f == 2);
    if (half && T % CW == 0 && (N / 2) % T == 0) {
        const int m0 = N - u1;  // u1 == 0: m0 = N is never dereferenced with q = 0 (slot 0 is plain)
        const unsigned voffm = (unsigned)(((m0 / CW) * N + r1) * CW + (m0 % CW));
        const cf* E0 = A.E + ((size_t)step * 3 + 0) * N * N + (size_t)ab * R2 * CW;  // height rows (for the slope assembly)
        const float kscale = 2.0f * MW_PI_F / A.c.length;
#pragma unroll
        for (int q = 0; q < P; q++) {
            const bool all_mir = T * q > N / 2, edge = (T * q == N / 2);
            const bool mir = all_mir || (edge && u1 > 0);
            const size_t chunk = (size_t)(T / CW) * q * N * CW;
            // the edge slot's two candidates differ per lane: select the (non-negative) offset, not the data
            const size_t ub = edge ? 0 : chunk;  // uniform part
            const unsigned off = all_mir ? voffm : (edge ? (u1 > 0 ? voffm - (unsigned)chunk : voff + (unsigned)chunk) : voff);
            cf v = (PART == 2) ? x[q] : mw_load_stream(&(all_mir ? Ef - ub : Ef + ub)[off], f != 0);
            if (PART == 1) { x[q] = v; continue; }
            if (f == 0 && keep && (all_mir || edge)) keep[q - P / 2] = v;  // T q >= N/2  <=>  q >= P/2
            if (mw_split_slopes(N) == 1 && f == 2) {  // T3(a,j) = G'(a,j) + kz(j) T1'(a,j)
                const cf t = mw_load_stream(&(all_mir ? E0 - ub : E0 + ub)[off], true);
                const float kz = wave_k_fast(N, kscale, u1 + T * q);
                v = mk(__builtin_fmaf(kz, t.x, v.x), __builtin_fmaf(kz, t.y, v.y));
            }
            if (mw_split_slopes(N) == 2 && f == 2 && (all_mir || edge)) {  // T3(a,j) = conj(T3(a,m) - 2 kz(m) T1(a,m)), kz(m) = -kz(j)
                const cf t = keep ? keep[q - P / 2] : mw_load_stream(&(all_mir ? E0 - ub : E0 + ub)[off], true);
                const float k2 = 2.0f * wave_k_fast(N, kscale, u1 + T * q);
                const float c = mir ? k2 : 0.f;  // the edge slot's lane u1 == 0 is j = N/2: plain
                v = mk(__builtin_fmaf(c, t.x, v.x), __builtin_fmaf(c, t.y, v.y));
            }
            x[q] = mir ? cconj(v) : v;
            if (PART == 2 && MW_SLOPE_FENCE_Q && q % MW_SLOPE_FENCE_Q == MW_SLOPE_FENCE_Q - 1) mw_sched_fence();  // cap the height-row loads in flight
        }
        if (PART != 1 && f != 0) {  // Nyquist column j = 0
            if (nyq) *nyq = (u1 == 0) ? A.Cj0[((size_t)step * 3 + f) * N + row] : mk(0.f, 0.f);
            else if (u1 == 0) x[0] = x[0] + A.Cj0[((size_t)step * 3 + f) * N + row];
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int j = u1 + T * q;
        if (half && j > N / 2) {  // stored for j <= N/2 only; T(a, j) = conj T(a, N - j)
            const int m = N - j;
            const unsigned off = (unsigned)(((m / CW) * N + r1) * CW + (m % CW));
            cf v = mw_load_stream(&Ef[off]);
            if (mw_split_slopes(N) && f == 2) {
                const cf t = mw_load_stream(&(A.E + ((size_t)step * 3 + 0) * N * N + (size_t)ab * R2 * CW)[off]);
                const float kz = (mw_split_slopes(N) == 2 ? 2.0f : 1.0f) * wave_k_fast(N, 2.0f * MW_PI_F / A.c.length, j);
                v = mk(__builtin_fmaf(kz, t.x, v.x), __builtin_fmaf(kz, t.y, v.y));
            }
            x[q] = cconj(v);
        } else {
            const unsigned off = (unsigned)(((j / CW) * N + r1) * CW + (j % CW));
            cf v = (T % CW == 0) ? mw_load_stream(&(Ef + (size_t)(T / CW) * q * N * CW)[voff], f != 0) : Ef[off];
            if (mw_split_slopes(N) == 1 && f == 2) {
                const cf t = mw_load_stream(&(A.E + ((size_t)step * 3 + 0) * N * N + (size_t)ab * R2 * CW)[off]);
                const float kz = wave_k_fast(N, 2.0f * MW_PI_F / A.c.length, j);
                v = mk(__builtin_fmaf(kz, t.x, v.x), __builtin_fmaf(kz, t.y, v.y));
            }
            x[q] = v;
        }
    }
    if (f != 0) {  // Nyquist column j = 0
        if (nyq) *nyq = (u1 == 0) ? A.Cj0[((size_t)step * 3 + f) * N + row] : mk(0.f, 0.f);
        else if (u1 == 0) x[0] = x[0] + A.Cj0[((size_t)step * 3 + f) * N + row];
    }
}
template <int N, int P, int R2>
MW_HD void p2_stage0(int tid, cf (&x)[P], cf* lds) {
    int r1, u1;
    p2_load_map<N, P, R2>(tid, &r1, &u1);
    stage0_store<N, P, +1>(x, u1, lds + r1 * P2Buf<N, P>::BUFSTRIDE);
}
template <int N, int P, int R2>
MW_HD void p2_load(const P2Args& A, int ab, int step, int tid, int f, cf (&x)[P], cf* lds, cf* keep = nullptr) {
    p2_fetch<N, P, R2>(A, ab, step, tid, f, x, nullptr, keep);
    p2_stage0<N, P, R2>(tid, x, lds);
}
// the slope field can be loaded in two parts (p2_fetch) when the fast half-field path applies
template <int N, int P>
struct P2SlopeParts { static constexpr bool value = mw_split_slopes(N) != 0 && (FftGeom<N, P>::T % Exch<N>::CW == 0); };

// middle passes: with the padded LDS layout they keep the load-side (row-interleaved) mapping (measured fewer bank
// conflicts than row-major); with the exact layouts (XLay) a wave stays inside one row, where every access is conflict-free
template <int N, int P, int R2>
MW_HD void p2_mid_map(int tid, int* r1, int* u1) {
    if (XLay<N, P>::EXACT) { *r1 = tid / FftGeom<N, P>::T; *u1 = tid % FftGeom<N, P>::T; }
    else p2_load_map<N, P, R2>(tid, r1, u1);
}
template <int N, int P, int R2>
MW_HD void p2_mid_load(int tid, int s, cf (&x)[P], const cf* lds) {  // s: the pass this load feeds
    int r1, u1;
    p2_mid_map<N, P, R2>(tid, &r1, &u1);
    load_slots<N, P>(x, u1, lds + r1 * P2Buf<N, P>::BUFSTRIDE, s - 1);
}
template <int N, int P, int R2>
MW_HD void p2_mid_store(const Twiddles& tw, int tid, int s, cf (&x)[P], cf* lds) {
    int r1, u1;
    p2_mid_map<N, P, R2>(tid, &r1, &u1);
    if (mw_pass_in_regs<N, P>(s)) stage_last_regs<N, P, +1>(x, u1, tw, s);  // stays in registers (or moves inside the wave): p2_last_load reads nothing
    else stage_store<N, P, +1>(x, u1, lds + r1 * P2Buf<N, P>::BUFSTRIDE, tw, s);
}
// input of the final pass in the row-major mapping: the last exchange, unless the last radix-P pass left it in registers (LastInRegs:
// the middle passes of an exact layout run in the same row-major mapping, p2_mid_map)
template <int N, int P>
MW_HD void p2_last_load(cf (&x)[P], int u, const cf* buf) {
    if (LastStays<N, P>::value) load_last_regs<N, P>(x);
    else load_last<N, P>(x, u, buf);
}

// final pass in the row-major mapping (thread (g,u) owns row a0+g, columns b = u + T q)
template <int N, int P, int R2, class ST>
MW_HD void p2_finish(const P2Args& A, const Twiddles& tw, int ab, int step, int tid, int f, cf (&x)[P], ST& st,
                     const cf* lds, float* noise_lds) {
    constexpr int T = FftGeom<N, P>::T;
    const int g = tid / T, u = tid % T, a = ab * R2 + g;
    p2_last_load<N, P>(x, u, lds + g * P2Buf<N, P>::BUFSTRIDE);
    final_stage<N, P, +1>(x, u, tw.TF);
    if (f == 2) {  // slopes -> unit normal (S/FFTMesh.cs:218), stored at once
        float* nblk = A.normals + ((size_t)step * N * N + (size_t)ab * R2 * N) * 3;  // block-uniform
        const unsigned noff = (unsigned)((g * N + u) * 3);
#pragma unroll
        for (int q = 0; q < P; q++) {
            const int b = u + T * q;
            float* nq = nblk + (size_t)T * q * 3;  // uniform; element (g, b, c) = nq[noff + c]
            const float sg = post_sign(a, b);
            const float sx = sg * x[q].x, sz = sg * x[q].y;
            // Vector3.Normalize(up - n): |(sx,1,sz)| >= 1, so Unity's 1e-5 zero-guard never fires
            const float inv = mw_rsqrt(__builtin_fmaf(sz, sz, __builtin_fmaf(sx, sx, 1.0f)));  // FMAs written out: the same bits in every instantiation
            const float nx = sx * inv, ny = inv, nz = sz * inv;
            mw_store_stream<mw_nt_results(N)>(&nq[noff + 0], nx);
            mw_store_stream<mw_nt_results(N)>(&nq[noff + 1], ny);
            mw_store_stream<mw_nt_results(N)>(&nq[noff + 2], nz);
            const float n0 = smul(fabsf(nx), 0.3f), n1 = smul(fabsf(nz), 0.3f);
            const float nz_ = ssqrt(sadd(smul(n0, n0), smul(n1, n1)));
            if (!ST::NOISE_REG) noise_lds[g * N + b] = nz_; else st.noise[ST::NOISE_REG ? q : 0] = nz_;
        }
    } else if (f == 0) {
#pragma unroll
        for (int q = 0; q < P; q++) st.h[q] = post_sign(a, u + T * q) * x[q].x;
    } else {
#pragma unroll
        for (int q = 0; q < P; q++) {
            const float sg = post_sign(a, u + T * q);
            st.d[q] = mk(sg * x[q].x, sg * x[q].y);
        }
    }
}

// hds rows into LDS (plain index b) so that neighbours (a+1,b) and (a,b+1) can be read back
template <int N, int P, int R2, class ST>
MW_HD void p2_publish_hds(int tid, const ST& st, cf* lds) {
    constexpr int T = FftGeom<N, P>::T;
    const int g = tid / T, u = tid % T;
    cf* row = lds + g * P2Buf<N, P>::BUFSTRIDE;
#pragma unroll
    for (int q = 0; q < P; q++) row[u + T * q] = st.d[q];
}

// test hook (P2Args::hds_dump, kernels instantiated with DUMP = true only for mw_debug_evaluate_hds): the R2 published hds
// rows of this block, LDS -> global.  The product kernels (DUMP = false) contain none of it: even a never-taken branch
// changed the register allocation of the 2-virtual-thread kernels (256 VGPRs -> spills).
template <int N, int P, int R2>
MW_HD void p2_dump_hds(const P2Args& A, int ab, int step, int lane, int nlanes, const cf* lds) {
    cf* out = A.hds_dump + (size_t)step * N * N + (size_t)ab * R2 * N;
    for (int i = lane; i < R2 * N; i += nlanes) out[i] = lds[(i / N) * P2Buf<N, P>::BUFSTRIDE + (i % N)];
}

// S/FFTMesh.cs:243-247: the displaced vertex
template <int N, int P, int R2, class ST>
MW_HD void p2_vertices(const P2Args& A, int ab, int step, int tid, const ST& st) {
    constexpr int T = FftGeom<N, P>::T;
    const int g = tid / T, u = tid % T, a = ab * R2 + g;
    float* vblk = A.vertices + ((size_t)step * N * N + (size_t)ab * R2 * N) * 3;          // block-uniform
    const unsigned voff = (unsigned)((g * N + u) * 3);
    const float rx = rest_coord(N, A.c.unit_width, a);
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int b = u + T * q;
        const cf d = st.d[q];
        float* vq = vblk + (size_t)T * q * 3;                                              // uniform
        mw_store_stream<mw_nt_results(N)>(&vq[voff + 0], ssub(rx, smul(d.x, A.c.choppiness)));                                // :245
        mw_store_stream<mw_nt_results(N)>(&vq[voff + 1], st.h[q]);                                                            // :243
        mw_store_stream<mw_nt_results(N)>(&vq[voff + 2], ssub(rest_coord(N, A.c.unit_width, b), smul(d.y, A.c.choppiness)));  // :244
    }
}

// neighbour provider of the halo-group variant: the row's own published copy
struct NbRow {
    const cf* row;
    template <int P>
    MW_HD cf right(const cf (&)[P], int, int b) const { return row[b + 1]; }
};

// S/FFTMesh.cs:258-268 for slot q of thread (g,u): 1 - J, J the Jacobian of the choppy displacement from forward
// differences.  nxt = published hds of row a+1; nb.right() = hds of (a, b+1).
template <int N, int P, int R2, class ST, class NB>
MW_HD float p2_one_minus_jacobian(int a, int b, int q, const ST& st, const cf* nxt, const NB& nb) {
    const bool has_i = (a != N - 1), has_j = (b != N - 1);
    const cf d = st.d[q];
    const cf rj = nb.right(st.d, q, has_j ? b : b - 1);
    const cf dn_i = has_i ? nxt[b] : mk(0.f, 0.f);
    const cf dn_j = has_j ? rj : mk(0.f, 0.f);
    float ax = 0.f, ay = 0.f, bx = 0.f, by = 0.f;
    if (has_i) { ax = smul(0.5f, ssub(d.x, dn_i.x)); ay = smul(0.5f, ssub(d.y, dn_i.y)); }  // :260-263
    if (has_j) { bx = smul(0.5f, ssub(d.x, dn_j.x)); by = smul(0.5f, ssub(d.y, dn_j.y)); }  // :264-267
    const float jac = ssub(smul(sadd(1.f, ax), sadd(1.f, by)), smul(ay, bx));              // :268
    return ssub(1.f, jac);
}
// S/FFTMesh.cs:269-274: turbulence = max(1 - J + |0.3 n.xz|, 0) -> smoothstep -> Color
template <bool NT>
MW_HD void p2_store_white(float* wq, unsigned woff, int white_stride, float one_minus_jac, float noise) {
    const float turb = fmaxf(sadd(one_minus_jac, noise), 0.f);  // :270
    const float xx = smoothstep01(turb);                        // :273
    if (white_stride == 1) {
        mw_store_stream<NT>(&wq[woff], xx);
    } else {
        mw_store_stream<NT>(&wq[woff + 0], xx); mw_store_stream<NT>(&wq[woff + 1], xx);  // :274
        mw_store_stream<NT>(&wq[woff + 2], xx); mw_store_stream<NT>(&wq[woff + 3], xx);
    }
}
// halo-group variant: vertex + whitecap of one thread's slots from the rows published in LDS, slot by slot (measured
// 1 % faster than all vertices first, then all whitecaps)
template <int N, int P, int R2, class ST>
MW_HD void p2_epilogue(const P2Args& A, int ab, int step, int tid, const ST& st, const cf* lds, const float* noise_lds) {
    constexpr int T = FftGeom<N, P>::T;
    const int g = tid / T, u = tid % T, a = ab * R2 + g;
    NbRow nb;
    nb.row = lds + g * P2Buf<N, P>::BUFSTRIDE;
    const cf* nxt = lds + (g + 1) * P2Buf<N, P>::BUFSTRIDE;
    float* vblk = A.vertices + ((size_t)step * N * N + (size_t)ab * R2 * N) * 3;          // block-uniform
    float* wblk = A.white + ((size_t)step * N * N + (size_t)ab * R2 * N) * A.white_stride;
    const unsigned voff = (unsigned)((g * N + u) * 3), woff = (unsigned)((g * N + u) * A.white_stride);
    const float rx = rest_coord(N, A.c.unit_width, a);
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int b = u + T * q;
        const cf d = st.d[q];
        float* vq = vblk + (size_t)T * q * 3;                                              // uniform
        mw_store_stream<mw_nt_results(N)>(&vq[voff + 0], ssub(rx, smul(d.x, A.c.choppiness)));                                // :245
        mw_store_stream<mw_nt_results(N)>(&vq[voff + 1], st.h[q]);                                                            // :243
        mw_store_stream<mw_nt_results(N)>(&vq[voff + 2], ssub(rest_coord(N, A.c.unit_width, b), smul(d.y, A.c.choppiness)));  // :244
        const float omj = p2_one_minus_jacobian<N, P, R2>(a, b, q, st, nxt, nb);
        const float nz_ = !ST::NOISE_REG ? noise_lds[g * N + b] : st.noise[ST::NOISE_REG ? q : 0];
        p2_store_white<mw_nt_results(N)>(wblk + (size_t)T * q * A.white_stride, woff, A.white_stride, omj, nz_);
    }
}

// ---- sequential-halo variant (P2Geom<..., true>) ---------------------------------------------------------------
// Field order height, displacement, [vertices out, halo row, 1 - J per slot], slopes: the whitecap is finished in the
// slope field's final pass, so no noise term waits anywhere and at most h+x / d+x / omj+x are live at a time.
template <int P>
struct P2StateHS {
    float h[P];    // height                    (dead once the vertices are stored)
    cf d[P];       // hds                       (dead once omj is formed)
    float omj[P];  // 1 - Jacobian per slot     (born after the halo row)
};
MW_HD int p2_hs_field(int k) { return k == 0 ? 0 : (k == 1 ? 1 : 2); }

// final pass of the height (f = 0) or displacement (f = 1) field, row-major mapping
template <int N, int P, int R2>
MW_HD void p2_hs_finish(const Twiddles& tw, int ab, int tid, int f, cf (&x)[P], P2StateHS<P>& st, const cf* lds) {
    constexpr int T = FftGeom<N, P>::T;
    const int g = tid / T, u = tid % T, a = ab * R2 + g;
    p2_last_load<N, P>(x, u, lds + g * P2Buf<N, P>::BUFSTRIDE);
    final_stage<N, P, +1>(x, u, tw.TF);
#pragma unroll
    for (int q = 0; q < P; q++) {
        const float sg = post_sign(a, u + T * q);
        if (f == 0) st.h[q] = sg * x[q].x;
        else st.d[q] = mk(sg * x[q].x, sg * x[q].y);
    }
}
// 1 - J of a thread's slots: d from registers, (a, b+1) from the row's own published copy, (a+1, b) from nxt
template <int N, int P, int R2>
MW_HD void p2_hs_jacobian(int ab, int tid, P2StateHS<P>& st, const cf* own, const cf* nxt) {
    constexpr int T = FftGeom<N, P>::T;
    const int g = tid / T, u = tid % T, a = ab * R2 + g;
    NbRow nb;
    nb.row = own;
#pragma unroll
    for (int q = 0; q < P; q++) {
        st.omj[q] = p2_one_minus_jacobian<N, P, R2>(a, u + T * q, q, st, nxt, nb);
        if (q % 4 == 3) mw_sched_fence();
    }
}
// the same entirely from LDS (own = the row's published hds copy, nxt = row a+1 or the halo row): the block's last row
// after the halo transform, and -- with P = 16, where d[] alone is 32 VGPRs -- every row (HS_JAC_FROM_LDS)
template <int P>
struct P2RowView { cf d[P]; };
template <int N, int P, int R2>
MW_HD void p2_hs_jacobian_lds(int ab, int tid, P2StateHS<P>& st, const cf* own, const cf* nxt) {
    constexpr int T = FftGeom<N, P>::T;
    const int g = tid / T, u = tid % T, a = ab * R2 + g;
    NbRow nb;
    nb.row = own;
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int b = u + T * q;
        P2RowView<P> v;
        v.d[q] = own[b];
        st.omj[q] = p2_one_minus_jacobian<N, P, R2>(a, b, q, v, nxt, nb);
        if (q % 4 == 3) mw_sched_fence();
    }
}
// final pass of the slope field: unit normal (S/FFTMesh.cs:218) and, with the waiting 1 - J, the whitecap
template <int N, int P, int R2>
MW_HD void p2_hs_finish_slopes(const P2Args& A, const Twiddles& tw, int ab, int step, int tid, cf (&x)[P],
                               const P2StateHS<P>& st, const cf* lds) {
    constexpr int T = FftGeom<N, P>::T;
    const int g = tid / T, u = tid % T, a = ab * R2 + g;
    p2_last_load<N, P>(x, u, lds + g * P2Buf<N, P>::BUFSTRIDE);
    final_stage<N, P, +1>(x, u, tw.TF);
    float* nblk = A.normals + ((size_t)step * N * N + (size_t)ab * R2 * N) * 3;  // block-uniform
    float* wblk = A.white + ((size_t)step * N * N + (size_t)ab * R2 * N) * A.white_stride;
    const unsigned noff = (unsigned)((g * N + u) * 3), woff = (unsigned)((g * N + u) * A.white_stride);
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int b = u + T * q;
        float* nq = nblk + (size_t)T * q * 3;
        const float sg = post_sign(a, b);
        const float sx = sg * x[q].x, sz = sg * x[q].y;
        const float inv = mw_rsqrt(__builtin_fmaf(sz, sz, __builtin_fmaf(sx, sx, 1.0f)));  // FMAs written out: the same bits in every instantiation
        const float nx = sx * inv, ny = inv, nz = sz * inv;
        mw_store_stream<mw_nt_results(N)>(&nq[noff + 0], nx);
        mw_store_stream<mw_nt_results(N)>(&nq[noff + 1], ny);
        mw_store_stream<mw_nt_results(N)>(&nq[noff + 2], nz);
        const float n0 = smul(fabsf(nx), 0.3f), n1 = smul(fabsf(nz), 0.3f);
        const float nz_ = ssqrt(sadd(smul(n0, n0), smul(n1, n1)));  // :269
        p2_store_white<mw_nt_results(N)>(wblk + (size_t)T * q * A.white_stride, woff, A.white_stride, st.omj[q], nz_);
    }
}

// halo row a0+R2 of the displacement field, loaded by group 0 in the row-major mapping (thread u: j = u + T q)
template <int N, int P, int R2>
MW_HD void p2_hs_halo_fetch(const P2Args& A, int ab, int step, int u, cf (&x)[P], cf* nyq = nullptr) {
    constexpr int T = FftGeom<N, P>::T;
    const int row = ab * R2 + R2;
    constexpr int CW = Exch<N>::CW;
    const cf* Ef = A.E + ((size_t)step * 3 + 1) * N * N + (size_t)row * CW;  // block-uniform
    const unsigned voff = (unsigned)((u / CW) * N * CW + (u % CW));
#pragma unroll
    for (int q = 0; q < P; q++) x[q] = (Ef + (size_t)(T / CW) * q * N * CW)[voff];
    if (nyq) *nyq = (u == 0) ? A.Cj0[((size_t)step * 3 + 1) * N + row] : mk(0.f, 0.f);
    else if (u == 0) x[0] = x[0] + A.Cj0[((size_t)step * 3 + 1) * N + row];  // Nyquist column j = 0
}
// transformed halo row -> plain hds row in buffer 0
template <int N, int P, int R2>
MW_HD void p2_hs_halo_publish(int ab, int u, const cf (&x)[P], cf* buf0) {
    constexpr int T = FftGeom<N, P>::T;
    const int a = ab * R2 + R2;
#pragma unroll
    for (int q = 0; q < P; q++) {
        const float sg = post_sign(a, u + T * q);
        buf0[u + T * q] = mk(sg * x[q].x, sg * x[q].y);
    }
}

// ---- frame variant (k_pass2_frame, single-step enqueues): the three fields of a row block side by side ----------------
// One step at 1024^2 is 256 row blocks on 256 CUs: its latency is that of ONE workgroup, and the sequential-halo kernel runs four
// transforms (height, displacement, halo row, slopes) one after the other in it.  Here a workgroup is 3 R2 + 1 row groups of T
// threads: group fg * R2 + g transforms row a0 + g of field fg, the last group the halo row, ALL AT ONCE in 3 R2 + 1 row buffers.
// After the final pass the fields meet through LDS, and the three kinds of group share the stores: the displacement groups
// publish hds and the slope groups store the normals and leave the whitecap's noise term in their row buffers; one barrier; then
// the HEIGHT groups store the vertices (h from their registers, hds from the published rows) while the displacement groups form
// 1 - J and store the whitecap.  Every value is formed by the expressions of the batched plan: the same bits.
#ifndef MW_FRAME_NT_RESULTS
#define MW_FRAME_NT_RESULTS 1
#endif
#ifndef MW_FRAME_R2
#define MW_FRAME_R2 4  // rows per workgroup of k_pass2_frame at 1024^2 (2: 16.9 against 15.6 us)
#endif
#ifndef MW_FRAME_R2_SMALL
#define MW_FRAME_R2_SMALL 2
#endif
// grids whose single-step enqueues run the frame plan (a wave must hold whole row groups: N / P2 == 64 or 32), and the rows per
// workgroup there: 512^2 has 128 4-row blocks for 256 CUs, so 2 rows (256 workgroups of 7 waves); 256^2: 128 workgroups of 3.5 waves
MW_HD constexpr bool mw_frame_plan_n(int N) { return N == 256 || N == 512 || N == 1024; }
MW_HD constexpr int mw_frame_r2(int N) { return N <= 512 ? MW_FRAME_R2_SMALL : MW_FRAME_R2; }
template <int N, int P, int R2>
struct P2FrameGeom {
    static constexpr int T = FftGeom<N, P>::T;
    static constexpr int FT = R2 * T;            // threads of one field's row groups
    static constexpr int NTHREADS = 3 * FT + T;  // + the halo row's group
    static constexpr int BUFSTRIDE = P2Buf<N, P>::BUFSTRIDE;
    static constexpr int SETSTRIDE = R2 * BUFSTRIDE;
    static constexpr int TW_LDS = (TwGeom<N, P>::LDS_ALL + 1) & ~1;
    static constexpr int LDS_BYTES = (TW_LDS + 3 * SETSTRIDE + BUFSTRIDE) * (int)sizeof(cf);
    static constexpr bool FITS = NTHREADS <= 1024 && LDS_BYTES <= 160 * 1024 && BUFSTRIDE >= N;
    // on the device a wave must hold whole row groups of ONE field (the field is treated as wave-uniform, and a row buffer is owned by one
    // wave from the final pass on): one wave per row (T == 64), or two rows per wave (T == 32, R2 even)
    static constexpr bool OK = FITS && (T == 64 || T == 32) && (FT % 64) == 0;
};
// after the final pass, before the barrier.  tl = thread within the field's groups (row g = tl / T), x = its transformed row.
// A row buffer is read in the final pass by its own row group alone, and that group writes it here.
template <int N, int P, int R2>
MW_HD void p2_frame_hds(int ab, int tl, const cf (&x)[P], cf* set_d) {
    constexpr int T = FftGeom<N, P>::T;
    const int g = tl / T, u = tl % T, a = ab * R2 + g;
    cf* drow = set_d + g * P2Buf<N, P>::BUFSTRIDE;
#pragma unroll
    for (int q = 0; q < P; q++) {
        const float sg = post_sign(a, u + T * q);
        drow[u + T * q] = mk(sg * x[q].x, sg * x[q].y);
    }
}
// slopes -> unit normal, stored at once -- BEFORE the barrier: the slope groups run at the highest issue priority and their stores
// overlap the other groups' transforms (all normals stored after the barrier instead: pass 2 of a lone step 15.5 -> 18.1 us) --; the
// whitecap's noise term |0.3 n.xz| into the float view of the group's row buffer (the expressions of p2_hs_finish_slopes)
template <int N, int P, int R2>
MW_HD void p2_frame_normals(const P2Args& A, int ab, int step, int tl, const cf (&x)[P], cf* set_s) {
    constexpr int T = FftGeom<N, P>::T;
    const int g = tl / T, u = tl % T, a = ab * R2 + g;
    float* nblk = A.normals + ((size_t)step * N * N + (size_t)ab * R2 * N) * 3;  // block-uniform
    const unsigned noff = (unsigned)((g * N + u) * 3);
    float* nrow = reinterpret_cast<float*>(set_s + g * P2Buf<N, P>::BUFSTRIDE);
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int b = u + T * q;
        float* nq = nblk + (size_t)T * q * 3;
        const float sg = post_sign(a, b);
        const float sx = sg * x[q].x, sz = sg * x[q].y;
        const float inv = mw_rsqrt(__builtin_fmaf(sz, sz, __builtin_fmaf(sx, sx, 1.0f)));
        const float nx = sx * inv, ny = inv, nz = sz * inv;
        mw_store_stream<MW_FRAME_NT_RESULTS != 0>(&nq[noff + 0], nx);
        mw_store_stream<MW_FRAME_NT_RESULTS != 0>(&nq[noff + 1], ny);
        mw_store_stream<MW_FRAME_NT_RESULTS != 0>(&nq[noff + 2], nz);
        const float n0 = smul(fabsf(nx), 0.3f), n1 = smul(fabsf(nz), 0.3f);
        nrow[b] = ssqrt(sadd(smul(n0, n0), smul(n1, n1)));  // :269
    }
}
// after the barrier.  Height groups: the vertices (p2_vertices' expressions; hds from the published row)
template <int N, int P, int R2>
MW_HD void p2_frame_vertices(const P2Args& A, int ab, int step, int tl, const cf (&x)[P], const cf* set_d) {
    constexpr int T = FftGeom<N, P>::T;
    const int g = tl / T, u = tl % T, a = ab * R2 + g;
    const cf* drow = set_d + g * P2Buf<N, P>::BUFSTRIDE;
    float* vblk = A.vertices + ((size_t)step * N * N + (size_t)ab * R2 * N) * 3;          // block-uniform
    const unsigned voff = (unsigned)((g * N + u) * 3);
    const float rx = rest_coord(N, A.c.unit_width, a);
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int b = u + T * q;
        const cf d = drow[b];
        float* vq = vblk + (size_t)T * q * 3;                                              // uniform
        mw_store_stream<MW_FRAME_NT_RESULTS != 0>(&vq[voff + 0], ssub(rx, smul(d.x, A.c.choppiness)));                                // :245
        mw_store_stream<MW_FRAME_NT_RESULTS != 0>(&vq[voff + 1], post_sign(a, b) * x[q].x);                                           // :243
        mw_store_stream<MW_FRAME_NT_RESULTS != 0>(&vq[voff + 2], ssub(rest_coord(N, A.c.unit_width, b), smul(d.y, A.c.choppiness)));  // :244
    }
}
// displacement groups: 1 - J from the published rows (nxt: row a + 1, the halo row's buffer for the block's last row), then the whitecap
// with the slope group's noise term
template <int N, int P, int R2>
MW_HD void p2_frame_white(const P2Args& A, int ab, int step, int tl, const cf* set_d, const cf* halo, const cf* set_s) {
    constexpr int T = FftGeom<N, P>::T, BS = P2Buf<N, P>::BUFSTRIDE;
    const int g = tl / T, u = tl % T, a = ab * R2 + g;
    NbRow nb;
    nb.row = set_d + g * BS;
    const cf* nxt = (g + 1 < R2) ? set_d + (g + 1) * BS : halo;
    const float* nrow = reinterpret_cast<const float*>(set_s + g * BS);
    float* wblk = A.white + ((size_t)step * N * N + (size_t)ab * R2 * N) * A.white_stride;
    const unsigned woff = (unsigned)((g * N + u) * A.white_stride);
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int b = u + T * q;
        P2RowView<P> v;
        v.d[q] = nb.row[b];
        const float omj = p2_one_minus_jacobian<N, P, R2>(a, b, q, v, nxt, nb);
        p2_store_white<MW_FRAME_NT_RESULTS != 0>(wblk + (size_t)T * q * A.white_stride, woff, A.white_stride, omj, nrow[b]);
    }
}

// points per thread / rows per pass-2 block for a given N (MW_PT overrides P for experiments)
#ifndef MW_PT
#define MW_PT 8
#endif
#ifndef MW_PT_OR
#define MW_PT_OR MW_PT
#endif
#ifndef MW_PT1
#define MW_PT1 16  // pass 1: 16 points/thread (one exchange fewer; 125 VGPRs, 4 workgroups of 4 waves per CU) measured 1 % ahead of 8
#endif
#ifndef MW_PT2
#define MW_PT2 MW_PT  // pass 2 (the exchange-buffer layout does not depend on P, so the passes may differ)
#endif
// Pass-2 plans of the large grids (A/B-measured): points per thread, rows per workgroup and whether
// the sequential-halo kernel is used.  The aim from 2048^2 up is TWO resident workgroups per CU (<= 80 KiB of LDS and
// <= 64 VGPRs at 1024 threads), which only 8 points per thread allows.
#ifndef MW_PT2_2048
#define MW_PT2_2048 16
#endif
#ifndef MW_PT2_4096
#define MW_PT2_4096 16
#endif
#ifndef MW_HS_2048
#define MW_HS_2048 1  // 2048^2: sequential halo, 4 rows, 2 virtual threads per lane, two 4-wave workgroups per CU: +10 % over 4 rows + halo group
#endif
#ifndef MW_HS_4096
#define MW_HS_4096 1  // sequential-halo pass 2 (P2Geom<..., true>): 4096^2 +9 % over 2 rows + halo group
#endif
#ifndef MW_R2_2048
#define MW_R2_2048 4
#endif
#ifndef MW_R2_4096
#define MW_R2_4096 4
#endif
#ifndef MW_VT_4096
#define MW_VT_4096 2
#endif
#ifndef MW_VT_2048
#define MW_VT_2048 2
#endif
#ifndef MW_R2_1024
#define MW_R2_1024 4
#endif
// 1024^2 (A/B at steady clocks, pass 2 per 32 steps): 4 rows + halo group at 8 points per thread 336 us; sequential halo at
// 16 points per thread, one wave per row, 325-327 us either as 4-wave workgroups or as 2-wave workgroups of 2 virtual
// threads per lane (kept: with the early halo fetch it reads 21.3 instead of 24.2 B per grid point)
#ifndef MW_PT2_1024
#define MW_PT2_1024 16
#endif
#ifndef MW_HS_1024
#define MW_HS_1024 1
#endif
#ifndef MW_VT_1024
#define MW_VT_1024 2
#endif
#ifndef MW_VT1_4096
#define MW_VT1_4096 1
#endif
// software prefetch level of k_pass2_hs (0 none, 1 displacement rows during the height field, 2 + slope rows during displacement)
#ifndef MW_PF_4096
#define MW_PF_4096 0  // displacement rows prefetched during the height field: -1.5 % at 4096^2 in round 2; with KeepT1 (round 4) the plan
#endif                // without it is 1 % ahead (265.0 vs 267.9 us per step), and 2048^2 / 1024^2 were always slower with it
#ifndef MW_PF_2048
#define MW_PF_2048 0
#endif
#ifndef MW_PF_1024
#define MW_PF_1024 0
#endif
#ifndef MW_VT1_2048
#define MW_VT1_2048 1
#endif
#ifndef MW_R2_SMALL_N
#define MW_R2_SMALL_N 256  // grids up to this size use 8 rows + halo per pass-2 workgroup (256^2: +3.5 %; 512^2: -3 %)
#endif
template <int N> struct Plan {
    static constexpr int P = (N >= 2048) ? 16 : MW_PT_OR;  // OceanRenderer passes
    static constexpr int P1 = (N >= 2048) ? 16 : MW_PT1;  // 5 x N/8 threads would exceed 1024 at N = 2048
    static constexpr int P2 = (N >= 4096) ? MW_PT2_4096 : (N == 2048 ? MW_PT2_2048 : (N == 1024 ? MW_PT2_1024 : MW_PT2));
    static constexpr bool HS = (N >= 4096) ? (MW_HS_4096 != 0) : (N == 2048 ? (MW_HS_2048 != 0) : (N == 1024 ? (MW_HS_1024 != 0) : false));
    static constexpr int R2 = (N >= 4096) ? MW_R2_4096 : (N == 2048 ? MW_R2_2048 : (N == 1024 ? MW_R2_1024 : ((N <= MW_R2_SMALL_N) ? 8 : 4)));
    // virtual threads per lane of the sequential-halo kernel (k_pass2_hs): 2 = 8 fat waves with a 256-VGPR budget
    static constexpr int VT = (N >= 4096) ? MW_VT_4096 : (N == 2048 ? MW_VT_2048 : MW_VT_1024);
    static constexpr int VT1 = (N >= 4096) ? MW_VT1_4096 : (N == 2048 ? MW_VT1_2048 : 1);  // the same for pass 1
    static constexpr int PF = (N >= 4096) ? MW_PF_4096 : (N == 2048 ? MW_PF_2048 : MW_PF_1024);
};

}  // namespace mw
