// ocean_renderer_kernels.h -- MW_SEM_OCEANRENDERER: the reference's fragment-shader pipeline ("B", SURVEY.md 8a
// b1-b13) as three kernels instead of 1 + 45 full-screen blits per frame:
//   k_or_pass1   Dispersion + Spectrum + SpectrumHeight passes fused with the transform along py; grid (M/4, 3 fields):
//                a frame is ONE 1024^2 texture, so the launch is latency- not bandwidth-bound and the three fields of
//                a column job run as three concurrent blocks (each recomputes the cheap h~, one of them stores the phase)
//                (F/Dispersion.shader:32-41, F/Spectrum.shader:34-51, F/SpectrumHeight.shader:34-47, F/Stockham.shader)
//   k_or_pass2   transform along px, writes height.r / displacement.rgb      (S/OceanRenderer.cs:229-298)
//   k_or_normal_white   F/OceanNormal.shader:39-56 + F/WhiteCap.shader:33-45 (need +-1 / +-8 texel neighbours)
// The Stockham pass schedule of S/OceanRenderer.cs:229-262 is a forward, unnormalised, natural-order DFT
// (first probe), done here by the same LDS-staged radix-P passes as the FFTMesh path with
// SGN = -1.  The 2-D DFT is separable, so the column direction is transformed first (from a transposed copy of the
// initial spectrum and phase) to make the final stores contiguous; results differ from rows-first only by rounding.
//
// Departures from the shader text, all documented in oracle/ocean_renderer_oracle.c: the sine-hash RNG is replaced by
// the library's counter RNG; HLSL fmod = exact IEEE remainder; border addressing = clamp.
#pragma once
#include "fftmesh_kernels.h"

namespace mw {

struct OrConsts {
    int M;  // texture size = 8 * resolution (S/OceanRenderer.cs:136)
    float length, gravity, choppiness;
    float normal_length;  // normalMat's _Length: set once in SetParams (S/OceanRenderer.cs:163), NOT updated when `length` changes
};

// F/FFTCommon.cginc:58-67 GetWave component for texel index p (n = p + 0.5 in the shader, minus 0.5 again)
MW_HD float or_wave(int M, float length, int p) {
    const float n = (float)((p < M / 2) ? p : p - M);  // (n < res*0.5) ? n : n - res, n integral here
    return sdiv(smul(smul(2.0f, MW_PI_F), n), length);
}
// F/FFTCommon.cginc:69-85 Phillips (damping 0.01); amp = amplitude / 10000 (S/OceanRenderer.cs:149)
MW_HD float or_phillips(int M, float length, float wind_x, float wind_y, float amp, float gravity, float kx, float kz) {
    float klen = sqrtf(kx * kx + kz * kz);
    float klen2 = klen * klen, klen4 = klen2 * klen2;
    if (klen < MW_EPS_F) return 0.f;
    float wlen = sqrtf(wind_x * wind_x + wind_y * wind_y);
    float kDotW = (kx / klen) * (wind_x / wlen) + (kz / klen) * (wind_y / wlen);
    float l = wlen * wlen / gravity, l2 = l * l;
    float damping = 0.01f, L2 = l2 * damping * damping;
    return amp * expf(-1.f / (klen2 * l2)) / klen4 * (kDotW * kDotW) * expf(-klen2 * L2);
}
// F/InitialSpectrum.shader:42-54 for texel (px,py); written TRANSPOSED: initT[px*M + py]
MW_HD void or_init_element(int M, float length, float wind_x, float wind_y, float amp, float gravity, uint64_t seed,
                           int px, int py, f4* initT, float* phaseT) {
    const uint64_t idx = (uint64_t)py * M + px;  // RNG counter follows the texel order of the reference
    const float phi1 = or_phillips(M, length, wind_x, wind_y, amp, gravity, or_wave(M, length, px), or_wave(M, length, py));
    // Phillips(_Resolution - n, _Resolution - m): texel M-1-p, the off-by-one mirror (:48)
    const float phi2 = or_phillips(M, length, wind_x, wind_y, amp, gravity, or_wave(M, length, M - 1 - px),
                                   or_wave(M, length, M - 1 - py));
    float o[4];
    for (int d = 0; d < 2; d++) {
        float r1 = uniform01(seed, 4 * idx + 2 * d), r2 = uniform01(seed, 4 * idx + 2 * d + 1);
        r1 = r1 < 0.01f ? 0.01f : (r1 > 1.f ? 1.f : r1);  // F/FFTCommon.cginc:92-93
        r2 = r2 < 0.01f ? 0.01f : (r2 > 1.f ? 1.f : r2);
        const float x = sqrtf(-2.f * logf(r1));
        float s, c;
        mw_sincos(2.0f * MW_PI_F * r2, &s, &c);
        const float sc = sqrtf((d ? phi2 : phi1) / 2.f);
        o[2 * d] = x * c * sc;
        o[2 * d + 1] = x * s * sc;
    }
    f4 v;
    v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = -o[3];  // Conj (:51)
    initT[(size_t)px * M + py] = v;
    if (phaseT) phaseT[(size_t)px * M + py] = 0.f;  // the phase render targets start at 0; RenderInitial() alone (NULL) keeps them
}
// F/FFTCommon.cginc:101-114: phase <- fmod(phase + sqrt(G |k| (1 + |k|^2/370^2)) dt, 2 pi), strict float32.
// The angular frequency does not depend on time: it is tabulated once (or_omega, 4 strict divisions + 2 square roots per
// texel) and the per-frame update is one multiply, one add and the remainder -- the same operation sequence, bit for bit.
MW_HD float or_omega(const OrConsts& c, int px, int py) {
    const float kx = or_wave(c.M, c.length, px), kz = or_wave(c.M, c.length, py);
    const float wlen = ssqrt(sadd(smul(kx, kx), smul(kz, kz)));
    const float q = sdiv(sdiv(smul(wlen, wlen), 370.f), 370.f);
    const float inner = smul(smul(c.gravity, wlen), sadd(1.f, q));
    return ssqrt(inner);
}
MW_HD float or_phase_step(float omega, float old_phase, float dt) {
    const float x = sadd(old_phase, smul(omega, dt)), twopi = smul(2.0f, MW_PI_F);
    // fmod(x, 2 pi) is x on [0, 2 pi) and exactly x - 2 pi on [2 pi, 4 pi) (Sterbenz); anything else takes the library path
    if (x >= 0.f && x < twopi) return x;
    if (x >= twopi && x < sadd(twopi, twopi)) return ssub(x, twopi);
    return fmodf(x, twopi);
}
MW_HD float or_phase_advance(const OrConsts& c, int px, int py, float old_phase, float dt) {
    return or_phase_step(or_omega(c, px, py), old_phase, dt);
}

// ---------------------------------------------------------------------------------------------------------
struct OrP1Args {
    const f4* PQT;    // [px][py] (P, Q): the Hermitian parts of (h0, conj h0'), packed plan only (or_prep_element)
    const f4* initT;  // [px][py] (h0, conj h0')
    const float* omT;       // [px][py] or_omega table
    const float* phase_in;  // [px][py] stateful phase of the previous frame
    float* phase_out;       // [px][py] advanced phase (ping-pong like the reference's two R32F targets, S/OceanRenderer.cs:221)
    const cf* TW;     // forward (SGN = -1) twiddle tables, TwGeom layout
    cf* E;            // [3][M/4][M][4]
    OrConsts c;
    float dt;         // deltaTime * mult (S/OceanRenderer.cs:223)
    int stream_E;     // non-temporal exchange stores (batched handles)
};
template <int N, int P>
struct OrP1Geom {
    static constexpr int T = FftGeom<N, P>::T;
    static constexpr int NTHREADS = 4 * T;
    static constexpr int BUFSTRIDE = FftGeom<N, P>::LBUF + (XLay<N, P>::EXACT ? XLay<N, P>::PAD_RD4 : 4);
    static constexpr int TW_LDS = (TwGeom<N, P>::LDS_CF + 1) & ~1;
    static constexpr int LDS_BYTES = (TW_LDS + 4 * BUFSTRIDE) * (int)sizeof(cf);
};
// Dispersion + h~ for the 4 texel columns px = 4 jb .. 4 jb + 3 (transform index = py = u + T q)
// Every field's block recomputes the advanced phase from phase_in; only the field-0 block stores it (write_phase).
#ifndef MW_OR_P1_CHUNK
#define MW_OR_P1_CHUNK 8  // points whose loads the bandwidth-bound launch forms request together (1: point by point, A/B)
#endif
// CH: points whose loads are requested together
template <int N, int P, int CH_>
MW_HD void or_p1_animate(const OrP1Args& A, int jb, int tid, bool write_phase, cf (&h)[P]) {
    constexpr int T = FftGeom<N, P>::T;
    const int w = wave_uniform<(T % 64) == 0>(tid / T), u = tid % T, px = 4 * jb + w;
    // column bases stay uniform (SGPRs when columns are whole waves), element py = u + T q is base + T q + a 32-bit lane offset
    const size_t col = (size_t)px * N;
    const float* const c_om = A.omT + col;
    const float* const c_pi = A.phase_in + col;
    float* const c_po = A.phase_out + col;
    const f4* const c_in = A.initT + col;
    const unsigned uo = (unsigned)u;
    // CH = 8: every request of eight points first, then the arithmetic and the phase stores (round 5).  Written point by point (CH = 1) -- load
    // omega, phase and spectrum, advance, STORE the phase, next point -- the store to phase_out, which may alias the next point's loads as far
    // as the compiler knows, keeps every load behind the previous point's store: `L L L s_waitcnt S` x 8 in the ISA.  Measured
    // (profiles/r05_ab_notes.md): together they are faster where the launch is bandwidth-bound (4 tiles per frame 102.6 -> 98.5 us, 2048^2
    // textures 129 -> 126) and SLOWER in a lone 1024^2 frame (35.9 -> 38.0 us: its 771 workgroups start at the same instant, and the three
    // field workgroups of a column job then miss L2 together where point by point they drift apart and two of them hit): the kernel picks
    // by the launch form (gridDim.y).
    constexpr int CH = P < CH_ ? P : CH_;
#pragma unroll
    for (int q0 = 0; q0 < P; q0 += CH) {
        float om[CH], pi[CH];
        f4 v[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
            om[k] = (c_om + T * (q0 + k))[uo];
            pi[k] = (c_pi + T * (q0 + k))[uo];
            v[k] = (c_in + T * (q0 + k))[uo];
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const float ph = or_phase_step(om[k], pi[k], A.dt);
            if (write_phase) (c_po + T * (q0 + k))[uo] = ph;
            float s, c;
            mw_sincos(ph, &s, &c);
            h[q0 + k] = animate(v[k].x, v[k].y, v[k].z, v[k].w, c, s);  // h0*pv + h0conj*Conj(pv), F/Spectrum.shader:45
        }
    }
}
// ---- nframes consecutive GenerateTexture() calls of ONE ocean in one enqueue (mw_ocean_generate_texture_steps_device) ----
// The recurrence of F/FFTCommon.cginc:101-104 chains the PHASE of a texel from frame to frame -- one multiply, one add and a
// remainder -- and nothing else: the three transforms of frame k + 1 do not read anything frame k produced.  A workgroup keeps the
// phases of its 4 columns in registers, advances them frame by frame with the same strict-float32 or_phase_step (so the chain
// is, bit for bit, the one nframes single calls walk) and emits every frame's three spectra; the final phase is stored once.
// Frames are cut into groups of `group` consecutive frames (blockIdx.z): the workgroup of group g first walks the chain over the
// g * group frames before its own (phase steps only: ~6 VALU per texel and frame, against ~500 for a frame's spectra and transforms).
#define MW_OR_MAX_FRAMES 32
struct OrP1StepsArgs {
    OrP1Args a;                    // a.dt unused; a.E = [nframes][3][M/4][M][4]
    float dt[MW_OR_MAX_FRAMES];    // deltaTime * mult of every frame (S/OceanRenderer.cs:223)
    int nframes, group;
};
// the phases and angular frequencies of this thread's P points (columns px = 4 jb + w, rows py = u + T q)
template <int N, int P>
MW_HD void or_p1_steps_begin(const OrP1Args& A, int jb, int tid, float (&om)[P], float (&ph)[P]) {
    constexpr int T = FftGeom<N, P>::T;
    const int w = wave_uniform<(T % 64) == 0>(tid / T), u = tid % T, px = 4 * jb + w;
    const size_t col = (size_t)px * N;
    const float* const c_om = A.omT + col;
    const float* const c_pi = A.phase_in + col;
    const unsigned uo = (unsigned)u;
#pragma unroll
    for (int q = 0; q < P; q++) {
        om[q] = (c_om + T * q)[uo];
        ph[q] = (c_pi + T * q)[uo];
    }
}
template <int N, int P>
MW_HD void or_p1_steps_spectrum(const OrP1Args& A, int jb, int tid, f4 (&v)[P]) {
    constexpr int T = FftGeom<N, P>::T;
    const int w = wave_uniform<(T % 64) == 0>(tid / T), u = tid % T, px = 4 * jb + w;
    const f4* const c_in = A.initT + (size_t)px * N;
    const unsigned uo = (unsigned)u;
#pragma unroll
    for (int q = 0; q < P; q++) v[q] = (c_in + T * q)[uo];
}
template <int P>
MW_HD void or_p1_steps_advance(const float (&om)[P], float (&ph)[P], float dt) {
#pragma unroll
    for (int q = 0; q < P; q++) ph[q] = or_phase_step(om[q], ph[q], dt);
}
template <int P>
MW_HD void or_p1_steps_animate(const f4 (&v)[P], const float (&ph)[P], cf (&h)[P]) {
#pragma unroll
    for (int q = 0; q < P; q++) {
        float s, c;
        mw_sincos(ph[q], &s, &c);
        h[q] = animate(v[q].x, v[q].y, v[q].z, v[q].w, c, s);  // F/Spectrum.shader:45, as or_p1_animate
    }
}
template <int N, int P>
MW_HD void or_p1_steps_store_phase(const OrP1Args& A, int jb, int tid, const float (&ph)[P]) {
    constexpr int T = FftGeom<N, P>::T;
    const int w = wave_uniform<(T % 64) == 0>(tid / T), u = tid % T, px = 4 * jb + w;
    float* const c_po = A.phase_out + (size_t)px * N;
    const unsigned uo = (unsigned)u;
#pragma unroll
    for (int q = 0; q < P; q++) (c_po + T * q)[uo] = ph[q];
}

// the real multiplier of a displacement field at one texel: chop * k_c / max(1e-4, |k|) (F/Spectrum.shader:47-49).  One function, its rounding
// written out (one FMA under the root), for every kernel that forms it -- the steps kernel computes it ONCE per workgroup, the others per frame
MW_HD float or_gain(float kc, float kx, float kz, float chop) {
    const float wl = fmaxf(0.0001f, sqrtf(__builtin_fmaf(kx, kx, smul(kz, kz))));
    return smul(sdiv(kc, wl), chop);
}
// f = 0: h (height);  f = 1: hx = -i h kx/w chop;  f = 2: hz   (F/Spectrum.shader:47-49)
template <int N, int P>
MW_HD void or_p1_build(const OrP1Args& A, int jb, int tid, int f, const cf (&h)[P], cf (&x)[P]) {
    constexpr int T = FftGeom<N, P>::T;
    const int w = tid / T, u = tid % T, px = 4 * jb + w;
    const float kx = or_wave(N, A.c.length, px);
#pragma unroll
    for (int q = 0; q < P; q++) {
        if (f == 0) { x[q] = h[q]; continue; }
        const float kz = or_wave(N, A.c.length, u + T * q);
        const float g = or_gain((f == 1) ? kx : kz, kx, kz, A.c.choppiness);  // :47
        // -MultByI(h * k/w) * chop.  Rounded products (smul): the first butterfly adds these values, and a kernel whose field index is a
        // compile-time constant would otherwise fuse product and sum into an FMA where the one-field-per-workgroup form (a select in
        // between) cannot -- the frames of a steps call must equal single calls bit for bit
        x[q] = mk(smul(h[q].y, g), smul(-h[q].x, g));
    }
}
// ---- the packed plan: TWO complex transforms per frame instead of three --------------------------------------------------------------
// A consumer of the planar textures needs four real planes: height.r = Re F(h), displacement.r / .g = Re / Im F(hx) (OceanNormal's
// `center = D.rgb` reads the imaginary part of Dx, F/OceanNormal.shader:44) and displacement.b = Re F(hz).  Re F(A) is the transform of
// the Hermitian part A_h[p] = (A[p] + conj A[m(p)]) / 2, m(p) = (M - p) mod M per axis, so the two real-part-only fields share ONE transform:
//     G = h_h + i (hz)_h        F(G) = Re F(h) + i Re F(hz)
// With the phase texture symmetric under m (it starts at 0 and omega(m(p)) == omega(p) bit for bit, so every frame keeps it so; a
// phase injected with mw_ocean_set_phase is checked), h_h = P e^{i phi} + Q e^{-i phi} with P = (h0 + conj h0'[m]) / 2, Q = (h0' + conj
// h0[m]) / 2 fixed per spectrum (or_prep_element; the FFTMesh path's (P, Q) packing), and hz = -i g_z h with
// g_z = chop kz / |k| odd under m gives (hz)_h = -i g_z h_h -- except on the Nyquist row py = M/2, its own mirror, where g_z is EVEN and
// (hz)_h = -i g_z (h - h_h).  Hence G = h_h + g_z (py == M/2 ? h - h_h : h_h); hx keeps its full transform.  The exchange buffer carries
// 16 instead of 24 B per texel and frame in each direction and a third of the butterflies is gone.  The RGBA form (Im h, Im Dz are
// channels of the reference's render targets) and a handle whose phase is not symmetric run the three-transform plan.
MW_HD void or_prep_element(int M, int px, int py, const f4* initT, f4* PQT) {
    const int mx = (M - px) % M, my = (M - py) % M;
    const f4 a = initT[(size_t)px * M + py], b = initT[(size_t)mx * M + my];  // a = (h0, h0c) here, b = the same at the mirror texel
    f4 o;
    o.x = smul(0.5f, sadd(a.x, b.z)); o.y = smul(0.5f, ssub(a.y, b.w));       // P = (h0 + conj h0c[m]) / 2
    o.z = smul(0.5f, sadd(a.z, b.x)); o.w = smul(0.5f, ssub(a.w, b.y));       // Q = (h0c + conj h0[m]) / 2
    PQT[(size_t)px * M + py] = o;
}
// field 0: X = hx = -i h kx/w chop (as or_p1_build's f = 1);  field 1: G (above).  h = the animated spectrum, hh = its Hermitian part
template <int N, int P>
MW_HD void or_p1_build_packed(const OrP1Args& A, int jb, int tid, int f, const cf (&h)[P], const cf (&hh)[P], cf (&x)[P]) {
    constexpr int T = FftGeom<N, P>::T;
    const int w = tid / T, u = tid % T, px = 4 * jb + w;
    const float kx = or_wave(N, A.c.length, px);
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int py = u + T * q;
        const float kz = or_wave(N, A.c.length, py);
        const float g = or_gain((f == 0) ? kx : kz, kx, kz, A.c.choppiness);
        if (f == 0) { x[q] = mk(smul(h[q].y, g), smul(-h[q].x, g)); continue; }
        const bool nyq = (py == N / 2);
        const float tx = nyq ? ssub(h[q].x, hh[q].x) : hh[q].x, ty = nyq ? ssub(h[q].y, hh[q].y) : hh[q].y;
        x[q] = mk(sadd(hh[q].x, smul(g, tx)), sadd(hh[q].y, smul(g, ty)));
    }
}
// h and its Hermitian part for the lone-frame / tile forms: the phase advance of or_p1_animate, then both animated spectra
template <int N, int P, int CH_>
MW_HD void or_p1_animate_packed(const OrP1Args& A, int jb, int tid, bool write_phase, bool want_h, bool want_hh, cf (&h)[P], cf (&hh)[P]) {
    constexpr int T = FftGeom<N, P>::T;
    const int w = wave_uniform<(T % 64) == 0>(tid / T), u = tid % T, px = 4 * jb + w;
    const size_t col = (size_t)px * N;
    const float* const c_om = A.omT + col;
    const float* const c_pi = A.phase_in + col;
    float* const c_po = A.phase_out + col;
    const f4* const c_in = A.initT + col;
    const f4* const c_pq = A.PQT + col;
    const unsigned uo = (unsigned)u;
    constexpr int CH = P < CH_ ? P : CH_;
#pragma unroll
    for (int q0 = 0; q0 < P; q0 += CH) {
        float om[CH], pi[CH];
        f4 v[CH], pq[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
            om[k] = (c_om + T * (q0 + k))[uo];
            pi[k] = (c_pi + T * (q0 + k))[uo];
            // h itself: everywhere for X, on the Nyquist row (slot P/2 of the threads u == 0; loaded by every thread of that slot) for G
            if (want_h || (q0 + k) == P / 2) v[k] = (c_in + T * (q0 + k))[uo];
            if (want_hh) pq[k] = (c_pq + T * (q0 + k))[uo];
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const float ph = or_phase_step(om[k], pi[k], A.dt);
            if (write_phase) (c_po + T * (q0 + k))[uo] = ph;
            float s, c;
            mw_sincos(ph, &s, &c);
            if (want_h || (q0 + k) == P / 2) h[q0 + k] = animate(v[k].x, v[k].y, v[k].z, v[k].w, c, s);
            if (want_hh) hh[q0 + k] = animate(pq[k].x, pq[k].y, pq[k].z, pq[k].w, c, s);
        }
    }
}
// The steps kernel of the packed plan runs ONE field per workgroup (blockIdx.y): field 0 keeps (h0, h0c) of its points in registers, field 1
// (P, Q) and, for the Nyquist row, (h0, h0c) of slot P/2 -- both coefficient sets in one workgroup took it to 160 registers and three waves per SIMD
// (spectrum kernel 198 -> 230 us per 32 frames with a third less to transform).  a = the field's own animated spectrum (h for field 0, its
// Hermitian part for field 1), hn = h at slot P/2 (field 1); the arithmetic per element is or_p1_build_packed's.
template <int N, int P>
MW_HD void or_p1_steps_coeff(const OrP1Args& A, int jb, int tid, int f, f4 (&c)[P], f4& vn) {
    constexpr int T = FftGeom<N, P>::T;
    const int w = wave_uniform<(T % 64) == 0>(tid / T), u = tid % T, px = 4 * jb + w;
    const f4* const c_c = (f == 0 ? A.initT : A.PQT) + (size_t)px * N;
    const unsigned uo = (unsigned)u;
#pragma unroll
    for (int q = 0; q < P; q++) c[q] = (c_c + T * q)[uo];
    vn = (A.initT + (size_t)px * N + T * (P / 2))[uo];
}
// the field's multipliers of this thread's P points: they do not depend on the frame (two divisions and a root per point -- as many
// instructions as the point's share of the transform when formed per frame: 280 of a frame's ~860 VALU per thread)
template <int N, int P>
MW_HD void or_p1_steps_gain(const OrP1Args& A, int jb, int tid, int f, float (&g)[P]) {
    constexpr int T = FftGeom<N, P>::T;
    const int w = tid / T, u = tid % T, px = 4 * jb + w;
    const float kx = or_wave(N, A.c.length, px);
#pragma unroll
    for (int q = 0; q < P; q++) {
        const float kz = or_wave(N, A.c.length, u + T * q);
        g[q] = or_gain((f == 0) ? kx : kz, kx, kz, A.c.choppiness);
    }
}
template <int N, int P>
MW_HD void or_p1_steps_build_split(int tid, int f, const float (&g_)[P], const f4 (&c)[P], const f4& vn, const float (&ph)[P], cf (&x)[P]) {
    constexpr int T = FftGeom<N, P>::T;
    const int u = tid % T;
#pragma unroll
    for (int q = 0; q < P; q++) {
        float sn, cs;
        mw_sincos(ph[q], &sn, &cs);
        const cf a = animate(c[q].x, c[q].y, c[q].z, c[q].w, cs, sn);
        const int py = u + T * q;
        const float g = g_[q];
        if (f == 0) { x[q] = mk(smul(a.y, g), smul(-a.x, g)); continue; }
        float tx = a.x, ty = a.y;
        if (q == P / 2) {  // only slot P/2 can hold the Nyquist row (py = N/2 <=> u = 0): h there, from its own coefficients
            const cf hn = animate(vn.x, vn.y, vn.z, vn.w, cs, sn);
            const bool nyq = (py == N / 2);
            tx = nyq ? ssub(hn.x, a.x) : a.x;
            ty = nyq ? ssub(hn.y, a.y) : a.y;
        }
        x[q] = mk(sadd(a.x, smul(g, tx)), sadd(a.y, smul(g, ty)));
    }
}

// f = the field's plane in the exchange buffer (three planes per frame, two in the packed plan: the caller advances A.E per frame)
template <int N, int P>
MW_HD void or_p1_finish(const OrP1Args& A, const Twiddles& tw, int jb, int tid, int f, cf (&x)[P], const cf* lds) {
    constexpr int T = FftGeom<N, P>::T;
    const int w2 = tid & 3, u2 = tid >> 2;
    load_last<N, P>(x, u2, lds + w2 * OrP1Geom<N, P>::BUFSTRIDE);
    final_stage<N, P, -1>(x, u2, tw.TF);
    cf* Ef = A.E + (size_t)f * N * N + (size_t)jb * N * 4;
    // element (u2 + T q, w2) of the block's 4-column slab = Ef[(u2 + T q) * 4 + w2] = (Ef + 4 T q)[tid]: block-uniform base + lane offset
    const unsigned to = (unsigned)tid;
    if (A.stream_E) {  // several tiles or one big one: the exchange buffers exceed the caches -- write-once stream
#pragma unroll
        for (int q = 0; q < P; q++) mw_store_stream<true>(&(Ef + (size_t)4 * T * q)[to], x[q]);
    } else {           // one 1024^2 texture: 24 MB, pass 2 finds most of it in L2 / the Infinity Cache
#pragma unroll
        for (int q = 0; q < P; q++) (Ef + (size_t)4 * T * q)[to] = x[q];
    }
}

struct OrP2Args {
    const cf* E;
    const cf* TW;
    float* height;  // [py][px]          heightTexture.r
    cf* disp;       // [py][px] (r, b)   displacementTexture.rb
    float* disp_g;  // [py][px]          displacementTexture.g  (read by OceanNormal's `center`, :44)
    float* height_g;  // [py][px] heightTexture.g = Im h, or NULL   (only the RGBA texture layout needs these two)
    float* disp_a;    // [py][px] displacementTexture.a = Im Dz, or NULL
    OrConsts c;
};
template <int N, int P>
struct OrP2Geom {
    static constexpr int T = FftGeom<N, P>::T;
    static constexpr int R2 = 4;
    static constexpr int NTHREADS = R2 * T;
    static constexpr int BUFSTRIDE = FftGeom<N, P>::LBUF + (XLay<N, P>::EXACT ? XLay<N, P>::PAD_WR4 : 4);
    static constexpr int TW_LDS = (TwGeom<N, P>::LDS_CF + 1) & ~1;
    static constexpr int LDS_BYTES = (TW_LDS + R2 * BUFSTRIDE) * (int)sizeof(cf);
};
MW_HD int or_p2_field(int k) { return k == 0 ? 1 : (k == 1 ? 2 : 0); }  // hx, hz, h
template <int N, int P>
MW_HD void or_p2_load(const OrP2Args& A, int ab, int tid, int f, cf (&x)[P], cf* lds) {  // f = plane of the exchange buffer
    constexpr int T = FftGeom<N, P>::T;
    const int r1 = tid & 3, u1 = tid >> 2, row = ab * 4 + r1;
    const cf* Ef = A.E + (size_t)f * N * N;
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int j = u1 + T * q;
        x[q] = Ef[((size_t)(j >> 2) * N + row) * 4 + (j & 3)];
    }
    stage0_store<N, P, -1>(x, u1, lds + r1 * OrP2Geom<N, P>::BUFSTRIDE);
}
template <int N, int P>
MW_HD void or_p2_finish(const OrP2Args& A, const Twiddles& tw, int ab, int tid, int f, cf (&x)[P], float (&dx)[P],
                        const cf* lds) {
    constexpr int T = FftGeom<N, P>::T;
    // the row of a wave is wave-uniform when rows are whole waves: row bases stay in SGPRs and every store is base + 32-bit lane offset
    // (per-element 64-bit addresses of five arrays, hoisted out of the field loop, took k_or_pass2<2048> to 228 VGPRs and <4096> past its 128)
    const int g = wave_uniform<(T % 64) == 0>(tid / T), u = tid % T, a = ab * 4 + g;  // a = py', b = px'
    load_last<N, P>(x, u, lds + g * OrP2Geom<N, P>::BUFSTRIDE);
    final_stage<N, P, -1>(x, u, tw.TF);
    const size_t rowoff = (size_t)a * N;
    float* const r_dg = A.disp_g + rowoff;
    cf* const r_d = A.disp + rowoff;
    float* const r_da = A.disp_a ? A.disp_a + rowoff : nullptr;
    float* const r_h = A.height + rowoff;
    float* const r_hg = A.height_g ? A.height_g + rowoff : nullptr;
    const unsigned uo = (unsigned)u;
#pragma unroll
    for (int q = 0; q < P; q++) {
        if (f == 1) { dx[q] = x[q].x; (r_dg + T * q)[uo] = x[q].y; }
        else if (f == 2) { (r_d + T * q)[uo] = mk(dx[q], x[q].x); if (r_da) (r_da + T * q)[uo] = x[q].y; }
        else if (f == 3) { (r_d + T * q)[uo] = mk(dx[q], x[q].y); (r_h + T * q)[uo] = x[q].x; }  // packed plan: F(G) = height + i Dz
        // packed plan, one FIELD per workgroup (the lone frame): each writes its half of displacement.rb -- 4-byte pieces 8 bytes apart
        else if (f == 4) { (reinterpret_cast<float*>(r_d) + 2 * T * q)[2 * uo] = x[q].x; (r_dg + T * q)[uo] = x[q].y; }
        else if (f == 5) { (reinterpret_cast<float*>(r_d) + 2 * T * q)[2 * uo + 1] = x[q].y; (r_h + T * q)[uo] = x[q].x; }
        else { (r_h + T * q)[uo] = x[q].x; if (r_hg) (r_hg + T * q)[uo] = x[q].y; }
    }
}

// F/OceanNormal.shader:39-56 and F/WhiteCap.shader:33-45 for one texel; clamp addressing
MW_HD int or_clamp(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
// NT: non-temporal normal/whitecap stores.  A single ocean's textures are consumed right away (renderer,
// mw_ocean_displace_mesh, RGBA packing) and should stay cache-resident: plain stores (NT would be -2 % frame time).  A batched
// handle's textures together exceed the caches anyway: streamed out from MW_OR_STREAM_E_TILES tiles (4 tiles: 0.59 -> 0.61
// of the 120-B figure, 8 tiles: 0.54 -> 0.60).  The height / displacement textures stay plain stores in every case: the
// normal kernel reads them back at once (non-temporal: 103 -> 117 us per 4 frames).
// normal_xz (optional): the normal's x and z as stored, for the whitecap of the same texel (F/WhiteCap.shader:38)
// the arithmetic of the two shaders for one texel, from values already in registers.  c? = D.rgb at the texel (`center`, :44); r/l/t/b =
// (disp.r, height, disp.b) of the right / left / top / bottom neighbour (:45-48)
MW_HD void or_normal_math(float ts, float cx, float cy, float cz, float rx, float rh, float rz, float lx, float lh, float lz, float tx,
                          float th, float tz, float bx, float bh, float bz, float (&n)[3]) {
    const float r0 = ts + rx - cx, r1 = rh - cy, r2 = rz - cz;    // :45
    const float l0 = -ts + lx - cx, l1 = lh - cy, l2 = lz - cz;   // :46
    const float t0 = tx - cx, t1 = th - cy, t2 = -ts + tz - cz;   // :47
    const float b0 = bx - cx, b1 = bh - cy, b2 = ts + bz - cz;    // :48
    // topRight = right x top, topLeft = top x left, bottomLeft = left x bottom, bottomRight = bottom x right
    float nx = (r1 * t2 - r2 * t1) + (t1 * l2 - t2 * l1) + (l1 * b2 - l2 * b1) + (b1 * r2 - b2 * r1);
    float ny = (r2 * t0 - r0 * t2) + (t2 * l0 - t0 * l2) + (l2 * b0 - l0 * b2) + (b2 * r0 - b0 * r2);
    float nz = (r0 * t1 - r1 * t0) + (t0 * l1 - t1 * l0) + (l0 * b1 - l1 * b0) + (b0 * r1 - b1 * r0);
    const float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
    n[0] = nx * inv; n[1] = ny * inv; n[2] = nz * inv;  // :55
}
// ym / yp / xm / xp = displacement.rb at -8 / +8 texels in y and x; (nx, nz) = the normal of the texel (F/WhiteCap.shader:38)
MW_HD float or_white_math(cf ym, cf yp, cf xm, cf xp, float nx, float nz) {
    const float dDdy_x = -0.5f * (ym.x - yp.x) / 8.f, dDdy_y = -0.5f * (ym.y - yp.y) / 8.f;  // :36
    const float dDdx_x = -0.5f * (xm.x - xp.x) / 8.f, dDdx_y = -0.5f * (xm.y - xp.y) / 8.f;  // :37
    const float n0 = 0.3f * nx, n1 = 0.3f * nz;                                                // :38
    const float jac = (1.f + dDdx_x) * (1.f + dDdy_y) - dDdx_y * dDdy_x;                       // :39
    const float turb = fmaxf(0.f, 1.f - jac + sqrtf(n0 * n0 + n1 * n1));                      // :40
    const float t = turb > 1.f ? 1.f : turb;
    return t * t * (3.f - 2.f * t);  // smoothstep(0,1,turb), :43
}
template <bool NT = false>
MW_HD void or_normal_element(const OrConsts& c, int px, int py, const float* height, const cf* disp, const float* disp_g,
                             float* normal, float* normal_xz = nullptr) {
    const int M = c.M;
    const float ts = c.normal_length / (float)M;  // F/OceanNormal.shader:42 with the length of SetParams
    const size_t idx = (size_t)py * M + px;
    const size_t ir = (size_t)py * M + or_clamp(px + 1, M - 1), il = (size_t)py * M + or_clamp(px - 1, M - 1);
    const size_t it = (size_t)or_clamp(py - 1, M - 1) * M + px, ib = (size_t)or_clamp(py + 1, M - 1) * M + px;
    float n[3];
    or_normal_math(ts, disp[idx].x, disp_g[idx], disp[idx].y, disp[ir].x, height[ir], disp[ir].y, disp[il].x, height[il], disp[il].y,
                   disp[it].x, height[it], disp[it].y, disp[ib].x, height[ib], disp[ib].y, n);
    mw_store_stream<NT>(&normal[3 * idx], n[0]);
    mw_store_stream<NT>(&normal[3 * idx + 1], n[1]);
    mw_store_stream<NT>(&normal[3 * idx + 2], n[2]);
    if (normal_xz) { normal_xz[0] = n[0]; normal_xz[1] = n[2]; }
}
// normal_xz: (n.x, n.z) of this texel when the caller has just computed it, else read from `normal`
template <bool NT = false>
MW_HD void or_white_element(const OrConsts& c, int px, int py, const cf* disp, const float* normal, float* white,
                            const float* normal_xz = nullptr) {
    const int M = c.M;
    const size_t idx = (size_t)py * M + px;  // reads the normal of its own texel only, so one thread can do both passes
    // texelSize = 1/_Length with _Length = resolution = M/8 (S/OceanRenderer.cs:306): +-8 texels
    const cf ym = disp[(size_t)or_clamp(py - 8, M - 1) * M + px], yp = disp[(size_t)or_clamp(py + 8, M - 1) * M + px];
    const cf xm = disp[(size_t)py * M + or_clamp(px - 8, M - 1)], xp = disp[(size_t)py * M + or_clamp(px + 8, M - 1)];
    mw_store_stream<NT>(&white[idx], or_white_math(ym, yp, xm, xp, normal_xz ? normal_xz[0] : normal[3 * idx],
                                                   normal_xz ? normal_xz[1] : normal[3 * idx + 2]));
}
// Both passes for the FOUR texels px0 .. px0 + 3 (px0 a multiple of 4) of row py from 16-byte loads: 6 loads and 1 store instruction per
// texel instead of 14 and 4, the normal leaves as three float4 (48 contiguous bytes per thread).  Same arithmetic per texel.
// (or_normal_white_quad_compute: the values; the device kernel turns a wave's normals through LDS before storing them, k_or_normal_white)
MW_HD void or_normal_white_quad_compute(const OrConsts& c, int px0, int py, const float* height, const cf* disp, const float* disp_g,
                                        float (&n)[4][3], float (&w)[4]) {
    const int M = c.M;
    const float ts = c.normal_length / (float)M;
    const size_t rc = (size_t)py * M, rt = (size_t)or_clamp(py - 1, M - 1) * M, rb = (size_t)or_clamp(py + 1, M - 1) * M;
    const size_t r8m = (size_t)or_clamp(py - 8, M - 1) * M, r8p = (size_t)or_clamp(py + 8, M - 1) * M;
    auto ld4 = [](const void* p) { return *reinterpret_cast<const f4*>(p); };
    cf dc[6], dt[4], db[4], dym[4], dyp[4], dxm[4], dxp[4];
    float hc[6], ht[4], hb[4], gc[4];
    auto split = [&](const cf* p, cf* o) {  // 4 consecutive cf = 2 x 16 bytes
        const f4 a = ld4(p), b = ld4(p + 2);
        o[0] = mk(a.x, a.y); o[1] = mk(a.z, a.w); o[2] = mk(b.x, b.y); o[3] = mk(b.z, b.w);
    };
    auto split1 = [&](const float* p, float* o) { const f4 a = ld4(p); o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; };
    split(disp + rc + px0, dc + 1);
    dc[0] = disp[rc + or_clamp(px0 - 1, M - 1)]; dc[5] = disp[rc + or_clamp(px0 + 4, M - 1)];
    split1(height + rc + px0, hc + 1);
    hc[0] = height[rc + or_clamp(px0 - 1, M - 1)]; hc[5] = height[rc + or_clamp(px0 + 4, M - 1)];
    split1(disp_g + rc + px0, gc);
    split(disp + rt + px0, dt); split1(height + rt + px0, ht);
    split(disp + rb + px0, db); split1(height + rb + px0, hb);
    split(disp + r8m + px0, dym); split(disp + r8p + px0, dyp);
    // +-8 texels along the row: px0 - 8 .. px0 - 5 and px0 + 8 .. px0 + 11; outside the row all four clamp to the edge texel
    split(disp + rc + or_clamp(px0 - 8, M - 4), dxm);
    split(disp + rc + or_clamp(px0 + 8, M - 4), dxp);
    if (px0 < 8) { const cf e = disp[rc]; dxm[0] = dxm[1] = dxm[2] = dxm[3] = e; }
    if (px0 + 8 > M - 4) { const cf e = disp[rc + M - 1]; dxp[0] = dxp[1] = dxp[2] = dxp[3] = e; }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        or_normal_math(ts, dc[k + 1].x, gc[k], dc[k + 1].y, dc[k + 2].x, hc[k + 2], dc[k + 2].y, dc[k].x, hc[k], dc[k].y, dt[k].x, ht[k], dt[k].y,
                       db[k].x, hb[k], db[k].y, n[k]);
        w[k] = or_white_math(dym[k], dyp[k], dxm[k], dxp[k], n[k][0], n[k][2]);
    }
}
// the same with the stores straight from the lane (three float4 of the normal 48 bytes apart): host emulation / reference form
template <bool NT = false>
MW_HD void or_normal_white_quad(const OrConsts& c, int px0, int py, const float* height, const cf* disp, const float* disp_g, float* normal,
                                float* white) {
    float n[4][3], w[4];
    or_normal_white_quad_compute(c, px0, py, height, disp, disp_g, n, w);
    const size_t rc = (size_t)py * c.M;
    f4 o;
    float* np_ = normal + 3 * (rc + px0);
    o.x = n[0][0]; o.y = n[0][1]; o.z = n[0][2]; o.w = n[1][0]; mw_store_stream<NT>(reinterpret_cast<f4*>(np_), o);
    o.x = n[1][1]; o.y = n[1][2]; o.z = n[2][0]; o.w = n[2][1]; mw_store_stream<NT>(reinterpret_cast<f4*>(np_ + 4), o);
    o.x = n[2][2]; o.y = n[3][0]; o.z = n[3][1]; o.w = n[3][2]; mw_store_stream<NT>(reinterpret_cast<f4*>(np_ + 8), o);
    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3]; mw_store_stream<NT>(reinterpret_cast<f4*>(white + rc + px0), o);
}


// ---- consumer-side packing -----------------------------------------------------------------
// The four ARGBFloat render targets exactly as the shaders leave them (S/OceanRenderer.cs:143-146,310-313):
//   heightTexture       = (Re h, Im h, Re h, Im h)     SpectrumHeight returns float4(h, h) (F/SpectrumHeight.shader:46),
//                                                      the Stockham passes transform both complex halves alike
//   displacementTexture = (Re Dx, Im Dx, Re Dz, Im Dz) (F/Spectrum.shader:50, F/Stockham.shader:56)
//   normalTexture       = (n.xyz, 1)                   (F/OceanNormal.shader:55)
//   whiteTexture        = (w, w, w, 1)                 (F/WhiteCap.shader:44)
MW_HD void or_pack_rgba_element(size_t idx, const float* height, const float* height_g, const cf* disp, const float* disp_g,
                                const float* disp_a, const float* normal, const float* white, f4* H, f4* D, f4* Nn, f4* W) {
    f4 v;
    if (H) { v.x = height[idx]; v.y = height_g[idx]; v.z = v.x; v.w = v.y; H[idx] = v; }
    if (D) { v.x = disp[idx].x; v.y = disp_g[idx]; v.z = disp[idx].y; v.w = disp_a[idx]; D[idx] = v; }
    if (Nn) { v.x = normal[3 * idx]; v.y = normal[3 * idx + 1]; v.z = normal[3 * idx + 2]; v.w = 1.f; Nn[idx] = v; }
    if (W) { v.x = v.y = v.z = white[idx]; v.w = 1.f; W[idx] = v; }
}

// tex2Dlod(tex, float4(uv, 0, 0)) on an M x M bilinear, clamp-addressed texture (Unity's RenderTexture defaults): texel
// centres sit at (p + 0.5) / M.  Exact f32 weights, not the 8-bit fixed-point weights of a texture unit.
MW_HD void or_bilinear_axis(int M, float u, int* p0, int* p1, float* w) {
    const float x = u * (float)M - 0.5f, fl = floorf(x);
    const int i0 = (int)fl;
    *w = x - fl;
    *p0 = or_clamp(i0, M - 1);
    *p1 = or_clamp(i0 + 1, M - 1);
}
MW_HD float or_lerp2(float a00, float a10, float a01, float a11, float wx, float wy) {
    const float a0 = a00 + (a10 - a00) * wx, a1 = a01 + (a11 - a01) * wx;
    return a0 + (a1 - a0) * wy;
}
// The ocean material's vertex stage for mesh vertex (i, j) of the res x res grid of S/OceanRenderer.cs:172-207:
//   v.vertex.y += _Height(uv).r / 8;  v.vertex.xz += _Anim(uv).rb / 8      (W/TestOcean.shader:65-66,
//                                                                            W/MistralWaterCommon.cginc:22-23)
//   normal = normalize(_Bump(uv).rgb)  (W/TestOcean.shader:70, identity object-to-world)   color = _White(uv).r  (:72)
MW_HD void or_mesh_vertex(int M, int res, float unit_width, int i, int j, const float* height, const cf* disp,
                          const float* normal, const float* white, float* vert, float* nrm, float* col) {
    const float u = sdiv((float)i, (float)(res - 1)), v = sdiv((float)j, (float)(res - 1));  // S/OceanRenderer.cs:184
    int x0, x1, y0, y1;
    float wx, wy;
    or_bilinear_axis(M, u, &x0, &x1, &wx);
    or_bilinear_axis(M, v, &y0, &y1, &wy);
    const size_t i00 = (size_t)y0 * M + x0, i10 = (size_t)y0 * M + x1, i01 = (size_t)y1 * M + x0, i11 = (size_t)y1 * M + x1;
    const size_t cur = (size_t)i * res + j;
    const float h = or_lerp2(height[i00], height[i10], height[i01], height[i11], wx, wy);
    const float dx = or_lerp2(disp[i00].x, disp[i10].x, disp[i01].x, disp[i11].x, wx, wy);
    const float dz = or_lerp2(disp[i00].y, disp[i10].y, disp[i01].y, disp[i11].y, wx, wy);
    vert[3 * cur] = rest_coord(res, unit_width, i) + dx / 8.f;
    vert[3 * cur + 1] = h / 8.f;
    vert[3 * cur + 2] = rest_coord(res, unit_width, j) + dz / 8.f;
    if (nrm) {
        float n[3];
        for (int k = 0; k < 3; k++)
            n[k] = or_lerp2(normal[3 * i00 + k], normal[3 * i10 + k], normal[3 * i01 + k], normal[3 * i11 + k], wx, wy);
        const float inv = 1.f / sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        nrm[3 * cur] = n[0] * inv; nrm[3 * cur + 1] = n[1] * inv; nrm[3 * cur + 2] = n[2] * inv;
    }
    if (col) col[cur] = or_lerp2(white[i00], white[i10], white[i01], white[i11], wx, wy);
}

}  // namespace mw
