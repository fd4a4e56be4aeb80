// ocean_renderer_kernels.h -- MW_SEM_OCEANRENDERER: the reference's fragment-shader pipeline ("B", SURVEY.md 8a
// b1-b13) as three kernels instead of 1 + 45 full-screen blits per frame:
//   k_or_pass1   Dispersion + Spectrum + SpectrumHeight passes fused with the transform along py
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
    phaseT[(size_t)px * M + py] = 0.f;  // the phase render targets start at 0
}
// F/FFTCommon.cginc:101-114: phase <- fmod(phase + sqrt(G |k| (1 + |k|^2/370^2)) dt, 2 pi), strict float32
MW_HD float or_phase_advance(const OrConsts& c, int px, int py, float old_phase, float dt) {
    const float kx = or_wave(c.M, c.length, px), kz = or_wave(c.M, c.length, py);
    const float wlen = ssqrt(sadd(smul(kx, kx), smul(kz, kz)));
    const float q = sdiv(sdiv(smul(wlen, wlen), 370.f), 370.f);
    const float inner = smul(smul(c.gravity, wlen), sadd(1.f, q));
    const float dphi = smul(ssqrt(inner), dt);
    return fmodf(sadd(old_phase, dphi), smul(2.0f, MW_PI_F));
}

// ---------------------------------------------------------------------------------------------------------
struct OrP1Args {
    const f4* initT;  // [px][py] (h0, conj h0')
    float* phaseT;    // [px][py] stateful phase
    const cf* TW;     // forward (SGN = -1) twiddle tables, TwGeom layout
    cf* E;            // [3][M/4][M][4]
    OrConsts c;
    float dt;         // deltaTime * mult (S/OceanRenderer.cs:223)
};
template <int N, int P>
struct OrP1Geom {
    static constexpr int T = FftGeom<N, P>::T;
    static constexpr int NTHREADS = 4 * T;
    static constexpr int BUFSTRIDE = FftGeom<N, P>::LBUF + 4;
    static constexpr int TW_LDS = TwGeom<N, P>::IN_LDS ? ((TwGeom<N, P>::TOTAL + 1) & ~1) : 0;
    static constexpr int LDS_BYTES = (TW_LDS + 4 * BUFSTRIDE) * (int)sizeof(cf);
};
// Dispersion + h~ for the 4 texel columns px = 4 jb .. 4 jb + 3 (transform index = py = u + T q)
template <int N, int P>
MW_HD void or_p1_animate(const OrP1Args& A, int jb, int tid, cf (&h)[P]) {
    constexpr int T = FftGeom<N, P>::T;
    const int w = tid / T, u = tid % T, px = 4 * jb + w;
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int py = u + T * q;
        const size_t idx = (size_t)px * N + py;
        const float ph = or_phase_advance(A.c, px, py, A.phaseT[idx], A.dt);
        A.phaseT[idx] = ph;
        const f4 v = A.initT[idx];
        float s, c;
        mw_sincos(ph, &s, &c);
        h[q] = animate(v.x, v.y, v.z, v.w, c, s);  // h0*pv + h0conj*Conj(pv), F/Spectrum.shader:45
    }
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
        const float wl = fmaxf(0.0001f, sqrtf(kx * kx + kz * kz));  // :47
        const float g = ((f == 1) ? kx : kz) / wl * A.c.choppiness;
        x[q] = mk(h[q].y * g, -h[q].x * g);  // -MultByI(h * k/w) * chop
    }
}
template <int N, int P>
MW_HD void or_p1_finish(const OrP1Args& A, const Twiddles& tw, int jb, int tid, int f, cf (&x)[P], const cf* lds) {
    constexpr int T = FftGeom<N, P>::T;
    const int w2 = tid & 3, u2 = tid >> 2;
    load_slots<N, P>(x, u2, lds + w2 * OrP1Geom<N, P>::BUFSTRIDE);
    final_stage<N, P, -1>(x, u2, tw.TF);
    cf* Ef = A.E + (size_t)f * N * N + (size_t)jb * N * 4;
#pragma unroll
    for (int q = 0; q < P; q++) Ef[(size_t)(u2 + T * q) * 4 + w2] = x[q];
}

struct OrP2Args {
    const cf* E;
    const cf* TW;
    float* height;  // [py][px]          heightTexture.r
    cf* disp;       // [py][px] (r, b)   displacementTexture.rb
    float* disp_g;  // [py][px]          displacementTexture.g  (read by OceanNormal's `center`, :44)
    OrConsts c;
};
template <int N, int P>
struct OrP2Geom {
    static constexpr int T = FftGeom<N, P>::T;
    static constexpr int R2 = 4;
    static constexpr int NTHREADS = R2 * T;
    static constexpr int BUFSTRIDE = FftGeom<N, P>::LBUF + 4;
    static constexpr int TW_LDS = TwGeom<N, P>::IN_LDS ? ((TwGeom<N, P>::TOTAL + 1) & ~1) : 0;
    static constexpr int LDS_BYTES = (TW_LDS + R2 * BUFSTRIDE) * (int)sizeof(cf);
};
MW_HD int or_p2_field(int k) { return k == 0 ? 1 : (k == 1 ? 2 : 0); }  // hx, hz, h
template <int N, int P>
MW_HD void or_p2_load(const OrP2Args& A, int ab, int tid, int f, cf (&x)[P], cf* lds) {
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
    const int g = tid / T, u = tid % T, a = ab * 4 + g;  // a = py', b = px'
    load_slots<N, P>(x, u, lds + g * OrP2Geom<N, P>::BUFSTRIDE);
    final_stage<N, P, -1>(x, u, tw.TF);
    const size_t rowoff = (size_t)a * N;
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int b = u + T * q;
        if (f == 1) { dx[q] = x[q].x; A.disp_g[rowoff + b] = x[q].y; }
        else if (f == 2) A.disp[rowoff + b] = mk(dx[q], x[q].x);
        else A.height[rowoff + b] = x[q].x;
    }
}

// F/OceanNormal.shader:39-56 and F/WhiteCap.shader:33-45 for one texel; clamp addressing
MW_HD int or_clamp(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
MW_HD void or_normal_element(const OrConsts& c, int px, int py, const float* height, const cf* disp, const float* disp_g,
                             float* normal) {
    const int M = c.M;
    const float ts = c.length / (float)M;
    const size_t idx = (size_t)py * M + px;
    const float cx = disp[idx].x, cy = disp_g[idx], cz = disp[idx].y;  // center = D.rgb (:44)
    const size_t ir = (size_t)py * M + or_clamp(px + 1, M - 1), il = (size_t)py * M + or_clamp(px - 1, M - 1);
    const size_t it = (size_t)or_clamp(py - 1, M - 1) * M + px, ib = (size_t)or_clamp(py + 1, M - 1) * M + px;
    const float r0 = ts + disp[ir].x - cx, r1 = height[ir] - cy, r2 = disp[ir].y - cz;    // :45
    const float l0 = -ts + disp[il].x - cx, l1 = height[il] - cy, l2 = disp[il].y - cz;   // :46
    const float t0 = disp[it].x - cx, t1 = height[it] - cy, t2 = -ts + disp[it].y - cz;   // :47
    const float b0 = disp[ib].x - cx, b1 = height[ib] - cy, b2 = ts + disp[ib].y - cz;    // :48
    // topRight = right x top, topLeft = top x left, bottomLeft = left x bottom, bottomRight = bottom x right
    float nx = (r1 * t2 - r2 * t1) + (t1 * l2 - t2 * l1) + (l1 * b2 - l2 * b1) + (b1 * r2 - b2 * r1);
    float ny = (r2 * t0 - r0 * t2) + (t2 * l0 - t0 * l2) + (l2 * b0 - l0 * b2) + (b2 * r0 - b0 * r2);
    float nz = (r0 * t1 - r1 * t0) + (t0 * l1 - t1 * l0) + (l0 * b1 - l1 * b0) + (b0 * r1 - b1 * r0);
    const float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
    normal[3 * idx] = nx * inv; normal[3 * idx + 1] = ny * inv; normal[3 * idx + 2] = nz * inv;  // :55
}
MW_HD void or_white_element(const OrConsts& c, int px, int py, const cf* disp, const float* normal, float* white) {
    const int M = c.M;
    const size_t idx = (size_t)py * M + px;
    // texelSize = 1/_Length with _Length = resolution = M/8 (S/OceanRenderer.cs:306): +-8 texels
    const cf ym = disp[(size_t)or_clamp(py - 8, M - 1) * M + px], yp = disp[(size_t)or_clamp(py + 8, M - 1) * M + px];
    const cf xm = disp[(size_t)py * M + or_clamp(px - 8, M - 1)], xp = disp[(size_t)py * M + or_clamp(px + 8, M - 1)];
    const float dDdy_x = -0.5f * (ym.x - yp.x) / 8.f, dDdy_y = -0.5f * (ym.y - yp.y) / 8.f;  // :36
    const float dDdx_x = -0.5f * (xm.x - xp.x) / 8.f, dDdx_y = -0.5f * (xm.y - xp.y) / 8.f;  // :37
    const float n0 = 0.3f * normal[3 * idx], n1 = 0.3f * normal[3 * idx + 2];                 // :38
    const float jac = (1.f + dDdx_x) * (1.f + dDdy_y) - dDdx_y * dDdy_x;                     // :39
    const float turb = fmaxf(0.f, 1.f - jac + sqrtf(n0 * n0 + n1 * n1));                    // :40
    const float t = turb > 1.f ? 1.f : turb;
    white[idx] = t * t * (3.f - 2.f * t);                                                    // smoothstep(0,1,turb), :43
}

}  // namespace mw
