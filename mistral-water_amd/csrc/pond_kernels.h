// pond_kernels.h -- the pond material's vertex-stage Displacement() with every mode of the shader library:
//   MW_POND_WAVE                W/MistralWaterLib.cginc:127-152 Wave()  (+ finite-difference normal), applied as :160-166
//   MW_POND_GERSTNER            W/MistralWaterLib.cginc:71-99 Gerstner(), applied as :168-179 (4 waves from the material)
//   MW_POND_GERSTNER_LEVEL_ONE  W/MistralWaterLib.cginc:101-125 GerstnerLevelOne() (5 built-in waves); the reference never
//                               wires it into Displacement(), it is applied here the way the Gerstner branch applies its
//                               offsets (:176-177)
// Object space == world space (unity_ObjectToWorld = identity), which is how the pond plane sits in the shipped scene
// up to a translation.  `half` is f32 on desktop targets.  Streaming kernel: 12 B read + 12 B (position) and optionally
// + 12 B (normal) written per vertex.
#pragma once
#include "../../include/mistral_water.h"
#include "fftmesh_kernels.h"

namespace mw {

struct PondParams {
    int mode;
    float amplitude;   // _Amplitude as set on the material (the x0.01 of :134,:172 is applied here)
    float frequency;   // _Frequency
    float speed;       // _Speed (Wave)
    float steepness;   // _Steepness
    float smoothing;   // _Smoothing (Wave normal)
    float wspeed[4];   // _WSpeed
    float dir_ab[4];   // _WDirectionAB
    float dir_cd[4];   // _WDirectionCD
};

// W/MistralWaterLib.cginc:105-109
MW_HD void level_one_wave(int i, float* amp, float* steep, float* speed, float* dx, float* dy, float* fs) {
    const float amps[5] = {0.7f, 0.6f, 0.6f, 0.7f, 0.9f};
    const float steeps[5] = {0.95f, 0.615f, 0.821f, 0.462f, 0.611f};
    const float speeds[5] = {-2.112f, 0.6124f, -0.878f, -3.6234f, 1.f};
    const float dirx[5] = {1.f, -0.9f, 0.2f, -1.0f, 0.99f};
    const float diry[5] = {-0.2f, 1.f, 0.2f, 0.77f, -1.145f};
    const float fss[5] = {0.954f, 1.52f, 0.44f, 0.21f, 0.8f};
    *amp = amps[i]; *steep = steeps[i]; *speed = speeds[i]; *dx = dirx[i]; *dy = diry[i]; *fs = fss[i];
}

// one vertex of Displacement(): p = object/world position in, displaced position and shader normal out
MW_HD void pond_vertex(const PondParams& P, float t, float px, float py, float pz, float* o, float* n) {
    if (P.mode == MW_POND_WAVE) {
        const float sp = P.speed * t;          // :133
        const float A = P.amplitude * 0.01f;   // :134
        const float f = P.frequency;
        const float a = sp + px * f, b = sp + pz * f;  // phases of v0 (:136,:140)
        const float hd = 0.5f * (0.05f * f);           // half the phase step of the +0.05 neighbours v1, v2 (:130-131)
        float s0, c0, sb, cb, sa2, ca2, sb2, cb2, sh, ch;
        mw_sincos_fast(a, &s0, &c0);
        mw_sincos_fast(b, &sb, &cb);
        mw_sincos_fast(a + hd, &sa2, &ca2);
        mw_sincos_fast(b + hd, &sb2, &cb2);
        mw_sincos_fast(hd, &sh, &ch);
        const float y0 = py + s0 * A - cb * A;
        // v1.y - v0.y = (sin(a + 2hd) - sin a) A = 2 cos(a + hd) sin(hd) A and v2.y - v0.y = -(cos(b + 2hd) - cos b) A =
        // 2 sin(b + hd) sin(hd) A, then scaled by _Smoothing (:144-145).  The product form has no cancellation (the
        // f64 oracle is the judge, not an f32 evaluation of the shader text).
        const float d1 = 2.f * ca2 * sh * A * P.smoothing, d2 = 2.f * sb2 * sh * A * P.smoothing;
        // cross(v2 - v0, v1 - v0) with v2 - v0 = (0, d2, 0.05), v1 - v0 = (0.05, d1, 0)   (:147)
        const float nx = -0.05f * d1, ny = 0.05f * 0.05f, nz = -0.05f * d2;
        const float inv = 1.f / sqrtf(nx * nx + ny * ny + nz * nz);
        o[0] = px;
        o[1] = py + y0;  // v.vertex.y += offsets.y with offsets = v0 (:151,:163): the world y is counted twice, as in the shader
        o[2] = pz;
        n[0] = nx * inv; n[1] = ny * inv; n[2] = nz * inv;
        return;
    }
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (P.mode == MW_POND_GERSTNER) {
        const float A = P.amplitude * 0.01f;  // :172
        const float sa = P.steepness * A;     // :77-78
        const float dx[4] = {P.dir_ab[0], P.dir_ab[2], P.dir_cd[0], P.dir_cd[2]};
        const float dy[4] = {P.dir_ab[1], P.dir_ab[3], P.dir_cd[1], P.dir_cd[3]};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float s, c;
            mw_sincos_fast(P.frequency * (dx[i] * px + dy[i] * pz) + t * P.wspeed[i], &s, &c);  // :80-84
            sx += c * (sa * dx[i]);  // :86
            sz += c * (sa * dy[i]);  // :87
            sy += s * A;             // :88
        }
    } else {  // MW_POND_GERSTNER_LEVEL_ONE, :112-117; amplitude as passed by the caller (no x0.01 inside the function)
#pragma unroll
        for (int i = 0; i < 5; i++) {
            float amp, steep, speed, dx, dy, fs;
            level_one_wave(i, &amp, &steep, &speed, &dx, &dy, &fs);
            float s, c;
            mw_sincos_fast(P.frequency * fs * (px * dx + pz * dy) + speed * P.frequency * fs * t, &s, &c);
            sx += P.steepness * P.amplitude * steep * amp * dx * c;
            sz += P.steepness * P.amplitude * steep * amp * dy * c;
            sy += P.amplitude * amp * s;
        }
    }
    o[0] = px + sx; o[1] = py + sy; o[2] = pz + sz;  // :176
    n[0] = 0.f; n[1] = 1.f; n[2] = 0.f;              // :98, :121
}

#if defined(__HIPCC__)
// 4 vertices (= 3 x float4) per thread: every load/store is a 16-B access, lanes contiguous.  Vertices [0, nvec) take that
// path (nvec = 0 when a buffer is not 16-B aligned, gerstner_kernels.h), [nvec, nverts) a scalar grid-stride loop.
template <bool NORMALS>
__global__ __launch_bounds__(256) void k_pond(const float* __restrict__ pos, float* __restrict__ out, float* __restrict__ nrm,
                                              int64_t nverts, int64_t nvec, PondParams P, float t) {
    const int64_t nquads = nvec >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    __shared__ f4 tile[256 * 3];
    const int lane = threadIdx.x & 63;
    f4* wt = tile + (threadIdx.x - lane) * 3;
    for (int64_t q0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x - lane); q0 < nquads; q0 += stride) {  // wave-uniform
        const int64_t qd = q0 + lane;
        const f4* p = reinterpret_cast<const f4*>(pos) + (qd < nquads ? qd : q0) * 3;  // (idle lanes of the last wave redo its first quad)
        f4 a = p[0], b = p[1], c = p[2];
        float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
        float o[12], n[12];
#pragma unroll
        for (int k = 0; k < 4; k++) pond_vertex(P, t, v[3 * k], v[3 * k + 1], v[3 * k + 2], &o[3 * k], &n[3 * k]);
        const int nf4 = (int)(nquads - q0 < 64 ? nquads - q0 : 64) * 3;
        f4 r0 = {o[0], o[1], o[2], o[3]}, r1 = {o[4], o[5], o[6], o[7]}, r2 = {o[8], o[9], o[10], o[11]};
        wave_store_3f4<false>(wt, lane, r0, r1, r2, reinterpret_cast<f4*>(out) + q0 * 3, nf4);  // gerstner_kernels.h
        if (NORMALS) {
            f4 m0 = {n[0], n[1], n[2], n[3]}, m1 = {n[4], n[5], n[6], n[7]}, m2 = {n[8], n[9], n[10], n[11]};
            wave_store_3f4<false>(wt, lane, m0, m1, m2, reinterpret_cast<f4*>(nrm) + q0 * 3, nf4);
        }
    }
    for (int64_t vtx = nvec + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; vtx < nverts; vtx += stride) {
        float o[3], n[3];
        pond_vertex(P, t, pos[3 * vtx], pos[3 * vtx + 1], pos[3 * vtx + 2], o, n);
        out[3 * vtx] = o[0]; out[3 * vtx + 1] = o[1]; out[3 * vtx + 2] = o[2];
        if (NORMALS) { nrm[3 * vtx] = n[0]; nrm[3 * vtx + 1] = n[1]; nrm[3 * vtx + 2] = n[2]; }
    }
}

static inline hipError_t pond_launch(const PondParams& P, const float* d_pos, int64_t nverts, float t, float* d_out,
                                     float* d_nrm, hipStream_t st) {
    const bool vec = mw_aligned16(d_pos) && mw_aligned16(d_out) && (!d_nrm || mw_aligned16(d_nrm));
    const int64_t nvec = vec ? (nverts & ~(int64_t)3) : 0;
    int64_t blocks = ((vec ? (nverts >> 2) : nverts) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (d_nrm)
        k_pond<true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(d_pos, d_out, d_nrm, nverts, nvec, P, t);
    else
        k_pond<false><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(d_pos, d_out, nullptr, nverts, nvec, P, t);
    return hipGetLastError();
}
#endif

}  // namespace mw
