// direct_kernels.h -- separable direct-sum evaluation of FFTMesh.EvaluateWaves for grids the FFT cannot
// express: non-power-of-two N, N < 64, or unit_width != length/N (the SHIPPED scene: N=12, unitWidth=1,
// length=12.39, D/FFT Mesh.unity:147-150).  O(N^3) instead of the reference's O(N^4) by
//   sum_ij F(i,j) e^{i(kx_i x_a + kz_j z_b)} = sum_i e^{i kx_i x_a} [ sum_j F(i,j) e^{i kz_j z_b} ]
// (S/FFTMesh.cs:199-217).  Not a throughput path; correctness for the literal drop-in only.
#pragma once
#include "fftmesh_kernels.h"

namespace mw {

struct DirectState {
    cf* spec = nullptr;   // [5][N*N]  H, Dx, Dz, Sx, Sz spectra
    cf* tmp = nullptr;    // [5][N*N]  after the z-sum, indexed [f][i][b]
    cf* etab = nullptr;   // [N*N]     e^{i k_j pos_b}, phase formed in double
    cf* hds = nullptr;    // [N*N]
    int N = 0;
};

#if defined(__HIPCC__)
__global__ void k_direct_etab(int N, float length, float unit_width, cf* etab) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    int j = idx / N, b = idx % N;
    double k = (double)wave_k(N, length, j), pos = (double)rest_coord(N, unit_width, b);
    double s, c;
    sincos(k * pos, &s, &c);
    etab[idx] = mk((float)c, (float)s);
}

// S/FFTMesh.cs:178-190 htilde + the five multiplier spectra of :211-215
__global__ void k_direct_spec(OceanConsts C, const cf* h0, const cf* h0c, float t, cf* spec) {
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
    const size_t NN = (size_t)N * N;
    spec[idx] = h;
    spec[NN + idx] = cscale(h, ux);
    spec[2 * NN + idx] = cscale(h, uzn);
    spec[3 * NN + idx] = cscale(h, kx);
    spec[4 * NN + idx] = cscale(h, kz);
}

__global__ void k_direct_zsum(int N, const cf* spec, const cf* etab, cf* tmp) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    int i = idx / N, b = idx % N;
    const size_t NN = (size_t)N * N;
    cf acc[5];
    for (int f = 0; f < 5; f++) acc[f] = mk(0.f, 0.f);
    for (int j = 0; j < N; j++) {
        cf e = etab[(size_t)j * N + b];
        for (int f = 0; f < 5; f++) acc[f] = acc[f] + cmul(spec[f * NN + (size_t)i * N + j], e);
    }
    for (int f = 0; f < 5; f++) tmp[f * NN + idx] = acc[f];
}

__global__ void k_direct_xsum(OceanConsts C, const cf* tmp, const cf* etab, cf* hds, float* vertices, float* normals) {
    const int N = C.N;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    int a = idx / N, b = idx % N;
    const size_t NN = (size_t)N * N;
    cf acc[5];
    for (int f = 0; f < 5; f++) acc[f] = mk(0.f, 0.f);
    for (int i = 0; i < N; i++) {
        cf e = etab[(size_t)i * N + a];
        for (int f = 0; f < 5; f++) acc[f] = acc[f] + cmul(tmp[f * NN + (size_t)i * N + b], e);
    }
    const float h = acc[0].x, dx = acc[1].y, dz = acc[2].y, sx = acc[3].y, sz = acc[4].y;
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

static inline int direct_alloc(DirectState& d, int N) {
    d.N = N;
    const size_t NN = (size_t)N * N;
    if (hipMalloc((void**)&d.spec, sizeof(cf) * 5 * NN) != hipSuccess) return 4;
    if (hipMalloc((void**)&d.tmp, sizeof(cf) * 5 * NN) != hipSuccess) return 4;
    if (hipMalloc((void**)&d.etab, sizeof(cf) * NN) != hipSuccess) return 4;
    if (hipMalloc((void**)&d.hds, sizeof(cf) * NN) != hipSuccess) return 4;
    return 0;
}
static inline void direct_free(DirectState& d) {
    hipFree(d.spec); hipFree(d.tmp); hipFree(d.etab); hipFree(d.hds);
    d.spec = d.tmp = d.etab = d.hds = nullptr;
}
static inline hipError_t direct_evaluate(DirectState& d, OceanConsts C, const cf* h0, const cf* h0c, float t, float* dv,
                                         float* dn, float* dw, int white_stride, hipStream_t st) {
    const int N = C.N;
    const unsigned nb = (unsigned)(((size_t)N * N + 127) / 128);
    hipLaunchKernelGGL(k_direct_etab, dim3(nb), dim3(128), 0, st, N, C.length, C.unit_width, d.etab);
    hipLaunchKernelGGL(k_direct_spec, dim3(nb), dim3(128), 0, st, C, h0, h0c, t, d.spec);
    hipLaunchKernelGGL(k_direct_zsum, dim3(nb), dim3(128), 0, st, N, d.spec, d.etab, d.tmp);
    hipLaunchKernelGGL(k_direct_xsum, dim3(nb), dim3(128), 0, st, C, d.tmp, d.etab, d.hds, dv, dn);
    hipLaunchKernelGGL(k_direct_white, dim3(nb), dim3(128), 0, st, N, d.hds, dn, dw, white_stride);
    return hipGetLastError();
}
#endif

}  // namespace mw
