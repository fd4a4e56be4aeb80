// czt_kernels.h -- chirp-z (Bluestein) form of the separable sum for grids the FFT cannot express, O(N^2 log N).
//
// The reference's basis (S/FFTMesh.cs:107-112, 199-208) on ANY grid is bilinear in the two indices:
//     k_i x_a = theta (i - c)(a - c'),   theta = 2 pi unit_width / length,  c = N/2,  c' = (N - 1)/2
// (even N: x_a = (a - N/2 + 1/2) u; odd N: x_a = (a - (N-1)/2) u -- the same c' either way).  With
//     (i - c)(a - c') = 1/2 [ (i - c)^2 + (a - c')^2 - (i - a + delta)^2 ],   delta = c' - c = -1/2
// one axis of the sum is a chirp-modulated CONVOLUTION
//     out[a] = w2[a] * sum_i (x[i] w1[i]) g[i - a],   w1[i] = e^{i theta (i-c)^2/2},  w2[a] = e^{i theta (a-c')^2/2},
//                                                      g[d]  = e^{-i theta (d+delta)^2/2}
// evaluated by two LDS-staged Stockham transforms of size M >= n_in + n_out - 1 (the passes of mw_math.h, the same code the FFT path
// runs): forward transform of the zero-padded x w1, product with the precomputed transform of the wrapped kernel, inverse
// transform, post-chirp (the 1/M of the unnormalised inverse folded into w2).  The 2-D sum is this along j, then along i;
// every launch stores its result transposed, so both launches read contiguous rows and the second one lands in [a][b].
// All chirps and the kernel's transform are formed in f64 on the host once per handle (they do not depend on t).
//
// Phase functions are MW_HD so that tests/emul steps the same code on the host (tests/test_emul.py::test_chirp_z_*).
// STATUS (round 4): green on hardware at N = 12 ... 1500 and the default for every non-FFT grid with 2N <= 4096
// (direct_alloc); the MFMA GEMM form (direct_kernels.h) serves larger grids and MW_DIRECT_CZT=0.
#pragma once
#include "fftmesh_kernels.h"

#include <vector>

namespace mw {

struct CztArgs {
    const cf* in = nullptr;   // [field][row][in_ld]: rows of N values, contiguous
    cf* out = nullptr;        // transposed store: element (row r, column n) of field f at out[f * out_plane + n * out_ld + r]
    const cf* w1 = nullptr;   // [N] pre-chirp
    const cf* w2 = nullptr;   // [N] post-chirp / M
    const cf* Hh = nullptr;   // [M] forward transform of the wrapped kernel g
    const cf* TWf = nullptr;  // twiddle tables of the size-M forward (sign -1) and inverse (+1) transforms, TwGeom<M, P> layout
    const cf* TWi = nullptr;
    int nin = 0, nout = 0;    // values per input row (zero-padded up to M), values per output row
    int rows = 0, in_ld = 0, out_ld = 0;
    long long in_plane = 0, out_plane = 0;
    // first launch of a step: h0 != nullptr -> the input planes are not read but FORMED from the spectrum on the way in
    // (czt_packed_value; the multiplier spectra never exist in memory)
    const cf* h0 = nullptr;
    const cf* h0c = nullptr;
    float t = 0.f;
    OceanConsts C = {};
    // what of an element's arithmetic does not depend on t, tabulated when the handle is created (round 5; k_czt_tables): the quantised
    // dispersion omega(i, j) on [0, N]^2 -- row stride N + 1; omega(N - i, N - j) = omega(i, j) bit for bit, one value serves both halves -- and the
    // wave number k(i) of an index.  Formed by omega_f32 / wave_k themselves: the same bits the launch used to recompute per element and step
    // (three correctly rounded divisions and two square roots for omega, one division per wave number: half of the launch's instructions).
    const float* Om = nullptr;  // [(N + 1) * (N + 1)]
    const float* K = nullptr;   // [N + 1]
};

// HERMITIAN PACKING ON ANY GRID (round 4): five real outputs in THREE complex sums.  The outputs are real / imaginary parts of
// S_f(a,b) = sum_ij F_f(i,j) e^{i theta [(i-c)(a-c') + (j-c)(b-c')]}, c = N/2 (S/FFTMesh.cs:211-218).  conj S runs over the MIRRORED
// indices i' = N - i, which lie in [1, N]: on the index set [0, N]^2 (N + 1 points per axis, F = 0 outside [0, N)^2) the real part of
// S_0 is the sum of Hh(i,j) = 1/2 [h~(i,j) + conj h~(N-i, N-j)], and because every multiplier is real and ODD in k (k(N - i) = -k(i)),
//     H + i Dx = sum Hh (1 + kx/|k|),    Sx + i Sz = sum Hh (kz - i kx),    Dz = sum Hh (i kz/|k|)
// (checked against the brute-force sums to 1e-14 for even, odd and non-commensurate grids).  Fields of ONE scale share a plane:
// the slopes carry a factor |k| over height and displacement -- 220 on the Inspector-default grid --, and a float32 transform's
// error is relative to the plane's largest entry (Dz packed with Sx measured 7e-6 there instead of 2e-7).  omega(N - i, N - j) = omega(i, j) bit for bit, so one sine serves both halves.  Three planes of
// (N + 1) x (N + 1) go through the chirp-z launches instead of five of N x N: 0.6 of the transform work.
#define MW_CZT_PLANES 3
// the arithmetic of one element from the four spectrum values it needs: (a0, b0) = (h0, h0conj)(i, j) where in0, (a1, b1) = the same at the
// mirrored index (N - i, N - j) where in1
MW_HD cf czt_packed_from(const OceanConsts& C, cf a0, cf b0, bool in0, cf a1, cf b1, bool in1, float t, float om, float kx, float kz, int plane) {
    float s, c;
    mw_sincos(smul(om, t), &s, &c);  // omega * t: one float32 product (omega_t_f32)
    const cf z = mk(0.f, 0.f);
    const cf h = in0 ? animate(a0.x, a0.y, b0.x, b0.y, c, s) : z;   // :188
    const cf hm = in1 ? animate(a1.x, a1.y, b1.x, b1.y, c, s) : z;
    const cf Hh = mk(0.5f * (h.x + hm.x), 0.5f * (h.y - hm.y));
    if (plane == 1) return cmul(Hh, mk(kz, -kx));  // Hh (kz - i kx): real part Sx (= Im sum kx h~), imaginary part Sz
    const float kl = sqrtf(kx * kx + kz * kz);
    float ux = 0.f, uz = 0.f;
    if (!(kl < MW_EPS_F)) { ux = kx / kl; uz = kz / kl; }  // |k| < EPSILON: skipped, :213-215
    if (plane == 0) return cscale(Hh, 1.0f + ux);  // real part H, imaginary part Dx
    return mk(-(Hh.y * uz), Hh.x * uz);            // Hh (i kz/|k|): real part Dz (= Im sum (-kz/|k|) h~)
}
// one element with everything formed in place (what the tables hold is what these calls return)
MW_HD cf czt_packed_value(const OceanConsts& C, const cf* h0, const cf* h0c, float t, int i, int j, int plane) {
    const int N = C.N;
    const bool in0 = i < N && j < N, in1 = i > 0 && j > 0;
    const size_t i0 = in0 ? (size_t)i * N + j : 0, i1 = in1 ? (size_t)(N - i) * N + (N - j) : 0;
    return czt_packed_from(C, h0[i0], h0c[i0], in0, h0[i1], h0c[i1], in1, t, omega_f32(N, C.length, C.gravity, i, j), wave_k(N, C.length, i),
                           wave_k(N, C.length, j), plane);
}
// the tables of CztArgs::Om / K, element e of the (N + 1)^2 index set (and of the N + 1 wave numbers)
MW_HD void czt_table_element(int N, float length, float gravity, int e, float* Om, float* K) {
    const int i = e / (N + 1), j = e % (N + 1);
    Om[e] = omega_f32(N, length, gravity, i, j);
    if (e <= N) K[e] = wave_k(N, length, e);
}

// transform size and points per thread for a grid of N points per axis (0: N too large for one workgroup-resident transform)
inline int czt_size_io(int nin, int nout) {  // the cyclic convolution must hold nin + nout - 1 distinct lags
    int M = 64;
    while (M < nin + nout - 1) M *= 2;
    return M <= 4096 ? M : 0;
}
inline int czt_size(int N) { return czt_size_io(N + 1, N); }  // the packed planes carry N + 1 points per axis
constexpr int czt_points(int M) { return M >= 256 ? 16 : 8; }
constexpr int czt_rows(int M) { return (512 / (M / czt_points(M))) < 1 ? 1 : ((512 / (M / czt_points(M))) > 8 ? 8 : (512 / (M / czt_points(M)))); }

// Pre-chirped, zero-padded input of a line.  The loads of four elements are requested before the first of them is used (round 5):
// written element by element -- bounds test, load, arithmetic, next element -- each element's loads sat in a branch of their own behind the
// previous element's arithmetic (`j L L W W j L L W W ...` in the ISA: two memory latencies per element, 16-32 per line, in the launch whose
// input is formed from the spectrum).  Out-of-range elements load element 0 (no branch around a load); the arithmetic stays behind the bounds test.
// CH_: elements requested together (their loaded values are live until used: 10 registers per element of the spectrum form)
#ifndef MW_CZT_LOAD_CHUNK
#define MW_CZT_LOAD_CHUNK 4  // 1: element by element (A/B); 8: slower from M = 2048 up (140 registers)
#endif
template <int M, int P, int CH_ = MW_CZT_LOAD_CHUNK>
MW_HD void czt_load(const CztArgs& A, int f, int row_, int u, bool live, cf (&x)[P]) {
    constexpr int T = M / P, CH = P < CH_ ? P : CH_;
    const cf z = mk(0.f, 0.f);
    const int row = live ? row_ : 0;  // a line past the end reads row 0 (and keeps nothing): every address below is in range whatever the caller passed
    if (A.h0 == nullptr) {
        const cf* __restrict__ r = A.in + (size_t)f * A.in_plane + (size_t)row * A.in_ld;
#pragma unroll
        for (int q0 = 0; q0 < P; q0 += CH) {
            cf rv[CH], wv[CH];
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int n = u + T * (q0 + k);
                const int idx = (live && n < A.nin) ? n : 0;
                rv[k] = r[idx];
                wv[k] = A.w1[idx];
            }
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int n = u + T * (q0 + k);
                x[q0 + k] = (live && n < A.nin) ? cmul(rv[k], wv[k]) : z;  // zero padding up to M
            }
        }
        return;
    }
    const int N = A.C.N;
    const float kx = A.K[row];  // (row is 0 for a line past the end)
    const float* __restrict__ omrow = A.Om + (size_t)row * (N + 1);
#pragma unroll
    for (int q0 = 0; q0 < P; q0 += CH) {
        cf a0[CH], b0[CH], a1[CH], b1[CH], wv[CH];
        float om[CH], kz[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int n = u + T * (q0 + k);
            const bool valid = live && n < A.nin;
            const bool in0 = valid && row < N && n < N, in1 = valid && row > 0 && n > 0;
            const size_t i0 = in0 ? (size_t)row * N + n : 0, i1 = in1 ? (size_t)(N - row) * N + (N - n) : 0;
            a0[k] = A.h0[i0]; b0[k] = A.h0c[i0];
            a1[k] = A.h0[i1]; b1[k] = A.h0c[i1];
            wv[k] = A.w1[valid ? n : 0];
            om[k] = omrow[valid ? n : 0];
            kz[k] = A.K[valid ? n : 0];
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int n = u + T * (q0 + k);
            const bool valid = live && n < A.nin;
            const bool in0 = valid && row < N && n < N, in1 = valid && row > 0 && n > 0;
            x[q0 + k] = z;  // zero padding up to M: no arithmetic there (at N = 12 four fifths of a line are padding)
            if (valid) x[q0 + k] = cmul(czt_packed_from(A.C, a0[k], b0[k], in0, a1[k], b1[k], in1, A.t, om[k], kx, kz[k], f), wv[k]);
        }
    }
}
template <int M, int P>
MW_HD void czt_mul_kernel(const CztArgs& A, int u, cf (&x)[P]) {
    constexpr int T = M / P;
#pragma unroll
    for (int q = 0; q < P; q++) x[q] = cmul(x[q], A.Hh[u + T * q]);
}
template <int M, int P>
MW_HD void czt_store(const CztArgs& A, int f, int row, int u, const cf (&x)[P]) {
    constexpr int T = M / P;
    cf* __restrict__ o = A.out + (size_t)f * A.out_plane + row;
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int n = u + T * q;
        if (n < A.nout) o[(size_t)n * A.out_ld] = cmul(x[q], A.w2[n]);
    }
}

// ---- host-side tables (f64, once per handle) -------------------------------------------------------------------------
// theta from the SAME float32 unit_width and length the kernels' other paths use; everything after that in double
inline void czt_fft_f64(std::vector<double>& re, std::vector<double>& im, int sign) {  // in-place radix-2, size a power of two
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = sign * 2.0 * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; k++) {
                const double c = cos(ang * (double)k), s = sin(ang * (double)k);
                const double ur = re[i + k], ui = im[i + k];
                const double vr = re[i + k + len / 2] * c - im[i + k + len / 2] * s, vi = re[i + k + len / 2] * s + im[i + k + len / 2] * c;
                re[i + k] = ur + vr; im[i + k] = ui + vi;
                re[i + k + len / 2] = ur - vr; im[i + k + len / 2] = ui - vi;
            }
    }
}
// nin input indices i = 0 .. nin-1 (N, or N + 1 for the packed planes), N outputs a = 0 .. N-1; M >= nin + N - 1
inline void czt_build_tables(int N, int nin, int M, float unit_width, float length, std::vector<cf>& w1, std::vector<cf>& w2, std::vector<cf>& Hh) {
    const double theta = 2.0 * M_PI * (double)unit_width / (double)length;
    const double c = N / 2.0, cp = (N - 1) / 2.0, delta = cp - c;
    auto chirp = [&](double s) {  // e^{i theta s^2 / 2}, the phase reduced in double
        const double ph = fmod(theta * s * s / 2.0, 2.0 * M_PI);
        return std::pair<double, double>(cos(ph), sin(ph));
    };
    w1.resize(nin); w2.resize(N); Hh.resize(M);
    for (int i = 0; i < nin; i++) {
        auto a = chirp((double)i - c);
        w1[i] = mk((float)a.first, (float)a.second);
    }
    for (int i = 0; i < N; i++) {
        auto b = chirp((double)i - cp);
        w2[i] = mk((float)(b.first / M), (float)(b.second / M));
    }
    // conv[a] = sum_i y[i] h[a - i] with h[m] = g[-m] = e^{-i theta (-m + delta)^2 / 2}, -(nin - 1) <= m <= N - 1, wrapped modulo M
    std::vector<double> re(M, 0.0), im(M, 0.0);
    for (int m = -(nin - 1); m <= N - 1; m++) {
        auto g = chirp((double)(-m) + delta);
        re[(m + M) % M] = g.first;
        im[(m + M) % M] = -g.second;
    }
    czt_fft_f64(re, im, -1);
    for (int k = 0; k < M; k++) Hh[k] = mk((float)re[k], (float)im[k]);
}

}  // namespace mw
