// mw_math.h -- arithmetic shared by every kernel of libmistral_water.so.
//
// Everything here is `__host__ __device__` so that tests/emul (a g++-compiled, host-side
// lock-step emulation of the kernels' thread/LDS choreography) can exercise exactly the
// code the GPU runs.  The emulation is test infrastructure; the product library contains
// device code only and has no CPU fallback.
//
// Reference citations: S/ = /root/reference/Assets/Mistral Water/Scripts/.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MW_HD __host__ __device__ __forceinline__
#else
#define MW_HD inline
#endif

#define MW_PI_F 3.1415926536f  /* S/FFTMesh.cs:50 */
#define MW_EPS_F 0.0001f       /* S/FFTMesh.cs:54 */

namespace mw {

// ---- complex helper ---------------------------------------------------------------------
struct cf {
    float x, y;
};
MW_HD cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }
MW_HD cf operator+(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
MW_HD cf operator-(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
// MW_FFT_STRICT (default): the transform's rounding is fixed by the SOURCE -- cmul writes its two fused multiply-adds out
// and every butterfly below is compiled with FMA contraction off -- not by which a*b + c the optimiser happens to fuse in
// one instantiation of a kernel template and not in another.  A transformed row is then the same bit pattern whichever
// workgroup, thread mapping or kernel variant produced it: the halo row a workgroup transforms for its Jacobian IS the row
// its owner stores (tests/test_gpu_parity.py::test_whitecap_stage_bit_exact_on_device), and the host emulation equals the
// device up to the sine.
#ifndef MW_FFT_STRICT
#define MW_FFT_STRICT 1
#endif
MW_HD cf cmul(cf a, cf b) {
#if MW_FFT_STRICT && !defined(MW_CMUL_NO_FMA)
    return mk(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));  // the two FMAs written out
#else
    return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
#endif
}
MW_HD cf cscale(cf a, float s) { return mk(a.x * s, a.y * s); }
MW_HD cf cconj(cf a) { return mk(a.x, -a.y); }
template <int SGN>
MW_HD cf mul_si(cf a) {  // a * (SGN * i)
    return SGN > 0 ? mk(-a.y, a.x) : mk(a.y, -a.x);
}

// ---- strict IEEE float32 (no FMA contraction): the "index-like" scalars ------------------
// Device: `#pragma clang fp contract(off)` strips the contract flag from these instructions so they can
// never fuse after inlining (HIP's __fmul_rn/__fadd_rn are plain a*b / a+b and DO fuse; __fsqrt_rn is the
// approximate v_sqrt_f32).  `/` and sqrtf are correctly rounded under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt.  Host: compiled with -ffp-contract=off.
MW_HD float smul(float a, float b) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    return a * b;
}
MW_HD float sadd(float a, float b) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    return a + b;
}
MW_HD float ssub(float a, float b) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    return a - b;
}
MW_HD float sdiv(float a, float b) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    return a / b;
}
MW_HD float ssqrt(float a) { return sqrtf(a); }

// ---- sin/cos of a float32 phase, |x| <~ 1e5 rad ---------------------------------------------------
// The reference forms cos/sin of the FLOAT omega*t through double libm (Mathf.Cos, S/FFTMesh.cs:184-185).
// Here: 3-term Cody-Waite reduction by pi/2 carried in FMAs (exact for |k| < 2^17) + the classic
// degree-7/8 minimax polynomials on [-pi/4, pi/4]; absolute error <= ~1.5e-7 (about 1 ulp of a result
// near 1), identical code on host (emulation) and device.  ~25 VALU ops, no branches, 6 live registers.
MW_HD void sincos_f32(float x, float* sn, float* cs) {
    const float k = rintf(x * 0.63661977236758134308f);  // 2/pi
    float r = fmaf(-k, 1.5703125f, x);                    // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.5497899e-8
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.549789948768648e-8f, r);
    const float z = r * r;
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    ps = fmaf(ps * z, r, r);  // sin(r)
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    pc = fmaf(pc * z, z, fmaf(z, -0.5f, 1.0f));  // cos(r)
    const int q = (int)k;
    const float s0 = (q & 1) ? pc : ps;
    const float c0 = (q & 1) ? ps : pc;
    *sn = (q & 2) ? -s0 : s0;
    *cs = ((q + 1) & 2) ? -c0 : c0;
}

// sin/cos through the hardware's v_sin_f32 / v_cos_f32 (argument in revolutions, quarter-rate, ~1e-6 absolute) after a
// two-constant reduction that keeps the revolution count exact: p = x/2pi rounded, e = the rounding error of that
// product recovered by FMA plus the low part of 1/2pi, r = (p - rint(p)) + e in [-0.5, 0.5].  10 VALU issue slots
// against ~25 for sincos_f32: used where a kernel is VALU-bound and 1e-6 is inside its stated tolerance (the pond: eight
// sincos per vertex against 24 B).  The host (emulation) evaluates sin(2 pi r) in double: same r, correctly rounded
// result -- host and device agree to the hardware's error, not bit for bit.
MW_HD void sincos_fast_f32(float x, float* sn, float* cs) {
    const float hi = 0.15915494f, lo = 6.4206382e-09f;  // 1/(2 pi) = hi + lo (hi = the f32 nearest, lo = the rest)
    const float p = smul(x, hi);  // the ROUNDED product (a contracted x*hi - rint(p) would count its error twice)
    float e = fmaf(x, hi, -p);
    e = fmaf(x, lo, e);
    const float r = sadd(ssub(p, rintf(p)), e);
#if defined(__HIP_DEVICE_COMPILE__)
    *sn = __builtin_amdgcn_sinf(r);
    *cs = __builtin_amdgcn_cosf(r);
#else
    const double a = 6.283185307179586476925 * (double)r;
    *sn = (float)sin(a);
    *cs = (float)cos(a);
#endif
}

// S/FFTMesh.cs:141-147 Dispersion(n,m) * t  (:183) -- bit-for-bit the reference's float sequence.
MW_HD float omega_f32(int N, float length, float gravity, int n, int m) {
    float w = sdiv(smul(2.0f, MW_PI_F), length);
    float kx = sdiv(smul(MW_PI_F, (float)(2 * n - N)), length);
    float kz = sdiv(smul(MW_PI_F, (float)(2 * m - N)), length);
    float s = sadd(smul(kx, kx), smul(kz, kz));
    float r = ssqrt(smul(gravity, ssqrt(s)));
    return smul(floorf(sdiv(r, w)), w);
}
MW_HD float omega_t_f32(int N, float length, float gravity, int n, int m, float t) {
    return smul(omega_f32(N, length, gravity, n, m), t);
}

// S/FFTMesh.cs:201,204  kx = 2*PI*(i - resolution/2.0f)/length
MW_HD float wave_k(int N, float length, int i) {
    return sdiv(smul(smul(2.0f, MW_PI_F), ssub((float)i, sdiv((float)N, 2.0f))), length);
}

// S/FFTMesh.cs:107-112 rest coordinate of grid line a
MW_HD float rest_coord(int N, float unit_width, int a) {
    float base = smul((float)(a - N / 2), unit_width);
    return (N % 2 == 0) ? sadd(base, sdiv(unit_width, 2.0f)) : base;
}

// output sign of the shifted 2-D inverse DFT: -(-1)^(a+b)   (DESIGN.md "transform identity")
MW_HD float post_sign(int a, int b) { return ((a + b) & 1) ? 1.0f : -1.0f; }

// Mathf.SmoothStep(0,1,t) [unity], S/FFTMesh.cs:273
MW_HD float smoothstep01(float t) {
    t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
    // -2 t^3 + 3 t^2, evaluated left to right like the C# expression
    return sadd(smul(smul(smul(-2.0f, t), t), t), smul(smul(3.0f, t), t));
}

// S/FFTMesh.cs:258-274: whitecap scalar from hds (centre, +1 row, +1 col) and the unit normal.
// has_next_i / has_next_j encode the edge rules  i != N-1 (:260)  and  j != N-1 (:264).
MW_HD float whitecap(cf d, cf d_next_i, cf d_next_j, bool has_next_i, bool has_next_j, float nx, float nz) {
    float ax = 0.f, ay = 0.f, bx = 0.f, by = 0.f;
    if (has_next_i) { ax = smul(0.5f, ssub(d.x, d_next_i.x)); ay = smul(0.5f, ssub(d.y, d_next_i.y)); }
    if (has_next_j) { bx = smul(0.5f, ssub(d.x, d_next_j.x)); by = smul(0.5f, ssub(d.y, d_next_j.y)); }
    float jac = ssub(smul(sadd(1.f, ax), sadd(1.f, by)), smul(ay, bx));           // :268
    float n0 = smul(fabsf(nx), 0.3f), n1 = smul(fabsf(nz), 0.3f);                 // :269
    float turb = fmaxf(sadd(ssub(1.f, jac), ssqrt(sadd(smul(n0, n0), smul(n1, n1)))), 0.f);  // :270
    return smoothstep01(turb);                                                    // :273
}

// ---- the library's counter RNG (documented in DESIGN.md; mirrors oracle/fftmesh_oracle.c) ---
MW_HD uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
MW_HD float uniform01(uint64_t seed, uint64_t counter) {  // in (0,1]
    uint64_t bits = mix64(seed * 0xD1342543DE82EF95ull + counter);
    return (float)((uint32_t)(bits >> 40) + 1u) * (1.0f / 16777216.0f);
}

#if MW_FFT_STRICT && defined(__clang__)
#pragma clang fp contract(off)  // file scope: every function down to the matching pragma below
#endif
// ---- small in-register DFTs, natural-order in, natural-order out -------------------------
// SGN = +1: kernel e^{+2 pi i nk/R} (unnormalised inverse);  SGN = -1: forward.
template <int SGN>
MW_HD void dft2(cf& a, cf& b) {
    cf t = a;
    a = t + b;
    b = t - b;
}
template <int SGN>
MW_HD void dft4(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_si<SGN>(a1 - a3);
    a0 = t0 + t2;
    a1 = t1 + t3;
    a2 = t0 - t2;
    a3 = t1 - t3;
}
#define MW_SQRT1_2 0.70710678118654752440f
#define MW_COS_PI_8 0.92387953251128675613f
#define MW_SIN_PI_8 0.38268343236508977173f
template <int SGN>
MW_HD cf tw8(cf a, int m) {  // a * e^{SGN 2 pi i m/8}, m compile-time after unrolling
    switch (m & 7) {
        case 0: return a;
        case 1: return mk((a.x - SGN * a.y) * MW_SQRT1_2, (SGN * a.x + a.y) * MW_SQRT1_2);
        case 2: return mul_si<SGN>(a);
        case 3: return mk((-a.x - SGN * a.y) * MW_SQRT1_2, (SGN * a.x - a.y) * MW_SQRT1_2);
        case 4: return mk(-a.x, -a.y);
        case 5: return mk((-a.x + SGN * a.y) * MW_SQRT1_2, (-SGN * a.x - a.y) * MW_SQRT1_2);
        case 6: return mul_si<-SGN>(a);
        default: return mk((a.x + SGN * a.y) * MW_SQRT1_2, (-SGN * a.x + a.y) * MW_SQRT1_2);
    }
}
template <int SGN>
MW_HD cf tw16(cf a, int m) {  // a * e^{SGN 2 pi i m/16}
    m &= 15;
    if ((m & 1) == 0) return tw8<SGN>(a, m >> 1);
    // odd m: rotate by the multiple of 4 (quarter turns) then by +-pi/8 or +-3pi/8
    float c, s;
    switch (m) {
        case 1: c = MW_COS_PI_8; s = MW_SIN_PI_8; break;
        case 3: c = MW_SIN_PI_8; s = MW_COS_PI_8; break;
        case 5: c = -MW_SIN_PI_8; s = MW_COS_PI_8; break;
        case 7: c = -MW_COS_PI_8; s = MW_SIN_PI_8; break;
        case 9: c = -MW_COS_PI_8; s = -MW_SIN_PI_8; break;
        case 11: c = -MW_SIN_PI_8; s = -MW_COS_PI_8; break;
        case 13: c = MW_SIN_PI_8; s = -MW_COS_PI_8; break;
        default: c = MW_COS_PI_8; s = -MW_SIN_PI_8; break;
    }
    return cmul(a, mk(c, SGN * s));
}
template <int SGN>
MW_HD void dft8(cf (&x)[8]) {
    // n = 2 n1 + n2 ; X[k1 + 4 k2]
#pragma unroll
    for (int n2 = 0; n2 < 2; n2++) dft4<SGN>(x[n2], x[n2 + 2], x[n2 + 4], x[n2 + 6]);  // A[n2][k1] at x[n2+2k1]
#pragma unroll
    for (int k1 = 1; k1 < 4; k1++) x[1 + 2 * k1] = tw8<SGN>(x[1 + 2 * k1], k1);
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) dft2<SGN>(x[2 * k1], x[2 * k1 + 1]);  // X[k1 + 4k2] at x[2k1+k2]
    cf y[8];
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) y[k1 + 4 * k2] = x[2 * k1 + k2];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = y[k];
}
template <int SGN>
MW_HD void dft16(cf (&x)[16]) {
    // n = 4 n1 + n2 ; X[k1 + 4 k2]
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) dft4<SGN>(x[n2], x[n2 + 4], x[n2 + 8], x[n2 + 12]);  // A[n2][k1] at x[n2+4k1]
#pragma unroll
    for (int n2 = 1; n2 < 4; n2++)
#pragma unroll
        for (int k1 = 1; k1 < 4; k1++) x[n2 + 4 * k1] = tw16<SGN>(x[n2 + 4 * k1], n2 * k1);
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) dft4<SGN>(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
    cf y[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) y[k1 + 4 * k2] = x[4 * k1 + k2];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = y[k];
}

// ---- Stockham autosort passes, P points per thread (P = 8 or 16) -------------------------------
// An N-point transform is carried by T = N/P threads; thread u keeps element u + T*q in slot q before
// AND after the whole transform.  Passes: S radix-P passes (p = 1, P, P^2, ...) and, when P^S < N, a
// final radix-RL pass (RL = N / P^S < P) done as P/RL butterflies per thread.  Between passes the data
// goes through an LDS buffer of N + N/P complex (one pad per P elements: conflict-free b64 accesses).
//   P = 8  : half the registers per thread, twice the threads (used for N <= 2048)
//   P = 16 : one exchange fewer (used for N = 4096, where 4 columns x N/8 threads would exceed 1024)
template <int P> struct LogP;
template <> struct LogP<8> { static constexpr int v = 3; };
template <> struct LogP<16> { static constexpr int v = 4; };

constexpr int mw_full_stages(int N, int P) { int s = 0; long long m = 1; while (m * P <= N) { m *= P; s++; } return s; }
constexpr int mw_ipow(int P, int s) { int m = 1; for (int i = 0; i < s; i++) m *= P; return m; }

// LDS layouts of the exchanges between the radix-P passes (cf units; element n of the length-N intermediate sequence).
// A wave's ds_write_b64 is served in groups of 16 consecutive lanes over 32 dword banks, its ds_read_b64 in groups of 32
// lanes over 64: a group is conflict-free when its 16 (32) cf indices are distinct mod 16 (32).
//  * padded (every P, the only form up to round 2): n -> n + n/P in every exchange.  Stores are conflict-free, but the 32
//    lanes of a read, n = u + T q, span 33 entries (u + u/16): one bank pair is hit twice, i.e. every read at half rate --
//    and the compiler's ds_read2_b64 pairing (16-lane groups, 128 B/clk) merely hides it in the conflict counter.
//  * exact (P = 16, T a multiple of 32):
//      exchange 0  (stage 0 writes n = 16 u + r, a stride-16 scatter): the 16 x T transpose  n -> (n % 16) (T + 2) + n / 16.
//                  Stores: lanes write consecutive entries.  Reads: lane u of a 32-lane group reads at 2 (u % 16) + u / 16 + const.
//      exchange s >= 1 (pass s writes n = j + 16^s r with j = u within a 16-lane group): the identity, no padding; reads
//                  n = u + T q are consecutive in u.
//    Every read is then one conflict-free ds_read_b64 (256 B/clk); the build switches off the pairing of DS operations
//    (-target-feature -load-store-opt) because a paired read of the transposed layout is 2-way conflicted.
#ifndef MW_LDS_LAYOUT
#define MW_LDS_LAYOUT 2
#endif
#ifndef MW_LAST_IN_REGS
#define MW_LAST_IN_REGS 1  // skip the identity LDS round trip after the last radix-P pass when N = P^S (LastInRegs)
#endif
//  * exact, P = 8 (T a multiple of 64): exchange 0 is the 8 x T transpose n -> (n % 8)(T + 4) + n / 8 (reads: lane u at
//    4 (u % 8) + u / 8 + const); exchange 1 (pass 1 writes n = 64 (u/8) + 8 r + u % 8: two 8-entry runs 64 apart per 16-lane
//    group) pads 8 entries per 64, n -> n + 8 (n / 64); later exchanges are the identity.
template <int N, int P>
struct XLay {
    static constexpr int T = N / P;
    static constexpr bool EXACT = MW_LDS_LAYOUT && ((P == 16 && (T % 32) == 0) || (P == 8 && (T % 64) == 0 && MW_LDS_LAYOUT >= 2));
    static constexpr int ROW0 = T + (P == 16 ? 2 : 4);
    static constexpr int LBUF = !EXACT ? N + N / P : (P == 16 ? P * ROW0 : N + N / 8);
    // entries between the 4 buffers of a workgroup whose lanes interleave them (lane l -> buffer l % 4), LBUF = 0 mod 32:
    // a read of 8 consecutive n from each of 4 buffers wants stride 8 mod 32 (pass 1's final read), a store of 4 consecutive
    // n to each of 4 buffers stride 4 mod 16 (pass 2's stage 0)
    static constexpr int PAD_RD4 = 8, PAD_WR4 = 4;
};

template <int N, int P>
struct FftGeom {
    static constexpr int T = N / P;
    static constexpr int LOGP = LogP<P>::v;
    static constexpr int S = mw_full_stages(N, P);  // number of radix-P passes
    static constexpr int PS = mw_ipow(P, S);
    static constexpr int RL = N / PS;               // radix of the final pass (1 = none)
    static constexpr int NB = P / RL;               // butterflies per thread in the final pass
    static constexpr int LBUF = XLay<N, P>::LBUF;
    static_assert(N >= 64 && (N & (N - 1)) == 0 && N <= 4096, "N must be a power of two in [64,4096]");
    static_assert(S >= 1 && RL >= 1 && RL < P, "bad geometry");
};
template <int P>
MW_HD int lds_pad(int idx) { return idx + (idx >> LogP<P>::v); }

template <int P, int SGN> struct DftP;
template <int SGN> struct DftP<8, SGN> {
    static MW_HD void run(cf (&x)[8]) { dft8<SGN>(x); }
    static MW_HD cf rot(cf a, int m) { return tw8<SGN>(a, m); }
};
template <int SGN> struct DftP<16, SGN> {
    static MW_HD void run(cf (&x)[16]) { dft16<SGN>(x); }
    static MW_HD cf rot(cf a, int m) { return tw16<SGN>(a, m); }
};

// Twiddle tables (built on the host in double, rounded once to f32; SGN baked in):
//   TS[s][k*(P+1) + r] = e^{SGN 2 pi i r k / P^(s+1)}   k < P^s, r < P   (radix-P pass s >= 1, p = P^s)
//   TF[r*T + u]        = e^{SGN 2 pi i r u / N}         u < T,  r < RL   (final pass; the remaining factor
//                        e^{SGN 2 pi i r m / P} of thread u's m-th butterfly is a compile-time rotation)
// Each thread reads one contiguous run of TS per pass.  Rows are P+1 apart (odd stride) and TF is r-major so that,
// when the tables sit in LDS, the lanes of a wave (16 distinct k, consecutive u) fall on distinct banks: with the
// natural P-stride all 16 rows of a radix-16 pass land on two bank groups (an 8-way conflict on every read).
struct Twiddles {
    const cf* TS[4];
    const cf* TF;
    const cf* PW;  // compact base twiddles of the powers pass (TwGeom::PW_CF entries in LDS) or nullptr
};
// one concatenated table [TS1 | TS2 | TS3 | TF] (device global memory).  Its leading sub-tables are copied to LDS up to an
// 8-KiB budget (`LDS_CF` entries: the whole table when it fits -- every N <= 1024 plan -- else the longest prefix of whole
// sub-tables); the rest is read from global memory (L1/L2 hits).  Exception, `POW_STAGE`: the last radix-8 pass of a large
// transform (N = 4096 with P = 8: 512 rows of 7 twiddles, 36 KiB) reads ONE twiddle per thread, w = e^{SGN 2 pi i k/P^(s+1)},
// and forms w^2..w^7 by a product tree three multiplications deep: 2 VGPRs of loads in flight instead of 14, which is what
// lets that kernel fit the 64-VGPR budget of two resident 1024-thread workgroups per CU.
template <int N, int P>
struct TwGeom {
    static constexpr int S = FftGeom<N, P>::S;
    static constexpr int size_ts(int s) { return (s >= 1 && s < S) ? mw_ipow(P, s) * (P + 1) : 0; }
    static constexpr int OFF1 = 0;
    static constexpr int OFF2 = OFF1 + size_ts(1);
    static constexpr int OFF3 = OFF2 + size_ts(2);
    static constexpr int OFFF = OFF3 + size_ts(3);
    static constexpr int SIZE_TF = (FftGeom<N, P>::RL > 1) ? FftGeom<N, P>::T * FftGeom<N, P>::RL : 0;
    static constexpr int TOTAL = OFFF + SIZE_TF;
    static constexpr int LDS_BUDGET_CF = 1024;  // 8 KiB
    static constexpr int LDS_CF = TOTAL <= LDS_BUDGET_CF ? TOTAL
                                  : (OFFF <= LDS_BUDGET_CF ? OFFF : (OFF3 <= LDS_BUDGET_CF ? OFF3 : (OFF2 <= LDS_BUDGET_CF ? OFF2 : 0)));
    static constexpr bool IN_LDS = (LDS_CF == TOTAL);  // the whole table is staged
    static constexpr int off_ts(int s) { return s == 1 ? OFF1 : (s == 2 ? OFF2 : OFF3); }
#ifndef MW_POW_MIN_P
#define MW_POW_MIN_P 8
#endif
#ifndef MW_POW_MAX_P
#define MW_POW_MAX_P 16
#endif
    static constexpr int POW_STAGE =
        (P >= MW_POW_MIN_P && P <= MW_POW_MAX_P && S >= 2 && off_ts(S - 1) >= LDS_CF && size_ts(S - 1) > LDS_BUDGET_CF) ? S - 1 : 0;
    // The powers pass reads ONE twiddle per thread, w_k = TS[POW_STAGE][k (P+1) + 1], k < P^POW_STAGE: those P^s values are
    // gathered into LDS behind the staged prefix (2 KiB at N = 4096), so that no pass of a transform touches global memory
    // -- a global load in the middle of a transform would make its s_waitcnt (vmcnt is in-order) wait for every
    // exchange-buffer load prefetched before it (k_pass2_hs, PF).
#ifndef MW_POW_LDS
#define MW_POW_LDS 1
#endif
    static constexpr int PW_CF = (MW_POW_LDS && POW_STAGE != 0) ? mw_ipow(P, POW_STAGE) : 0;
    static constexpr int LDS_ALL = LDS_CF + PW_CF;  // cf entries a kernel stages (stage_twiddles)
    // sub-table at offset `off`: the LDS copy when it was staged, else global memory
    static MW_HD const cf* pick(const cf* glob, const cf* lds, int off) { return off < LDS_CF ? lds + off : glob + off; }
    static MW_HD Twiddles view(const cf* glob, const cf* lds) {
        Twiddles t;
        t.TS[0] = nullptr;
        t.TS[1] = pick(glob, lds, OFF1);
        t.TS[2] = pick(glob, lds, OFF2);
        t.TS[3] = pick(glob, lds, OFF3);
        t.TF = pick(glob, lds, OFFF);
        t.PW = PW_CF != 0 ? lds + LDS_CF : nullptr;
        return t;
    }
    static MW_HD Twiddles view(const cf* base) {  // host emulation: one flat table, no PW
        Twiddles t = view(base, base);
        t.PW = nullptr;
        return t;
    }
};

// Staging of the twiddle tables (TwGeom::LDS_CF leading entries + the PW_CF base twiddles of a powers pass) by NT threads, in two halves
// (round 5): load() puts EVERY global load of the workgroup's share in flight at once, store() writes them to LDS -- placed by the kernels
// behind the requests of their first input rows and in front of their first barrier, so that the tables ride under the same memory latency
// as the data.  Up to here it was a copy loop `for (i = tid; i < LDS_CF; i += NT) lds[i] = tw[i]`, which compiles to load -> s_waitcnt
// vmcnt(0) -> ds_write per iteration: three to five L2 round trips ONE AFTER THE OTHER at the head of every workgroup, in front of a barrier
// in front of its first data request (k_pass2_hs<1024>: five of them, 128 lanes for 528 entries).  MW_TW_STAGE=0 keeps that loop (A/B).
#ifndef MW_TW_STAGE
#define MW_TW_STAGE 2  // 2: loads at the top, LDS writes behind the first data requests; 1: loads then writes at the top; 0: the copy loop
#endif
#if defined(__HIPCC__)
// WITH_PW = false: a kernel that keeps the table in every pass (stage_regs<.., ALLOW_POW = false>: the OceanRenderer passes) and whose LDS
// layout has no room for the base twiddles of a powers pass
template <int N, int P, int NT, bool WITH_PW = true>
struct TwStage {
    using TG = TwGeom<N, P>;
    static constexpr int PW = WITH_PW ? TG::PW_CF : 0;
    static constexpr int IT = MW_TW_STAGE ? (TG::LDS_CF + NT - 1) / NT : 0, ITP = MW_TW_STAGE ? (PW + NT - 1) / NT : 0;
    cf r[IT > 0 ? IT : 1], rp[ITP > 0 ? ITP : 1];
    __device__ __forceinline__ void store_now(cf* dst, int tid) const {
#pragma unroll
        for (int k = 0; k < IT; k++) { const int i = tid + k * NT; if (IT * NT == TG::LDS_CF || i < TG::LDS_CF) dst[i] = r[k]; }
#pragma unroll
        for (int k = 0; k < ITP; k++) { const int i = tid + k * NT; if (ITP * NT == PW || i < PW) dst[TG::LDS_CF + i] = rp[k]; }
    }
    // at the top of the kernel
    __device__ __forceinline__ void load(cf* dst, const cf* __restrict__ src, int tid) {
        if (MW_TW_STAGE == 0) {
            for (int i = tid; i < TG::LDS_CF; i += NT) dst[i] = src[i];
            if (PW)  // base twiddles of the powers pass: entry 1 of every (P+1)-entry row
                for (int i = tid; i < PW; i += NT) dst[TG::LDS_CF + i] = src[TG::off_ts(TG::POW_STAGE) + i * (P + 1) + 1];
            return;
        }
#pragma unroll
        for (int k = 0; k < IT; k++) { const int i = tid + k * NT; r[k] = (IT * NT == TG::LDS_CF || i < TG::LDS_CF) ? src[i] : mk(0.f, 0.f); }
#pragma unroll
        for (int k = 0; k < ITP; k++) {
            const int i = tid + k * NT;
            rp[k] = (ITP * NT == PW || i < PW) ? src[TG::off_ts(TG::POW_STAGE) + i * (P + 1) + 1] : mk(0.f, 0.f);
        }
        if (MW_TW_STAGE == 1) store_now(dst, tid);
    }
    // behind the kernel's first data requests, in front of its first barrier
    __device__ __forceinline__ void store(cf* dst, int tid) const { if (MW_TW_STAGE == 2) store_now(dst, tid); }
};
#endif

// LDS indices are written as (per-thread base) + (compile-time constant) so that every ds_read / ds_write
// carries its offset in the instruction's immediate field and costs no address VALU.
template <int N, int P, int SGN>
MW_HD void stage0_store(cf (&x)[P], int u, cf* buf) {
    DftP<P, SGN>::run(x);
    if (XLay<N, P>::EXACT) {
        cf* __restrict__ b = buf + u;
#pragma unroll
        for (int r = 0; r < P; r++) b[r * XLay<N, P>::ROW0] = x[r];
        return;
    }
    cf* __restrict__ b = buf + (P + 1) * u;  // lds_pad(P*u + r) = (P+1)*u + r for r < P
#pragma unroll
    for (int r = 0; r < P; r++) b[r] = x[r];
}
// e: which exchange is read (0 = the one stage 0 wrote, s = the one pass s wrote)
template <int N, int P>
MW_HD void load_slots(cf (&x)[P], int u, const cf* buf, int e) {
    constexpr int T = FftGeom<N, P>::T;
    if (XLay<N, P>::EXACT) {
        if (e == 0) {  // n = u + T q = 16 (u/16 + (T/16) q) + u % 16
            const cf* __restrict__ b = buf + (u & (P - 1)) * XLay<N, P>::ROW0 + (u >> LogP<P>::v);
#pragma unroll
            for (int q = 0; q < P; q++) x[q] = b[(T / P) * q];
        } else if (P == 8 && e == 1) {  // n + 8 (n / 64), n = u + T q, T a multiple of 64
            const cf* __restrict__ b = buf + u + 8 * (u >> 6);
#pragma unroll
            for (int q = 0; q < P; q++) x[q] = b[(T + T / 8) * q];
        } else {
            const cf* __restrict__ b = buf + u;
#pragma unroll
            for (int q = 0; q < P; q++) x[q] = b[T * q];
        }
        return;
    }
    if (T % P == 0) {  // lds_pad(u + T*q) = lds_pad(u) + T*q + T*q/P
        const cf* __restrict__ b = buf + lds_pad<P>(u);
#pragma unroll
        for (int q = 0; q < P; q++) x[q] = b[T * q + (T * q) / P];
    } else {
#pragma unroll
        for (int q = 0; q < P; q++) x[q] = buf[lds_pad<P>(u + T * q)];
    }
}
// radix-P pass s (1 <= s < S): p = P^s
// ALLOW_POW = false keeps the table in every pass (the OceanRenderer kernels: their finite-difference normal amplifies
// transform rounding at ill-conditioned texels, and a frame is latency- not throughput-bound anyway)
template <int N, int P>
MW_HD void load_last(cf (&x)[P], int u, const cf* buf) { load_slots<N, P>(x, u, buf, FftGeom<N, P>::S - 1); }  // input of the final pass
// the arithmetic of radix-P pass s (twiddles + DFT), results left in registers: x[r] = element j + p r of the next sequence,
// j = ((u - k) << log2 P) + k, k = u mod p
template <int N, int P, int SGN, bool ALLOW_POW = true>
MW_HD void stage_regs(cf (&x)[P], int u, const Twiddles& tw, int s) {
    const int p = 1 << (LogP<P>::v * s);
    const int k = u & (p - 1);
    const cf* __restrict__ row = tw.TS[s] + k * (P + 1);
#ifdef MW_TW_POWERS
    constexpr bool POWERS = true;  // experiment: every pass by powers (+1 % step time at 1024^2: 24 VALU for 6 saved reads)
#else
    const bool POWERS = (ALLOW_POW && TwGeom<N, P>::POW_STAGE != 0 && s == TwGeom<N, P>::POW_STAGE);  // folds: s is unrolled
#endif
#if defined(MW_TW_SHUFFLE) && defined(__HIP_DEVICE_COMPILE__)
    // A/B experiment (BASELINE.json's "wavefront shuffle for the twiddle rows", profiles/r02_twiddle_shuffle_ab.json): the
    // first radix-8 pass needs e^{SGN 2 pi i r k / 64}, r k < 64: ONE 64-entry row held across the wave (lane l keeps entry l
    // in two VGPRs) and fetched by ds_bpermute -- two 32-bit permutes per complex twiddle instead of one ds_read_b64 of the
    // LDS table.  Both go through the LDS crossbar; the permute form issues twice as many DS instructions.
    if (P == 8 && s == 1) {
        const int lane = (int)__lane_id();
        float ws, wc;
        sincospif((float)lane * (1.0f / 32.0f), &ws, &wc);
        ws = SGN > 0 ? ws : -ws;
#pragma unroll
        for (int r = 1; r < P; r++) {
            const int src = (r * k) & 63;
            const cf w = mk(__shfl(wc, src), __shfl(ws, src));
            x[r] = cmul(x[r], w);
        }
    } else
#endif
    if (POWERS) {  // one table read per thread, the other P-2 twiddles as its powers (product tree <= log2 P deep)
        cf w[P];
        w[1] = tw.PW ? tw.PW[k] : row[1];
#pragma unroll
        for (int r = 2; r < P; r++) w[r] = cmul(w[r / 2], w[r - r / 2]);
#pragma unroll
        for (int r = 1; r < P; r++) x[r] = cmul(x[r], w[r]);
    } else {
#pragma unroll
        for (int r = 1; r < P; r++) x[r] = cmul(x[r], row[r]);
    }
    DftP<P, SGN>::run(x);
}
// When N = P^S (no final partial pass) the LAST radix-P pass has p = T and k = u: its results, element u + T r in slot r, are
// already the natural "thread u keeps element u + T q in slot q" order -- with the identity layout of the later exchanges a
// stage_store + load_last pair would write and read back the very same LDS words behind two barriers.  Kernels whose lane mapping
// does not change after the last pass skip that round trip (4096^2 with 16 points per thread, 512^2 with 8).
template <int N, int P>
struct LastInRegs { static constexpr bool value = MW_LAST_IN_REGS && FftGeom<N, P>::RL == 1 && FftGeom<N, P>::S >= 2 && XLay<N, P>::EXACT; };
// ---- the last exchange inside the wave (round 5; BASELINE.json's "wavefront shuffle") --------------------------------------------
// N = 1024 at 16 points per thread: T = 64, ONE WAVE carries a row, passes 16 x 16 x 4.  After the second radix-16 pass lane u = k + 16 c
// (k = u mod 16, c = its 16-lane row of the wave) holds in slot r the element n = 256 c + 16 r + k of the last intermediate sequence, and the
// final radix-4 pass of thread u' wants n = u' + 64 q, i.e. u' = k + 16 (r mod 4), q = 4 c + r / 4: the value moves from row c of the wave to
// row r mod 4 and nowhere else -- a 4 x 4 TRANSPOSITION between the wave's four 16-lane rows and the low two bits of the slot index.  gfx950
// has the instruction for exactly that: v_permlane16_swap_b32 (odd rows of one register <-> even rows of another) and v_permlane32_swap_b32
// (upper half of one <-> lower half of another) swap one lane bit with one register bit each, two VALU moves per complex value and bit, no
// LDS crossbar (ds_bpermute, the form that lost in round 2, goes through it), no barrier.  32 moves replace 16 ds_write_b64 + 16 ds_read_b64
// and two barriers per transform; after them slot 4 (q mod 4) + q / 4 holds what load_last would have put in slot q -- a renaming.  The
// arithmetic is untouched: the same passes, the same table twiddles, the same bits as the LDS exchange it replaces.
#ifndef MW_LAST_IN_WAVE
#define MW_LAST_IN_WAVE 1
#endif
template <int N, int P>
struct LastInWave {
    static constexpr bool value = MW_LAST_IN_WAVE && P == 16 && FftGeom<N, P>::T == 64 && FftGeom<N, P>::S == 2 && FftGeom<N, P>::RL == 4 && XLay<N, P>::EXACT;
};
// where lane l's slot rho ends up / comes from under the transposition (it is an involution): used by the host emulation and the device test
MW_HD void wave_transpose4_source(int lane, int rho, int* src_lane, int* src_rho) {
    *src_lane = (lane & 15) | ((rho & 3) << 4);
    *src_rho = (rho & ~3) | (lane >> 4);
}
#if defined(__HIP_DEVICE_COMPILE__)
// (the two results go through named scalars: __builtin_bit_cast applied DIRECTLY to a vector element, bit_cast(float, r[1]), reads element 0
// in this clang -- both outputs became the first one; tests/test_gpu_parity.py::test_wave_transpose4_instructions_match_the_index_map)
__device__ __forceinline__ void mw_swap16(float& a, float& b) {  // odd 16-lane rows of a <-> even rows of b
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void mw_swap32(float& a, float& b) {  // upper 32 lanes of a <-> lower 32 lanes of b
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
#endif
// all 64 lanes of the wave must be active.  Host (tests/emul): a no-op -- the emulation moves the values between its thread states itself
// (wave_transpose4_source), the phase functions on either side are the same code.
template <int P>
MW_HD void wave_transpose4(cf (&x)[P]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int r = 0; r < P; r++)
        if ((r & 1) == 0) { mw_swap16(x[r].x, x[r | 1].x); mw_swap16(x[r].y, x[r | 1].y); }  // slot bit 0 <-> lane bit 4
#pragma unroll
    for (int r = 0; r < P; r++)
        if ((r & 2) == 0) { mw_swap32(x[r].x, x[r | 2].x); mw_swap32(x[r].y, x[r | 2].y); }  // slot bit 1 <-> lane bit 5
#else
    (void)x;
#endif
}
// the last radix-P pass of a transform whose results stay in registers: LastInRegs (N = P^S: nothing to move) or LastInWave (the exchange
// behind it runs inside the wave); load_last_regs is the matching "input of the final pass"
template <int N, int P>
struct LastStays { static constexpr bool value = LastInRegs<N, P>::value || LastInWave<N, P>::value; };
// THE predicate "radix-P pass s leaves its results in registers": the phase functions (p2_mid_store / p2_last_load, p1_finish, the halo and
// chirp-z transforms) and the kernels' barrier elision all ask this one function, so a caller cannot skip the LDS write and still read
// the exchange afterwards.  A kernel that keeps its barriers anyway (ping-pong buffers, MW_DBUF) stays correct: the phase
// functions alone decide where the data is.
template <int N, int P>
MW_HD constexpr bool mw_pass_in_regs(int s) { return LastStays<N, P>::value && s == FftGeom<N, P>::S - 1; }
template <int N, int P, int SGN, bool ALLOW_POW = true>
MW_HD void stage_last_regs(cf (&x)[P], int u, const Twiddles& tw, int s) {
    stage_regs<N, P, SGN, ALLOW_POW>(x, u, tw, s);
    if (LastInWave<N, P>::value) wave_transpose4<P>(x);
}
template <int N, int P>
MW_HD void load_last_regs(cf (&x)[P]) {
    if (LastInWave<N, P>::value) {  // slot q of the final pass = slot 4 (q mod 4) + q / 4 after the transposition
        cf y[P];
#pragma unroll
        for (int q = 0; q < P; q++) y[q] = x[4 * (q & 3) + (q >> 2)];
#pragma unroll
        for (int q = 0; q < P; q++) x[q] = y[q];
    }
}
template <int N, int P, int SGN, bool ALLOW_POW = true>
MW_HD void stage_store(cf (&x)[P], int u, cf* buf, const Twiddles& tw, int s) {
    stage_regs<N, P, SGN, ALLOW_POW>(x, u, tw, s);
    const int p = 1 << (LogP<P>::v * s);
    const int k = u & (p - 1);
    const int j = ((u - k) << LogP<P>::v) + k;
    if (XLay<N, P>::EXACT) {
        cf* __restrict__ b = buf + j + ((P == 8 && s == 1) ? 8 * (j >> 6) : 0);  // P = 8, s = 1: j + 8 r stays inside its 64-block
#pragma unroll
        for (int r = 0; r < P; r++) b[p * r] = x[r];
        return;
    }
    cf* __restrict__ b = buf + lds_pad<P>(j);  // p is a multiple of P: lds_pad(j + p*r) = lds_pad(j) + p*r + p*r/P
#pragma unroll
    for (int r = 0; r < P; r++) b[p * r + ((p * r) >> LogP<P>::v)] = x[r];
}
template <int N, int P, int SGN>
MW_HD void final_stage(cf (&x)[P], int u, const cf* __restrict__ TF) {
    constexpr int RL = FftGeom<N, P>::RL, NB = FftGeom<N, P>::NB;
    if (RL == 1) return;
    cf tw[RL];
#ifndef MW_TF_POWERS_MIN_RL
#define MW_TF_POWERS_MIN_RL 8  // a radix-8 final pass (N = 2048, 128 at 16 points) loads ONE twiddle, w = e^{SGN 2 pi i u/N}, and
                               // forms w^2..w^7 by products: 7 loads and 12 VGPRs less -- what lets the 2048^2 pass 2 keep
                               // the early halo fetch (pass 2 -5.5 %, pass 1 -4 %)
#endif
    if (RL >= MW_TF_POWERS_MIN_RL) {
        tw[1] = TF[FftGeom<N, P>::T + u];
#pragma unroll
        for (int r = 2; r < RL; r++) tw[r] = cmul(tw[r / 2], tw[r - r / 2]);
    } else {
#pragma unroll
        for (int r = 1; r < RL; r++) tw[r] = TF[r * FftGeom<N, P>::T + u];
    }
#pragma unroll
    for (int m = 0; m < NB; m++) {
#pragma unroll
        for (int r = 1; r < RL; r++) x[m + r * NB] = DftP<P, SGN>::rot(cmul(x[m + r * NB], tw[r]), (r * m) & (P - 1));
        if (RL == 2) {
            dft2<SGN>(x[m], x[m + NB]);
        } else if (RL == 4) {
            dft4<SGN>(x[m], x[m + NB], x[m + 2 * NB], x[m + 3 * NB]);
        } else {  // RL == 8 (only with P == 16)
            cf y[8];
#pragma unroll
            for (int r = 0; r < 8; r++) y[r] = x[(m + r * NB) & (P - 1)];
            dft8<SGN>(y);
#pragma unroll
            for (int r = 0; r < 8; r++) x[(m + r * NB) & (P - 1)] = y[r];
        }
    }
}

#if MW_FFT_STRICT && defined(__clang__)
#pragma clang fp contract(fast)
#endif
}  // namespace mw
#include <vector>
namespace mw {
// Host side: the concatenated twiddle table [TS1 | TS2 | TS3 | TF] in the layout of TwGeom<N, P>, formed in double and rounded
// once to float32 (one definition for the library's upload and for tests/emul).
inline std::vector<cf> build_twiddle_table_host(int N, int P, int sgn) {
    const double two_pi = 6.283185307179586476925286766559;
    const int T = N / P;
    int S = 0;
    long long PS = 1;
    while (PS * P <= N) { PS *= P; S++; }
    const int RL = (int)(N / PS);
    std::vector<cf> tab;
    auto push = [&](double num, double den) {
        const double a = sgn * two_pi * num / den;
        tab.push_back(mk((float)cos(a), (float)sin(a)));
    };
    for (int s = 1; s < S; s++) {
        long long p = 1;
        for (int i = 0; i < s; i++) p *= P;
        for (long long k = 0; k < p; k++)
            for (int r = 0; r < P + 1; r++) push((double)((r % P) * k), (double)(p * P));  // row stride P+1 (one pad entry): see Twiddles
    }
    if (RL > 1)
        for (int r = 0; r < RL; r++)
            for (int u = 0; u < T; u++) push((double)r * u, (double)N);
    if (tab.empty()) tab.push_back(mk(1.f, 0.f));
    return tab;
}
// ---- spectrum algebra (DESIGN.md "Hermitian packing") ---------------------------------------
// h(k,t) = P e^{i th} + Q e^{-i th}
// The four products of each component as ONE product and three FMAs written out (round 5): with hipcc's -ffp-contract=fast the optimiser
// decides per inlined copy which of `a*b + c*d + ...` it fuses, and two launch forms of one kernel that reach this function through
// differently shaped loops (or_p1_animate: point by point in the lone frame, eight points at a time in the batched forms) produced spectra
// that differed in the last bit -- a batched OceanRenderer handle was no longer bit-identical to its single handles.  Same policy as cmul /
// MW_FFT_STRICT: the rounding is fixed in the source.
#ifndef MW_ANIMATE_EXPLICIT
#define MW_ANIMATE_EXPLICIT 1  // 0: the plain expression of rounds 1-4 (A/B of the speed only: its bits depend on the inlining context)
#endif
MW_HD cf animate(float px, float py, float qx, float qy, float c, float s) {
    if (!MW_ANIMATE_EXPLICIT) return mk(px * c - py * s + qx * c + qy * s, px * s + py * c - qx * s + qy * c);
    const float re = __builtin_fmaf(px, c, __builtin_fmaf(-py, s, __builtin_fmaf(qx, c, smul(qy, s))));
    const float im = __builtin_fmaf(px, s, __builtin_fmaf(py, c, __builtin_fmaf(-qx, s, smul(qy, c))));
    return mk(re, im);
}

}  // namespace mw
