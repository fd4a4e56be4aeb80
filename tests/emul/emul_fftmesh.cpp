// tests/emul/emul_fftmesh.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Host-side lock-step emulation of the FFTMesh-semantics kernels: the very same MW_HD phase functions
// that mistral_water.hip launches on the GPU (mistral-water_amd/csrc/fftmesh_kernels.h) are stepped
// here block by block, phase by phase, thread by thread, with a plain array standing in for LDS and a
// loop boundary standing in for __syncthreads().  It lets the CPU-only test tier (-m "not gpu") verify
// the thread/LDS choreography, the Hermitian packing and every index map against the oracle without
// a GPU.  It is never part of libmistral_water.so and is not a fallback: nothing in the product loads it.
//
// build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared (tests/emul/build.py)
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../mistral-water_amd/csrc/fftmesh_kernels.h"
#include "../../mistral-water_amd/csrc/gerstner_kernels.h"
#include "../../mistral-water_amd/csrc/pond_kernels.h"
#include "../../mistral-water_amd/csrc/ocean_renderer_kernels.h"
#include "../../mistral-water_amd/csrc/czt_kernels.h"

using namespace mw;

namespace {

// concatenated twiddle table in the layout of TwGeom<N,P> (mirrors build_twiddle_table in mistral_water.hip)
struct Tables {
    std::vector<cf> TW, Wpre;
    Tables(int N, int P, int sgn = +1) : Wpre(2 * N) {
        TW = build_twiddle_table_host(N, P, sgn);
        for (int m = 0; m < 2 * N; m++) {
            double a = M_PI * (double)m / (double)N;
            double sg = (m & 1) ? -1.0 : 1.0;
            Wpre[m] = mk((float)(sg * cos(a)), (float)(sg * sin(a)));
        }
    }
};

// The in-wave exchange of LastInWave (mw_math.h): on the device v_permlane16_swap / v_permlane32_swap move the values between the lanes of
// a wave; wave_transpose4 is a no-op on the host, so the emulation moves them between its thread states -- threads first .. first + count
// (whole waves of 64), x_of(tid) = that thread's slot array -- by the index map the device test checks the instructions against.
template <int N, int P, class F>
void emul_last_in_wave(int first, int count, F x_of) {
    if (!LastInWave<N, P>::value) return;
    std::vector<cf> old((size_t)64 * P);
    for (int w0 = first; w0 < first + count; w0 += 64) {
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < P; r++) old[(size_t)l * P + r] = x_of(w0 + l)[r];
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < P; r++) {
                int sl, sr;
                wave_transpose4_source(l, r, &sl, &sr);
                x_of(w0 + l)[r] = old[(size_t)sl * P + sr];
            }
    }
}

template <int N, int P>
void run_pass1(const P1Args& A, const StepTimes& tm, int nsteps) {
    constexpr int T = FftGeom<N, P>::T, NT = P1Geom<N, P>::NTHREADS, BS = P1Geom<N, P>::BUFSTRIDE;
    std::vector<cf> lds(P1Geom<N, P>::CW * BS);
    const Twiddles tw = TwGeom<N, P>::view(A.TW);
    struct St { P1State<P> s; cf x[P]; };
    std::vector<St> st(NT);
    for (int step = 0; step < nsteps; step++)
        for (int jb = 0; jb < P1Geom<N, P>::GRID_X; jb++) {
            const float t = tm.t[step];
            for (int tid = 0; tid < NT; tid++) p1_animate<N, P>(A, jb, tid, t, st[tid].s);
            for (int f = 0; f < 3; f++) {
                if (!p1_field_active(N, jb, f, P1Geom<N, P>::CW)) continue;
                for (int tid = 0; tid < NT; tid++) {
                    p1_build<N, P>(A, jb, tid, f, st[tid].s, st[tid].x);
                    stage0_store<N, P, +1>(st[tid].x, tid % T, lds.data() + (tid / T) * BS);
                }
                for (int s = 1; s < p1_mid_passes<N, P>(); s++) {
                    for (int tid = 0; tid < NT; tid++) load_slots<N, P>(st[tid].x, tid % T, lds.data() + (tid / T) * BS, s - 1);
                    for (int tid = 0; tid < NT; tid++)
                        stage_store<N, P, +1>(st[tid].x, tid % T, lds.data() + (tid / T) * BS, tw, s);
                }
                for (int tid = 0; tid < NT; tid++) p1_finish<N, P>(A, tw, jb, step, tid, f, st[tid].x, lds.data());
            }
        }
}

static bool g_force_hs = false;  // emul_set_variant: run the sequential-halo pass 2 regardless of Plan<N>::HS
static bool g_frame = false;     // emul_set_variant(2): the frame variant (k_pass2_frame) wherever its geometry exists

template <int N, int P, int R2>
void run_pass2(const P2Args& A, int nsteps) {
    constexpr int T = FftGeom<N, P>::T, NT = P2Geom<N, P, R2>::NTHREADS, BS = P2Geom<N, P, R2>::BUFSTRIDE;
    std::vector<cf> lds((R2 + 1) * BS);
    std::vector<float> noise((size_t)R2 * N);
    const Twiddles tw = TwGeom<N, P>::view(A.TW);
    struct St { P2State<P> s; cf x[P]; };
    std::vector<St> st(NT);
    for (int step = 0; step < nsteps; step++)
        for (int ab = 0; ab < N / R2; ab++) {
            for (int k = 0; k < 3; k++) {
                const int f = p2_field(k);
                for (int tid = 0; tid < NT; tid++)
                    if (p2_active<N, P, R2>(ab, tid, f)) p2_load<N, P, R2>(A, ab, step, tid, f, st[tid].x, lds.data());
                for (int s = 1; s < FftGeom<N, P>::S; s++) {
                    for (int tid = 0; tid < NT; tid++)
                        if (p2_active<N, P, R2>(ab, tid, f)) p2_mid_load<N, P, R2>(tid, s, st[tid].x, lds.data());
                    for (int tid = 0; tid < NT; tid++)
                        if (p2_active<N, P, R2>(ab, tid, f)) p2_mid_store<N, P, R2>(tw, tid, s, st[tid].x, lds.data());
                    if (s == FftGeom<N, P>::S - 1) emul_last_in_wave<N, P>(0, NT, [&](int tid) { return st[tid].x; });
                }
                for (int tid = 0; tid < NT; tid++)
                    if (p2_active<N, P, R2>(ab, tid, f))
                        p2_finish<N, P, R2>(A, tw, ab, step, tid, f, st[tid].x, st[tid].s, lds.data(), noise.data());
            }
            for (int tid = 0; tid < NT; tid++)
                if (p2_active<N, P, R2>(ab, tid, 1)) p2_publish_hds<N, P, R2>(tid, st[tid].s, lds.data());
            for (int tid = 0; tid < NT; tid++)
                if (tid / T < R2) p2_epilogue<N, P, R2>(A, ab, step, tid, st[tid].s, lds.data(), noise.data());
        }
}

// the sequential-halo variant (k_pass2_hs), phase by phase; runs at any N here so that small grids exercise it too
template <int N, int P, int R2>
void run_pass2_hs(const P2Args& A, int nsteps) {
    constexpr int T = FftGeom<N, P>::T, NT = P2Geom<N, P, R2, true>::NTHREADS, BS = P2Buf<N, P>::BUFSTRIDE;
    std::vector<cf> lds(R2 * BS);
    const Twiddles tw = TwGeom<N, P>::view(A.TW);
    struct St { P2StateHS<P> s; cf x[P]; cf t1m[P / 2]; };
    std::vector<St> st(NT);
    constexpr bool KEEP = KeepT1<N, P>::value && P2SlopeParts<N, P>::value;  // as k_pass2_hs: raw mirrored height values kept for the slopes
    for (int step = 0; step < nsteps; step++)
        for (int ab = 0; ab < N / R2; ab++)
            for (int k = 0; k < 3; k++) {
                const int f = p2_hs_field(k);
                for (int tid = 0; tid < NT; tid++)
                    p2_load<N, P, R2>(A, ab, step, tid, f, st[tid].x, lds.data(), (KEEP && f != 1) ? st[tid].t1m : nullptr);
                for (int s = 1; s < FftGeom<N, P>::S; s++) {
                    for (int tid = 0; tid < NT; tid++) p2_mid_load<N, P, R2>(tid, s, st[tid].x, lds.data());
                    for (int tid = 0; tid < NT; tid++) p2_mid_store<N, P, R2>(tw, tid, s, st[tid].x, lds.data());
                    if (s == FftGeom<N, P>::S - 1) emul_last_in_wave<N, P>(0, NT, [&](int tid) { return st[tid].x; });
                }
                if (f == 2) {
                    for (int tid = 0; tid < NT; tid++)
                        p2_hs_finish_slopes<N, P, R2>(A, tw, ab, step, tid, st[tid].x, st[tid].s, lds.data());
                    break;
                }
                for (int tid = 0; tid < NT; tid++) p2_hs_finish<N, P, R2>(tw, ab, tid, f, st[tid].x, st[tid].s, lds.data());
                if (f != 1) continue;
                for (int tid = 0; tid < NT; tid++) p2_vertices<N, P, R2>(A, ab, step, tid, st[tid].s);
                for (int tid = 0; tid < NT; tid++) p2_publish_hds<N, P, R2>(tid, st[tid].s, lds.data());
                for (int tid = 0; tid < NT; tid++)
                    if (tid / T != R2 - 1)
                        p2_hs_jacobian<N, P, R2>(ab, tid, st[tid].s, lds.data() + (tid / T) * BS, lds.data() + (tid / T + 1) * BS);
                if (ab * R2 + R2 < N) {
                    for (int u = 0; u < T; u++) {
                        p2_hs_halo_fetch<N, P, R2>(A, ab, step, u, st[u].x);
                        stage0_store<N, P, +1>(st[u].x, u, lds.data());
                    }
                    for (int s = 1; s < FftGeom<N, P>::S; s++) {
                        for (int u = 0; u < T; u++) load_slots<N, P>(st[u].x, u, lds.data(), s - 1);
                        const bool in_regs = mw_pass_in_regs<N, P>(s);  // as k_pass2_hs's halo transform
                        for (int u = 0; u < T; u++) {
                            if (in_regs) stage_last_regs<N, P, +1>(st[u].x, u, tw, s);
                            else stage_store<N, P, +1>(st[u].x, u, lds.data(), tw, s);
                        }
                        if (in_regs) emul_last_in_wave<N, P>(0, T, [&](int tid) { return st[tid].x; });
                    }
                    for (int u = 0; u < T; u++) { p2_last_load<N, P>(st[u].x, u, lds.data()); final_stage<N, P, +1>(st[u].x, u, tw.TF); }
                    for (int u = 0; u < T; u++) p2_hs_halo_publish<N, P, R2>(ab, u, st[u].x, lds.data());
                }
                for (int tid = (R2 - 1) * T; tid < NT; tid++)
                    p2_hs_jacobian_lds<N, P, R2>(ab, tid, st[tid].s, lds.data() + (R2 - 1) * BS, lds.data());
            }
}

// the frame variant (k_pass2_frame): 3 R2 + 1 row groups, the three fields and the halo row side by side
template <int N, int P, int R2>
void run_pass2_frame(const P2Args& A, int nsteps) {
    using G = P2FrameGeom<N, P, R2>;
    constexpr int T = G::T, FT = G::FT, NT = G::NTHREADS, BS = G::BUFSTRIDE, SS = G::SETSTRIDE;
    std::vector<cf> lds(3 * SS + BS);
    const Twiddles tw = TwGeom<N, P>::view(A.TW);
    struct St { cf x[P]; };
    std::vector<St> st(NT);
    cf* set_d = lds.data() + SS;
    cf* set_s = lds.data() + 2 * SS;
    cf* hbuf = lds.data() + 3 * SS;
    for (int step = 0; step < nsteps; step++)
        for (int ab = 0; ab < N / R2; ab++) {
            const bool has_halo = ab * R2 + R2 < N;
            auto each = [&](auto&& fn) {
                for (int tid = 0; tid < NT; tid++) {
                    const int fg = tid / FT, tl = tid - fg * FT;
                    if (fg == 3 && !has_halo) continue;
                    fn(fg, tl, st[tid], lds.data() + fg * SS);
                }
            };
            each([&](int fg, int tl, St& t, cf* mine) {
                if (fg < 3) { p2_fetch<N, P, R2>(A, ab, step, tl, fg, t.x); p2_stage0<N, P, R2>(tl, t.x, mine); }
                else { p2_hs_halo_fetch<N, P, R2>(A, ab, step, tl, t.x); stage0_store<N, P, +1>(t.x, tl, mine); }
            });
            for (int s = 1; s < FftGeom<N, P>::S; s++) {
                const bool in_regs = mw_pass_in_regs<N, P>(s);
                each([&](int fg, int tl, St& t, cf* mine) {
                    if (fg < 3) p2_mid_load<N, P, R2>(tl, s, t.x, mine); else load_slots<N, P>(t.x, tl, mine, s - 1);
                });
                each([&](int fg, int tl, St& t, cf* mine) {
                    if (fg < 3) p2_mid_store<N, P, R2>(tw, tl, s, t.x, mine);
                    else if (in_regs) stage_last_regs<N, P, +1>(t.x, tl, tw, s);
                    else stage_store<N, P, +1>(t.x, tl, mine, tw, s);
                });
                if (in_regs) emul_last_in_wave<N, P>(0, has_halo ? NT : 3 * FT, [&](int tid) { return st[tid].x; });
            }
            each([&](int fg, int tl, St& t, cf* mine) {
                p2_last_load<N, P>(t.x, tl % T, mine + (fg < 3 ? tl / T : 0) * BS);
                final_stage<N, P, +1>(t.x, tl % T, tw.TF);
            });
            each([&](int fg, int tl, St& t, cf*) {
                if (fg == 1) p2_frame_hds<N, P, R2>(ab, tl, t.x, set_d);
                else if (fg == 2) p2_frame_normals<N, P, R2>(A, ab, step, tl, t.x, set_s);
                else if (fg == 3) p2_hs_halo_publish<N, P, R2>(ab, tl, t.x, hbuf);
            });
            each([&](int fg, int tl, St& t, cf*) {
                if (fg == 0) p2_frame_vertices<N, P, R2>(A, ab, step, tl, t.x, set_d);
                else if (fg == 1) p2_frame_white<N, P, R2>(A, ab, step, tl, set_d, hbuf, set_s);
            });
        }
}

// PA / PB: points per thread of pass 1 / pass 2 (the exchange-buffer layout does not depend on them)
template <int N, int PA, int PB>
int evaluate_np(const OceanConsts& C, const cf* h0, const cf* h0c, const float* times, int nsteps, float* vertices,
                float* normals, float* white, int white_stride) {
    constexpr int R2 = Plan<N>::R2;
    const bool hs = g_force_hs || Plan<N>::HS;
    Tables tb(N, PA), tb2(N, PB);
    std::vector<f4> PQt((size_t)N * N), d_i0(N), d_j0(N);
    std::vector<float> Om((size_t)N * N);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            prep_element(N, C.length, C.gravity, i, j, h0, h0c, tb.Wpre.data(), PQt.data(), d_i0.data(), d_j0.data(), Om.data());
    std::vector<cf> E((size_t)nsteps * 3 * N * N), Cj0((size_t)nsteps * 3 * N);
    P1Args A1;
    A1.PQt = PQt.data(); A1.dPQ_i0 = d_i0.data(); A1.dPQ_j0 = d_j0.data(); A1.Om = Om.data(); A1.TW = tb.TW.data();
    A1.Cj0 = Cj0.data(); A1.E = E.data(); A1.c = C;
    StepTimes tm;
    for (int k = 0; k < nsteps; k++) tm.t[k] = times[k];
    run_pass1<N, PA>(A1, tm, nsteps);
    P2Args A2;
    A2.E = E.data(); A2.Cj0 = Cj0.data(); A2.TW = tb2.TW.data(); A2.vertices = vertices; A2.normals = normals; A2.white = white;
    A2.white_stride = white_stride; A2.c = C;
    if (g_frame) {
        constexpr int RF = mw_frame_r2(N);      // the product's rows per workgroup of the frame variant
        if constexpr (P2FrameGeom<N, PB, RF>::FITS) run_pass2_frame<N, PB, RF>(A2, nsteps);
        else return 5;
    } else if (hs) {
        run_pass2_hs<N, PB, R2>(A2, nsteps);
    } else {
        if constexpr ((R2 + 1) * (N / PB) <= 1024) run_pass2<N, PB, R2>(A2, nsteps);  // the halo-group kernel exists for this plan
        else return 4;
    }
    return 0;
}

// pts = 0: the product's plan (Plan<N>::P1 for pass 1, Plan<N>::P2 for pass 2); 8 / 16: both passes with that many points
// per thread where the geometry allows it
template <int N>
int evaluate_n(int pts, const OceanConsts& C, const cf* h0, const cf* h0c, const float* times, int nsteps, float* vertices,
               float* normals, float* white, int white_stride) {
    if (pts == 0) return evaluate_np<N, Plan<N>::P1, Plan<N>::P2>(C, h0, h0c, times, nsteps, vertices, normals, white, white_stride);
    if (pts == 16) return evaluate_np<N, 16, 16>(C, h0, h0c, times, nsteps, vertices, normals, white, white_stride);
    if constexpr (N <= 1024) {
        if (pts == 8) return evaluate_np<N, 8, 8>(C, h0, h0c, times, nsteps, vertices, normals, white, white_stride);
    }
    return 3;
}

template <int N, int P>
int fft1d_np(const float* in_xy, float* out_xy) {
    constexpr int T = FftGeom<N, P>::T;
    Tables tb(N, P);
    Twiddles tw = TwGeom<N, P>::view(tb.TW.data());
    std::vector<cf> lds(FftGeom<N, P>::LBUF + 8);
    struct S { cf x[P]; };
    std::vector<S> st(T);
    for (int u = 0; u < T; u++)
        for (int q = 0; q < P; q++) st[u].x[q] = mk(in_xy[2 * (u + T * q)], in_xy[2 * (u + T * q) + 1]);
    for (int u = 0; u < T; u++) stage0_store<N, P, +1>(st[u].x, u, lds.data());
    for (int s = 1; s < FftGeom<N, P>::S; s++) {
        for (int u = 0; u < T; u++) load_slots<N, P>(st[u].x, u, lds.data(), s - 1);
        for (int u = 0; u < T; u++) stage_store<N, P, +1>(st[u].x, u, lds.data(), tw, s);
    }
    for (int u = 0; u < T; u++) { load_last<N, P>(st[u].x, u, lds.data()); final_stage<N, P, +1>(st[u].x, u, tw.TF); }
    for (int u = 0; u < T; u++)
        for (int q = 0; q < P; q++) { out_xy[2 * (u + T * q)] = st[u].x[q].x; out_xy[2 * (u + T * q) + 1] = st[u].x[q].y; }
    return 0;
}

// ---- OceanRenderer semantics: one GenerateTexture() stepped through the kernels' phase functions ----------
// packed: the two-transform plan of planar-texture calls (k_or_pass1_packed / k_or_pass2_packed: height + i Dz from ONE transform)
template <int N, int P>
int or_step_np(const OrConsts& C, float dt, const f4* initT, float* phaseT, float* height, cf* disp, float* disp_g,
               float* normal, float* white, float* height_g, float* disp_a, bool packed) {
    constexpr int T = FftGeom<N, P>::T;
    Tables tb(N, P, -1);
    const Twiddles tw = TwGeom<N, P>::view(tb.TW.data());
    std::vector<cf> E((size_t)3 * N * N);
    std::vector<f4> PQT((size_t)N * N);
    if (packed)
        for (int px = 0; px < N; px++)
            for (int py = 0; py < N; py++) or_prep_element(N, px, py, initT, PQT.data());
    const int nf = packed ? 2 : 3;
    {
        OrP1Args A;
        A.stream_E = 0;
        A.PQT = PQT.data();
        std::vector<float> phase_next((size_t)N * N), om((size_t)N * N);
        for (int px = 0; px < N; px++)
            for (int py = 0; py < N; py++) om[(size_t)px * N + py] = or_omega(C, px, py);
        A.omT = om.data();
        A.initT = initT; A.phase_in = phaseT; A.phase_out = phase_next.data(); A.TW = tb.TW.data(); A.E = E.data(); A.c = C; A.dt = dt;
        constexpr int NT = OrP1Geom<N, P>::NTHREADS, BS = OrP1Geom<N, P>::BUFSTRIDE;
        std::vector<cf> lds(P1Geom<N, P>::CW * BS);
        struct St { cf h[P]; cf hh[P]; cf x[P]; };
        std::vector<St> st(NT);
        for (int jb = 0; jb < N / 4; jb++)
            for (int f = 0; f < nf; f++) {  // grid (N/4, 3 | 2): one field per block
                for (int tid = 0; tid < NT; tid++) {
                    if (packed) or_p1_animate_packed<N, P, 8>(A, jb, tid, f == 0, f == 0, f == 1, st[tid].h, st[tid].hh);
                    else or_p1_animate<N, P, 8>(A, jb, tid, f == 0, st[tid].h);
                }
                for (int tid = 0; tid < NT; tid++) {
                    if (packed) or_p1_build_packed<N, P>(A, jb, tid, f, st[tid].h, st[tid].hh, st[tid].x);
                    else or_p1_build<N, P>(A, jb, tid, f, st[tid].h, st[tid].x);
                    stage0_store<N, P, -1>(st[tid].x, tid % T, lds.data() + (tid / T) * BS);
                }
                for (int s = 1; s < FftGeom<N, P>::S; s++) {
                    for (int tid = 0; tid < NT; tid++) load_slots<N, P>(st[tid].x, tid % T, lds.data() + (tid / T) * BS, s - 1);
                    for (int tid = 0; tid < NT; tid++) stage_store<N, P, -1, false>(st[tid].x, tid % T, lds.data() + (tid / T) * BS, tw, s);
                }
                for (int tid = 0; tid < NT; tid++) or_p1_finish<N, P>(A, tw, jb, tid, f, st[tid].x, lds.data());
            }
        std::memcpy(phaseT, phase_next.data(), sizeof(float) * (size_t)N * N);  // the ping-pong swap
    }
    {
        OrP2Args A;
        A.E = E.data(); A.TW = tb.TW.data(); A.height = height; A.disp = disp; A.disp_g = disp_g; A.c = C;
        A.height_g = height_g; A.disp_a = disp_a;
        constexpr int NT = OrP2Geom<N, P>::NTHREADS, BS = OrP2Geom<N, P>::BUFSTRIDE;
        std::vector<cf> lds(P1Geom<N, P>::CW * BS);
        struct St { cf x[P]; float dx[P]; };
        std::vector<St> st(NT);
        for (int ab = 0; ab < N / 4; ab++)
            for (int k = 0; k < nf; k++) {
                const int f = packed ? (k == 0 ? 1 : 3) : or_p2_field(k);            // what the finish stores
                const int plane = packed ? k : f;                                     // where pass 1 put it
                for (int tid = 0; tid < NT; tid++) or_p2_load<N, P>(A, ab, tid, plane, st[tid].x, lds.data());
                for (int s = 1; s < FftGeom<N, P>::S; s++) {
                    constexpr bool EX = XLay<N, P>::EXACT;
                    constexpr int T2 = FftGeom<N, P>::T;
                    for (int tid = 0; tid < NT; tid++)
                        load_slots<N, P>(st[tid].x, EX ? tid % T2 : tid >> 2, lds.data() + (EX ? tid / T2 : tid & 3) * BS, s - 1);
                    for (int tid = 0; tid < NT; tid++)
                        stage_store<N, P, -1, false>(st[tid].x, EX ? tid % T2 : tid >> 2, lds.data() + (EX ? tid / T2 : tid & 3) * BS, tw, s);
                }
                for (int tid = 0; tid < NT; tid++) or_p2_finish<N, P>(A, tw, ab, tid, f, st[tid].x, st[tid].dx, lds.data());
            }
    }
    for (int py = 0; py < N; py++)
        for (int px = 0; px < N; px++) or_normal_element(C, px, py, height, disp, disp_g, normal);
    for (int py = 0; py < N; py++)
        for (int px = 0; px < N; px++) or_white_element(C, px, py, disp, normal, white);
    return 0;
}

// ---- chirp-z form of the separable sum (czt_kernels.h): one launch of k_czt stepped phase by phase -------------------------
// workgroup = RW rows x T threads; phases separated by the kernel's barriers; per-thread registers persist across phases
template <int M, int P>
int czt_pass_np(CztArgs A, int nfields) {
    constexpr int T = M / P, RW = czt_rows(M), BUF = FftGeom<M, P>::LBUF + 4, NT = RW * T;
    Tables tf(M, P, -1), ti(M, P, +1);
    A.TWf = tf.TW.data();
    A.TWi = ti.TW.data();
    const Twiddles twf = TwGeom<M, P>::view(A.TWf), twi = TwGeom<M, P>::view(A.TWi);
    struct S { cf x[P]; };
    std::vector<S> st(NT);
    std::vector<cf> lds((size_t)RW * BUF);
    for (int f = 0; f < nfields; f++)
        for (int rb = 0; rb * RW < A.rows; rb++) {
            auto all = [&](auto&& body) {
                for (int tid = 0; tid < NT; tid++) {
                    const int w = tid / T, u = tid % T, row = rb * RW + w;
                    body(st[tid].x, u, row, row < A.rows, lds.data() + (size_t)w * BUF);
                }
            };
            all([&](cf (&x)[P], int u, int row, bool live, cf* buf) { czt_load<M, P>(A, f, row, u, live, x); stage0_store<M, P, -1>(x, u, buf); });
            for (int s = 1; s < FftGeom<M, P>::S; s++) {
                all([&](cf (&x)[P], int u, int, bool, cf* buf) { load_slots<M, P>(x, u, buf, s - 1); });
                all([&](cf (&x)[P], int u, int, bool, cf* buf) { stage_store<M, P, -1, false>(x, u, buf, twf, s); });
            }
            all([&](cf (&x)[P], int u, int, bool, cf* buf) { load_last<M, P>(x, u, buf); final_stage<M, P, -1>(x, u, twf.TF); czt_mul_kernel<M, P>(A, u, x); });
            all([&](cf (&x)[P], int u, int, bool, cf* buf) { stage0_store<M, P, +1>(x, u, buf); });
            for (int s = 1; s < FftGeom<M, P>::S; s++) {
                all([&](cf (&x)[P], int u, int, bool, cf* buf) { load_slots<M, P>(x, u, buf, s - 1); });
                all([&](cf (&x)[P], int u, int, bool, cf* buf) { stage_store<M, P, +1, false>(x, u, buf, twi, s); });
            }
            all([&](cf (&x)[P], int u, int row, bool live, cf* buf) {
                load_last<M, P>(x, u, buf);
                final_stage<M, P, +1>(x, u, twi.TF);
                if (live) czt_store<M, P>(A, f, row, u, x);
            });
        }
    return 0;
}
int czt_pass(int M, const CztArgs& A, int nfields) {
    switch (M) {
        case 64: return czt_pass_np<64, czt_points(64)>(A, nfields);
        case 128: return czt_pass_np<128, czt_points(128)>(A, nfields);
        case 256: return czt_pass_np<256, czt_points(256)>(A, nfields);
        case 512: return czt_pass_np<512, czt_points(512)>(A, nfields);
        case 1024: return czt_pass_np<1024, czt_points(1024)>(A, nfields);
        case 2048: return czt_pass_np<2048, czt_points(2048)>(A, nfields);
        case 4096: return czt_pass_np<4096, czt_points(4096)>(A, nfields);
        default: return 1;
    }
}

}  // namespace

extern "C" {

// returns 0 on success, 1 for an unsupported N, 3 for an unsupported (N, pts) pair
int emul_fftmesh_evaluate(int N, int pts, float unit_width, float length, float gravity, float choppiness, const float* h0,
                          const float* h0c, const float* times, int nsteps, float* vertices, float* normals, float* white,
                          int white_stride) {
    OceanConsts C;
    C.N = N; C.length = length; C.gravity = gravity; C.unit_width = unit_width; C.choppiness = choppiness;
    const cf* a = reinterpret_cast<const cf*>(h0);
    const cf* b = reinterpret_cast<const cf*>(h0c);
    if (nsteps < 1 || nsteps > MW_MAX_BATCH) return 2;
    switch (N) {
        case 64: return evaluate_n<64>(pts, C, a, b, times, nsteps, vertices, normals, white, white_stride);
        case 128: return evaluate_n<128>(pts, C, a, b, times, nsteps, vertices, normals, white, white_stride);
        case 256: return evaluate_n<256>(pts, C, a, b, times, nsteps, vertices, normals, white, white_stride);
        case 512: return evaluate_n<512>(pts, C, a, b, times, nsteps, vertices, normals, white, white_stride);
        case 1024: return evaluate_n<1024>(pts, C, a, b, times, nsteps, vertices, normals, white, white_stride);
        case 2048: return evaluate_n<2048>(pts, C, a, b, times, nsteps, vertices, normals, white, white_stride);
        case 4096: return evaluate_n<4096>(pts, C, a, b, times, nsteps, vertices, normals, white, white_stride);
        default: return 1;
    }
}

// 1-D transform through the Stockham passes exactly as one FFT group of the kernels runs them
int emul_fft1d(int N, int pts, const float* in_xy, float* out_xy) {
#define RUN(NN)                                              \
    case NN:                                                 \
        if (pts == 8) return fft1d_np<NN, 8>(in_xy, out_xy); \
        return fft1d_np<NN, 16>(in_xy, out_xy);
    switch (N) {
        RUN(64) RUN(128) RUN(256) RUN(512) RUN(1024) RUN(2048) RUN(4096)
        default: return 1;
    }
#undef RUN
}

// OceanRenderer: init (transposed initial spectrum + zero phase) and one GenerateTexture step
void emul_or_init(int M, float length, float wind_x, float wind_y, float amplitude, float gravity, uint64_t seed, float* initT,
                  float* phaseT) {
    for (int px = 0; px < M; px++)
        for (int py = 0; py < M; py++)
            or_init_element(M, length, wind_x, wind_y, amplitude / 10000.f, gravity, seed, px, py, reinterpret_cast<f4*>(initT), phaseT);
}
int emul_or_step(int M, float length, float gravity, float choppiness, float dt, const float* initT, float* phaseT,
                 float* height, float* disp, float* disp_g, float* normal, float* white, float* height_g, float* disp_a, int packed) {
    OrConsts C;
    C.M = M; C.length = length; C.gravity = gravity; C.choppiness = choppiness; C.normal_length = length;
    const f4* it = reinterpret_cast<const f4*>(initT);
    cf* d = reinterpret_cast<cf*>(disp);
    switch (M) {
        case 64: return or_step_np<64, Plan<64>::P>(C, dt, it, phaseT, height, d, disp_g, normal, white, height_g, disp_a, packed != 0);
        case 128: return or_step_np<128, Plan<128>::P>(C, dt, it, phaseT, height, d, disp_g, normal, white, height_g, disp_a, packed != 0);
        case 256: return or_step_np<256, Plan<256>::P>(C, dt, it, phaseT, height, d, disp_g, normal, white, height_g, disp_a, packed != 0);
        case 512: return or_step_np<512, Plan<512>::P>(C, dt, it, phaseT, height, d, disp_g, normal, white, height_g, disp_a, packed != 0);
        case 1024: return or_step_np<1024, Plan<1024>::P>(C, dt, it, phaseT, height, d, disp_g, normal, white, height_g, disp_a, packed != 0);
        default: return 1;
    }
}

// consumer-side packing of one frame: the four RGBA render targets and the material's vertex stage on the mesh
void emul_or_pack_rgba(int M, const float* height, const float* height_g, const float* disp, const float* disp_g,
                       const float* disp_a, const float* normal, const float* white, float* H, float* D, float* Nn, float* W) {
    for (size_t idx = 0; idx < (size_t)M * M; idx++)
        or_pack_rgba_element(idx, height, height_g, reinterpret_cast<const cf*>(disp), disp_g, disp_a, normal, white,
                             reinterpret_cast<f4*>(H), reinterpret_cast<f4*>(D), reinterpret_cast<f4*>(Nn), reinterpret_cast<f4*>(W));
}
void emul_or_displace_mesh(int M, int res, float unit_width, const float* height, const float* disp, const float* normal,
                           const float* white, float* vert, float* nrm, float* col) {
    for (int i = 0; i < res; i++)
        for (int j = 0; j < res; j++)
            or_mesh_vertex(M, res, unit_width, i, j, height, reinterpret_cast<const cf*>(disp), normal, white, vert, nrm, col);
}

void emul_rest_mesh(int N, float unit_width, float* vertices, float* normals, float* uvs, int32_t* indices) {
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) rest_mesh_element(N, unit_width, i, j, vertices, normals, uvs, indices);
}

void emul_spectrum(int N, float length, float wind_x, float wind_y, float amplitude, float gravity, uint64_t seed, float* h0,
                   float* h0c) {
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            spectrum_element(N, length, wind_x, wind_y, amplitude, gravity, seed, i, j, reinterpret_cast<cf*>(h0),
                             reinterpret_cast<cf*>(h0c));
}

void emul_wave_transpose4_source(int lane, int rho, int* src_lane, int* src_rho) { wave_transpose4_source(lane, rho, src_lane, src_rho); }
void emul_omega_t(int N, float length, float gravity, float t, float* out) {
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) out[i * N + j] = omega_t_f32(N, length, gravity, i, j, t);
}

void emul_gerstner(const float* pos, long nverts, const float* waves, int nwaves, float amplitude, float frequency,
                   float steepness, float t, float* out) {
    GerstnerWaves wv;
    for (int i = 0; i < MW_GERSTNER_MAX_WAVES; i++) {
        wv.dx[i] = i < nwaves ? waves[3 * i] : 0.f;
        wv.dy[i] = i < nwaves ? waves[3 * i + 1] : 0.f;
        wv.speed[i] = i < nwaves ? waves[3 * i + 2] : 0.f;
    }
    for (long v = 0; v < nverts; v++)
        gerstner_vertex(wv, nwaves, amplitude, frequency, steepness, t, pos[3 * v], pos[3 * v + 1], pos[3 * v + 2], &out[3 * v],
                        &out[3 * v + 1], &out[3 * v + 2]);
}

void emul_set_variant(int v) { g_force_hs = (v == 1); g_frame = (v == 2); }

// the time-batched Gerstner path: position part once, angle addition per step (cb/sb as the launcher forms them)
void emul_gerstner_steps(const float* pos, long nverts, const float* waves, int nwaves, float amplitude, float frequency,
                         float steepness, const float* t, int nsteps, float* out) {
    GerstnerWaves wv;
    GerstnerPhases ph;
    for (int i = 0; i < MW_GERSTNER_MAX_WAVES; i++) {
        wv.dx[i] = i < nwaves ? waves[3 * i] : 0.f;
        wv.dy[i] = i < nwaves ? waves[3 * i + 1] : 0.f;
        wv.speed[i] = i < nwaves ? waves[3 * i + 2] : 0.f;
    }
    for (int k = 0; k < nsteps; k++)
        for (int i = 0; i < nwaves; i++) {
            const double b = (double)t[k] * (double)wv.speed[i];
            ph.cb[k * nwaves + i] = (float)cos(b);
            ph.sb[k * nwaves + i] = (float)sin(b);
        }
    for (long v = 0; v < nverts; v++) {
        if (nwaves == 4) {
            float sa[4], ca[4];
            gerstner_position_part<4>(wv, frequency, pos[3 * v], pos[3 * v + 2], sa, ca);
            for (int k = 0; k < nsteps; k++)
                gerstner_step_vertex<4>(wv, ph, k, amplitude, steepness, sa, ca, pos[3 * v], pos[3 * v + 1], pos[3 * v + 2],
                                        out + ((size_t)k * nverts + v) * 3);
        } else {
            float sa[8], ca[8];
            gerstner_position_part<8>(wv, frequency, pos[3 * v], pos[3 * v + 2], sa, ca);
            for (int k = 0; k < nsteps; k++)
                gerstner_step_vertex<8>(wv, ph, k, amplitude, steepness, sa, ca, pos[3 * v], pos[3 * v + 1], pos[3 * v + 2],
                                        out + ((size_t)k * nverts + v) * 3);
        }
    }
}

// pond Displacement(): p is an mw_pond_params
void emul_pond(const mw_pond_params* p, const float* pos, long nverts, float t, float* out, float* nrm) {
    PondParams P;
    P.mode = p->mode; P.amplitude = p->amplitude; P.frequency = p->frequency; P.speed = p->speed;
    P.steepness = p->steepness; P.smoothing = p->smoothing;
    for (int i = 0; i < 4; i++) { P.wspeed[i] = p->wspeed[i]; P.dir_ab[i] = p->dir_ab[i]; P.dir_cd[i] = p->dir_cd[i]; }
    for (long v = 0; v < nverts; v++) pond_vertex(P, t, pos[3 * v], pos[3 * v + 1], pos[3 * v + 2], &out[3 * v], &nrm[3 * v]);
}

// pass-1 block mapping (speed-only remap; must be a bijection onto (column job, step) plus padding blocks)
int emul_p1_block_map(int bid, int gx, int nsteps, int tgroup, int* jb, int* step) {
    return p1_block_map(bid, gx, nsteps, tgroup, jb, step) ? 1 : 0;
}
int emul_p1_grid_blocks(int gx, int nsteps, int tgroup) { return p1_grid_blocks(gx, nsteps, tgroup); }
// job list of the single-step plan's pass 1 (p1_frame_jobs): returns its length, copies up to cap entries
int emul_p1_frame_jobs(int N, int* out, int cap) {
    const int cw = N >= MW_CW2_MIN_N ? 2 : 4;
    const std::vector<int> j = p1_frame_jobs(N, cw);
    for (int i = 0; i < (int)j.size() && i < cap; i++) out[i] = j[i];
    return (int)j.size();
}
int emul_p1_field_active(int N, int jb, int f) { return p1_field_active(N, jb, f, N >= MW_CW2_MIN_N ? 2 : 4) ? 1 : 0; }

// sum_ij F_f(i,j) e^{i(k_i x_a + k_j z_b)} for nfields complex N x N fields through the two chirp-z launches (along j, then
// along i; each stores transposed), tables from czt_build_tables: in [f][i][j], out [f][a][b], interleaved (re, im) float32
int emul_czt2d(int N, float unit_width, float length, int nfields, const float* in_xy, float* out_xy) {
    const int M = czt_size_io(N, N);
    if (!M) return 2;
    std::vector<cf> w1, w2, Hh, tmp((size_t)nfields * N * N);
    czt_build_tables(N, N, M, unit_width, length, w1, w2, Hh);
    CztArgs A;
    A.w1 = w1.data(); A.w2 = w2.data(); A.Hh = Hh.data();
    A.nin = N; A.nout = N; A.rows = N; A.in_ld = N; A.out_ld = N; A.in_plane = (long long)N * N; A.out_plane = (long long)N * N;
    A.in = reinterpret_cast<const cf*>(in_xy); A.out = tmp.data();          // along j: tmp[f][b][i]
    int r = czt_pass(M, A, nfields);
    if (r) return r;
    A.in = tmp.data(); A.out = reinterpret_cast<cf*>(out_xy);               // along i: out[f][a][b]
    return czt_pass(M, A, nfields);
}
// the tabulated form of an element (what czt_load runs: omega(i, j) and the wave numbers from k_czt_tables' tables) against the form with
// everything computed in place (czt_packed_value): number of elements of the three planes on [0, N]^2 whose bits differ
int emul_czt_tables_vs_inline(int N, float length, float gravity, const float* h0_xy, const float* h0c_xy, float t) {
    OceanConsts C{};
    C.N = N; C.length = length; C.gravity = gravity; C.unit_width = 1.f; C.choppiness = 1.f;
    const cf* h0 = reinterpret_cast<const cf*>(h0_xy);
    const cf* h0c = reinterpret_cast<const cf*>(h0c_xy);
    std::vector<float> Om((size_t)(N + 1) * (N + 1)), K(N + 1);
    for (int e = 0; e < (N + 1) * (N + 1); e++) czt_table_element(N, length, gravity, e, Om.data(), K.data());
    int bad = 0;
    for (int plane = 0; plane < MW_CZT_PLANES; plane++)
        for (int i = 0; i <= N; i++)
            for (int j = 0; j <= N; j++) {
                const bool in0 = i < N && j < N, in1 = i > 0 && j > 0;
                const size_t i0 = in0 ? (size_t)i * N + j : 0, i1 = in1 ? (size_t)(N - i) * N + (N - j) : 0;
                const cf a = czt_packed_from(C, h0[i0], h0c[i0], in0, h0[i1], h0c[i1], in1, t, Om[(size_t)i * (N + 1) + j], K[i], K[j], plane);
                const cf b = czt_packed_value(C, h0, h0c, t, i, j, plane);
                if (std::memcmp(&a, &b, sizeof(cf)) != 0) bad++;
            }
    return bad;
}
// the product's form: the three Hermitian-packed planes formed from (h0, h0conj, t) on the index set [0, N]^2 (czt_packed_value),
// through the same two launches as czt_evaluate runs them; out [3][a][b] = (H + i Dx, Sx + i Sz, Dz + i 0)
int emul_czt_packed(int N, float unit_width, float length, float gravity, const float* h0_xy, const float* h0c_xy, float t, float* out_xy) {
    const int M = czt_size(N);
    if (!M) return 2;
    std::vector<cf> w1, w2, Hh, tmp((size_t)MW_CZT_PLANES * N * (N + 1));
    czt_build_tables(N, N + 1, M, unit_width, length, w1, w2, Hh);
    CztArgs A;
    A.w1 = w1.data(); A.w2 = w2.data(); A.Hh = Hh.data();
    A.nin = N + 1; A.nout = N;
    A.C.N = N; A.C.length = length; A.C.gravity = gravity; A.C.unit_width = unit_width; A.C.choppiness = 1.f;
    A.h0 = reinterpret_cast<const cf*>(h0_xy); A.h0c = reinterpret_cast<const cf*>(h0c_xy); A.t = t;
    std::vector<float> Om((size_t)(N + 1) * (N + 1)), K(N + 1);  // k_czt_tables
    for (int e = 0; e < (N + 1) * (N + 1); e++) czt_table_element(N, length, gravity, e, Om.data(), K.data());
    A.Om = Om.data(); A.K = K.data();
    A.in = tmp.data(); A.out = tmp.data();
    A.rows = N + 1; A.in_ld = N + 1; A.in_plane = (long long)N * (N + 1); A.out_ld = N + 1; A.out_plane = (long long)N * (N + 1);
    int r = czt_pass(M, A, MW_CZT_PLANES);
    if (r) return r;
    A.h0 = nullptr;
    A.out = reinterpret_cast<cf*>(out_xy);
    A.rows = N; A.out_ld = N; A.out_plane = (long long)N * N;
    return czt_pass(M, A, MW_CZT_PLANES);
}

}  // extern "C"
