// gerstner_kernels.h -- pond path: Gerstner vertex displacement (BASELINE config 5).
// Replaces W/MistralWaterLib.cginc:71-99 Gerstner() as called from :154-180 Displacement().
// Streaming kernel, 24 B/vertex algorithmic (read position 12 B + write displaced position 12 B).
#pragma once
#include "fftmesh_kernels.h"
#include "mw_switches.h"

#define MW_GERSTNER_MAX_WAVES 16

namespace mw {

struct GerstnerWaves {
    float dx[MW_GERSTNER_MAX_WAVES], dy[MW_GERSTNER_MAX_WAVES], speed[MW_GERSTNER_MAX_WAVES];
};

// one vertex: offsets of W/MistralWaterLib.cginc:77-88, added to the position (:176)
MW_HD void gerstner_vertex(const GerstnerWaves& wv, int nwaves, float amplitude, float frequency, float steepness, float t,
                           float px, float py, float pz, float* ox, float* oy, float* oz) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    const float sa = steepness * amplitude;  // :77-78
    for (int i = 0; i < nwaves; i++) {
        const float th = frequency * (wv.dx[i] * px + wv.dy[i] * pz) + t * wv.speed[i];  // :80-84 (sVertex.xz = world x,z)
        float s, c;
        mw_sincos_fast(th, &s, &c);
        sx += c * (sa * wv.dx[i]);  // :86
        sz += c * (sa * wv.dy[i]);  // :87
        sy += s;                    // :88
    }
    *ox = px + sx;
    *oy = py + amplitude * sy;
    *oz = pz + sz;
}

#if defined(__HIPCC__)
// 16-B alignment of every address a kernel forms by reinterpreting a float* as f4* (f4 is alignas(16)): the base pointers
// and, for step-strided outputs, the step stride.  A torch view or a caller's sub-array need not have it: the launchers
// then pass nvec = 0 and the scalar loop serves every vertex (correct at any alignment, ~2x slower).
static inline bool mw_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// 4 vertices (= 3 x float4) per thread: every load/store is a 16-B access, lanes contiguous.  Vertices [0, nvec) take that
// path (nvec a multiple of 4; 0 when the buffers are not 16-B aligned), [nvec, nverts) a scalar grid-stride loop.
// A lane of the vector paths owns 4 vertices = 3 float4 = 48 contiguous bytes.  Stored straight from the lane, each of the three
// store instructions writes 16-byte pieces 48 bytes apart: a third of every line per instruction, left to the L2 to merge (and
// with a non-temporal hint 2.5x slower).  wave_store_3f4 turns a wave's 64 x 48 B through LDS (wave-local: LDS operations of one
// wave execute in order, no workgroup barrier; ds_write_b128 at 48-byte lane stride is conflict-free) so that every store
// instruction writes 64 consecutive float4 = 1 KiB.  wt: the wave's 192-float4 tile; dst: the chunk's first float4; nf4: its
// valid float4 (192 except in the last wave).
template <bool NT>
__device__ __forceinline__ void wave_store_3f4(f4* wt, int lane, f4 r0, f4 r1, f4 r2, f4* dst, int nf4) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // earlier reads of the tile are done
    __builtin_amdgcn_wave_barrier();
    wt[lane * 3] = r0; wt[lane * 3 + 1] = r1; wt[lane * 3 + 2] = r2;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (j * 64 + lane < nf4) mw_store_stream<NT>(&dst[j * 64 + lane], wt[j * 64 + lane]);
}

__global__ __launch_bounds__(256) void k_gerstner(const float* __restrict__ pos, float* __restrict__ out, int64_t nverts, int64_t nvec,
                                                  GerstnerWaves wv, int nwaves, float amplitude, float frequency,
                                                  float steepness, float t) {
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
#pragma unroll
        for (int k = 0; k < 4; k++)
            gerstner_vertex(wv, nwaves, amplitude, frequency, steepness, t, v[3 * k], v[3 * k + 1], v[3 * k + 2], &v[3 * k],
                            &v[3 * k + 1], &v[3 * k + 2]);
        f4 r0 = {v[0], v[1], v[2], v[3]}, r1 = {v[4], v[5], v[6], v[7]}, r2 = {v[8], v[9], v[10], v[11]};
        wave_store_3f4<false>(wt, lane, r0, r1, r2, reinterpret_cast<f4*>(out) + q0 * 3, (int)(nquads - q0 < 64 ? nquads - q0 : 64) * 3);
    }
    // the rest (nverts % 4, or everything when the buffers are not 16-B aligned): one vertex per thread and trip
    for (int64_t vtx = nvec + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; vtx < nverts; vtx += stride) {
        float ox, oy, oz;
        gerstner_vertex(wv, nwaves, amplitude, frequency, steepness, t, pos[3 * vtx], pos[3 * vtx + 1], pos[3 * vtx + 2], &ox,
                        &oy, &oz);
        out[3 * vtx] = ox; out[3 * vtx + 1] = oy; out[3 * vtx + 2] = oz;
    }
}

#endif

// ---- many time-steps of one lattice in one launch ------------------------------------------------------------
// theta_i(x, t) = frequency * dot(dir_i, x.xz) + t * speed_i splits into a position part (per vertex and wave, one
// hardware sincos, kept in registers) and a time part (per wave and step, uniform: cos/sin(t_k * speed_i) come
// from the host in the kernel arguments), joined by the angle-addition formulas: 7 FMAs per vertex, wave and step instead
// of a sincos.  The positions are read once per launch; per step only the 12-B result leaves.
#define MW_GERSTNER_PHASES 256  // nsteps * nwaves per launch (2 KiB of kernel arguments)
#ifndef MW_POND_STEPS_PER_WG
#define MW_POND_STEPS_PER_WG 8  // time values per workgroup of k_gerstner_steps (switch MW_POND_STEPS_PER_WG overrides: A/B)
#endif
struct GerstnerPhases {
    float cb[MW_GERSTNER_PHASES], sb[MW_GERSTNER_PHASES];  // [step * nwaves + i]
};
template <int NW>
MW_HD void gerstner_position_part(const GerstnerWaves& wv, float frequency, float px, float pz, float (&sa)[NW], float (&ca)[NW]) {
#pragma unroll
    for (int i = 0; i < NW; i++) mw_sincos_fast(frequency * (wv.dx[i] * px + wv.dy[i] * pz), &sa[i], &ca[i]);
}
// V = float, or a 2-float vector (two vertices of a lane per v_pk_* instruction: the round-5 experiment, no faster -- see the note below)
template <int NW, class V>
MW_HD void gerstner_step_vertex(const GerstnerWaves& wv, const GerstnerPhases& ph, int step, float amplitude, float steepness,
                                const V (&sa)[NW], const V (&ca)[NW], V px, V py, V pz, V* o) {
    V sx = px * 0.f, sy = sx, sz = sx;
    const float sam = steepness * amplitude;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        const float cb = ph.cb[step * NW + i], sb = ph.sb[step * NW + i];
        const V c = ca[i] * cb - sa[i] * sb, s = sa[i] * cb + ca[i] * sb;  // cos/sin(theta_i)
        sx += c * (sam * wv.dx[i]);
        sz += c * (sam * wv.dy[i]);
        sy += s;
    }
    o[0] = px + sx; o[1] = py + sy * amplitude; o[2] = pz + sz;
}

#if defined(__HIPCC__)
// A wave's 256 vertices as 4 x 64: lane l owns vertices l, l + 64, l + 128, l + 192 of the chunk, so that every load and store
// instruction is 64 lanes x 12 B = 768 contiguous bytes (global_load / global_store_dwordx3, non-temporal) -- no LDS turn, no fences,
// no alignment requirement, no scalar tail (round 4; the float4 + LDS-turn form of round 2 measured 0.62-0.63 of the HBM peak, this
// one 0.67-0.70).  blockIdx.y = a group of `steps_per_wg` consecutive time values: the step axis is cut into workgroups that come
// and go, issued in address order, instead of one long-lived workgroup walking all 32 slabs -- on this memory system fresh
// workgroups beat long-lived ones for every store stream (profiles/r03_hbm_probe.txt); 8 steps per workgroup measured best (the
// position part, 8 sincos per vertex, is then re-formed 4 times per launch: +13 % VALU, still well under the store time).
#ifndef MW_POND_LOADS_FIRST
#define MW_POND_LOADS_FIRST 1
#endif
template <int NW>
__global__ __launch_bounds__(256) void k_gerstner_steps(const float* __restrict__ pos, float* __restrict__ out, int64_t nverts,
                                                        GerstnerWaves wv, GerstnerPhases ph, int nsteps, float amplitude,
                                                        float frequency, float steepness, int steps_per_wg, int xcd_blocks) {
    // xcd_blocks > 0 (round 5): 1-D grid; the step groups of ONE vertex chunk sit in consecutive slots of ONE XCD (the dispatcher
    // places workgroup b on XCD b % 8), so the first of them pulls the chunk's 12 KiB of positions into that XCD's L2 and the others
    // hit there: the positions cross HBM once per launch instead of once per step group (13.5 -> 12.4 B per vertex-step) while the
    // workgroups stay short-lived.  xcd_blocks = the number of vertex chunks per trip (what gridDim.x was in the 2-D form).
    int vb = (int)blockIdx.x, sg = (int)blockIdx.y, nvb = (int)gridDim.x;
    if (xcd_blocks > 0) {
        const int ngroups = (nsteps + steps_per_wg - 1) / steps_per_wg;
        const int xcd = (int)blockIdx.x % 8, slot = (int)blockIdx.x / 8;
        sg = slot % ngroups;
        vb = (slot / ngroups) * 8 + xcd;
        nvb = xcd_blocks;
        if (vb >= nvb) return;
    }
    const int step_lo = sg * steps_per_wg, step_hi = step_lo + steps_per_wg < nsteps ? step_lo + steps_per_wg : nsteps;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t v0 = ((int64_t)vb * 4 + wave) * 256; v0 < nverts; v0 += (int64_t)nvb * 1024) {  // wave-uniform chunk of 256 vertices
        float v[12];
        float sa[4][NW], ca[4][NW];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {  // the four position loads first (round 5: written load, position part, load, ... the compiler kept each
            const int64_t vid = v0 + k * 64 + lane;  // load behind the previous vertex's eight sines: four memory latencies at the head of a workgroup)
            ok[k] = vid < nverts;
            const float* p = pos + 3 * (ok[k] ? vid : v0);
            v[3 * k] = p[0]; v[3 * k + 1] = p[1]; v[3 * k + 2] = p[2];
            if (!MW_POND_LOADS_FIRST) gerstner_position_part<NW>(wv, frequency, v[3 * k], v[3 * k + 2], sa[k], ca[k]);
        }
        if (MW_POND_LOADS_FIRST) {
            mw_sched_fence();
#pragma unroll
            for (int k = 0; k < 4; k++) gerstner_position_part<NW>(wv, frequency, v[3 * k], v[3 * k + 2], sa[k], ca[k]);
        }
        for (int step = step_lo; step < step_hi; step++) {
            float* dst = out + (size_t)step * nverts * 3 + 3 * v0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float o[3];
                gerstner_step_vertex<NW, float>(wv, ph, step, amplitude, steepness, sa[k], ca[k], v[3 * k], v[3 * k + 1], v[3 * k + 2], o);
                if (ok[k]) {
                    float* q = dst + 3 * (k * 64 + lane);
                    mw_store_stream<true>(&q[0], o[0]); mw_store_stream<true>(&q[1], o[1]); mw_store_stream<true>(&q[2], o[2]);
                }
            }
        }
    }
}
#endif

// Round 5 measured what bounds this kernel (profiles/r05_ab_notes.md).  The VALU is busy 84 % of a launch (SQ_ACTIVE_INST_VALU), and yet
// neither halving the step loop's instructions (two vertices per v_pk_* instruction: MW_POND_PACKED) nor taking the step loop off the
// VALU altogether -- the offsets are bilinear in (cos a_i, sin a_i) and per-step coefficients, i.e. a K = 2 NW matrix product, built on
// v_mfma_f32_32x32x2_f32 with 8 time values = the 32 rows of a tile, parity green, commit 08cea07 -- changed the launch time (82 against
// 77 us): the launch waits for its 384 MB of result stores (4.9-5.5 TB/s of pure writes into 32 slabs; the box's best single write stream is
// 6.8), the arithmetic hides under them.  Neither is in the tree.
#if defined(__HIPCC__)
// nsteps time values in one launch; nwaves must be 4 or 8 and nsteps * nwaves <= MW_GERSTNER_PHASES (the caller checks)
static inline hipError_t gerstner_launch_steps(const float* d_pos, int64_t nverts, const float* waves, int nwaves, float amplitude,
                                               float frequency, float steepness, const float* t, int nsteps, float* d_out,
                                               hipStream_t st) {
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
    int64_t blocks = (nverts + 1023) / 1024;  // 1024 vertices per 256-thread workgroup and trip
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const int spw_env = sw(SW_POND_STEPS_PER_WG);
    const int spw = spw_env > 0 ? (spw_env < nsteps ? spw_env : nsteps) : (MW_POND_STEPS_PER_WG < nsteps ? MW_POND_STEPS_PER_WG : nsteps);
    // switch MW_POND_XCD = 1: the step groups of a vertex chunk on one XCD.  Measured round 5 (profiles/r05_ab_notes.md): 4.25e11 against 4.48e11
    // vertices/s for the 2-D grid -- the positions' second to fourth read is 1.1 of 13.5 B per vertex-step and not what the launch waits for: off
    const int xcd_env = sw(SW_POND_XCD);
    const int ngroups = (nsteps + spw - 1) / spw;
    const bool xcd = xcd_env != 0 && ngroups > 1;
    const dim3 grid = xcd ? dim3((unsigned)(((blocks + 7) / 8) * 8 * ngroups)) : dim3((unsigned)blocks, (unsigned)ngroups);
    const int xb = xcd ? (int)blocks : 0;
    if (nwaves == 4) k_gerstner_steps<4><<<grid, dim3(256), 0, st>>>(d_pos, d_out, nverts, wv, ph, nsteps, amplitude, frequency, steepness, spw, xb);
    else k_gerstner_steps<8><<<grid, dim3(256), 0, st>>>(d_pos, d_out, nverts, wv, ph, nsteps, amplitude, frequency, steepness, spw, xb);
    return hipGetLastError();
}
#endif

#if defined(__HIPCC__)
static inline hipError_t gerstner_launch(const float* d_pos, int64_t nverts, const float* waves, int nwaves, float amplitude,
                                         float frequency, float steepness, float t, float* d_out, hipStream_t st) {
    GerstnerWaves wv;
    for (int i = 0; i < MW_GERSTNER_MAX_WAVES; i++) {
        wv.dx[i] = i < nwaves ? waves[3 * i] : 0.f;
        wv.dy[i] = i < nwaves ? waves[3 * i + 1] : 0.f;
        wv.speed[i] = i < nwaves ? waves[3 * i + 2] : 0.f;
    }
    const bool vec = mw_aligned16(d_pos) && mw_aligned16(d_out);
    const int64_t nvec = vec ? (nverts & ~(int64_t)3) : 0;
    int64_t blocks = ((vec ? (nverts >> 2) : nverts) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 blocks per CU
    hipLaunchKernelGGL(k_gerstner, dim3((unsigned)blocks), dim3(256), 0, st, d_pos, d_out, nverts, nvec, wv, nwaves, amplitude,
                       frequency, steepness, t);
    return hipGetLastError();
}
#endif

}  // namespace mw
