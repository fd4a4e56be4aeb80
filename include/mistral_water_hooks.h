/*
 * mistral_water_hooks.h -- measurement and test hooks of libmistral_water.so.
 *
 * NOT part of the drop-in boundary (include/mistral_water.h): a host integration never calls these and the C# import table
 * (bindings/csharp/MistralWaterNative.cs) does not carry them.  bench.py uses the measurement hook, tests/ use the rest
 * to compare intermediate quantities of the kernels (omega*t, hds, the device sincos) with the oracle bit for bit.
 */
#ifndef MISTRAL_WATER_HOOKS_H
#define MISTRAL_WATER_HOOKS_H

#include "mistral_water.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement hook (bench.py): times each kernel of one FFTMesh step with hipEvents on the
 * handle's stream.  ms_out[k] = mean duration of kernel k over iters launches of `nsteps` batched
 * time-steps; names_out[k] = static kernel name.  Returns the kernel count through *nkernels.      */
mw_status mw_ocean_profile_kernels(mw_ocean* o, int32_t nsteps, int32_t iters, float* ms_out, const char** names_out,
                                   int32_t* nkernels);

/* the same measurement as a distribution: stats_out[6 k + 0..5] = mean, median, p10, p90, min, max of kernel k's `iters` launch
 * durations in milliseconds (bench.py prints the median beside the mean so that roofline.frac can be recomputed from a
 * rocprofv3 kernel trace of the same command, profiles/<round>_*_kernel_pcts.json)                                          */
mw_status mw_ocean_profile_kernels_stats(mw_ocean* o, int32_t nsteps, int32_t iters, float* stats_out, const char** names_out,
                                         int32_t* nkernels);

/* time-steps of one pass-1 column job kept on one XCD when an enqueue carries nsteps time-steps (0 = plain grid) */
int32_t mw_debug_pass1_time_group(mw_ocean* o, int32_t nsteps);

/* ---- test hooks (used by tests/ only; not needed by a host integration) ---------------------------------
 * mw_debug_omega_t: omega(i,j)*t [N*N, idx = i*N + j] exactly as the kernels form it -- the quantised dispersion
 * (S/FFTMesh.cs:146,183) is compared bit for bit with the oracle.  mw_debug_get_omega: the stored table in its
 * transposed [j][i] layout.  mw_debug_sincos: the library's range-reduced sin/cos on n host floats.            */
mw_status mw_debug_omega_t(mw_ocean* o, float t, float* out_host);
/* one EvaluateWaves(t) that also returns hds [N*N*2] = (d.x, d.z) of S/FFTMesh.cs:247 as the kernels hold it: the whitecap
 * stage (forward differences, edge rules :258-274, halo rows between workgroups) is then checked bit for bit           */
mw_status mw_debug_evaluate_hds(mw_ocean* o, float t, float* vertices_xyz, float* normals_xyz, float* colors_rgba, float* hds_xy);
mw_status mw_debug_get_omega(mw_ocean* o, float* out_host);
mw_status mw_debug_sincos(const float* x_host, int32_t n, float* s_host, float* c_host);
/* the pond kernels' hardware-sine variant (v_sin_f32 / v_cos_f32 after an exact revolution count) */
mw_status mw_debug_sincos_fast(const float* x_host, int32_t n, float* s_host, float* c_host);
/* one wave applies the in-wave exchange of the 1024-point transform (v_permlane16_swap / v_permlane32_swap: the wave's four 16-lane rows
 * <-> the low two bits of the slot index) to inout_host [64 lanes][16 slots][2]: the instructions are checked against the index map the
 * host emulation uses (index work: bit for bit)                                                                                      */
mw_status mw_debug_wave_transpose4(float* inout_host);
/* The library's run-time plan switches (csrc/mw_switches.h: MW_LATENCY_PLAN, MW_FRAME_KERNEL, MW_P1_FRAME_XCD, MW_P1_TGROUP, MW_CZT_ONE,
 * MW_CZT_FUSED, MW_DIRECT_CZT, MW_TILES_FORCE_RCCL, MW_POND_STEPS_PER_WG, MW_POND_XCD, MW_OR_PACKED).  Every alternative plan gives the same bits as
 * the default (MW_OR_PACKED = 0, the OceanRenderer's three-transform plan: the same textures to float32 rounding); the tests select them here.  Process-wide; MW_DIRECT_CZT and MW_P1_TGROUP are read when a handle is created, the rest per
 * call.  A product build never reads the environment; MW_EINVAL for an unknown name (get: INT32_MIN).                                 */
mw_status mw_debug_set_switch(const char* name, int32_t value);
int32_t mw_debug_get_switch(const char* name);
/* streams `bytes` of device memory through `width`-byte per-lane loads (4, 8 or 16): FETCH_SIZE calibration */
mw_status mw_debug_stream_read(int64_t bytes, int32_t width, int32_t iters);

#ifdef __cplusplus
}
#endif
#endif /* MISTRAL_WATER_HOOKS_H */
