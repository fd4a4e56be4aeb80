// valu_probe.hip -- issue rate of plain and packed float32 VALU instructions on gfx950 (MI355X), per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/valu_probe.hip && /tmp/valu_probe
// Question (round 4): is a wave64 v_add_f32 2 or 4 cycles of a SIMD's issue, and does v_pk_add_f32 / v_pk_fma_f32 (two floats per lane)
// cost the same slot -- i.e. does complex arithmetic written on float2 vectors halve the VALU time of the transform kernels?
// Each wave runs ITERS rounds of 16 independent chains; W waves per SIMD (1, 2, 4).  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHAINS 16
template <int MODE>
__global__ void k_probe(float* out, int iters, float seed) {
    float a[CHAINS];
    f2 p[CHAINS];
    const float s = seed + threadIdx.x * 1e-6f;
    const f2 s2 = {s, s * 0.5f};
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { a[c] = s * (c + 1); p[c] = f2{s * (c + 1), s * (c + 2)}; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (MODE == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[c]) : "v"(s));
            if (MODE == 1) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[c]) : "v"(s2));
            if (MODE == 2) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(a[c]) : "v"(s));
            if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(p[c]) : "v"(s2));
            if (MODE == 4) asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "+v"(p[c]) : "v"(s2));
            if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(p[c]) : "v"(s2));
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc += a[c] + p[c].x + p[c].y;
    if (acc == 123.456f) out[blockIdx.x] = acc;
}
template <int MODE>
static void run(const char* name, int waves_per_simd, float* d, double ghz) {
    const int iters = 20000, blocks = 256 * 4;  // 4 blocks per CU (one per SIMD if the dispatcher spreads them), 64*W threads each
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_probe<MODE><<<blocks, 64 * waves_per_simd>>>(d, 2000, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_probe<MODE><<<blocks, 64 * waves_per_simd>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves_per_simd waves x iters x CHAINS instructions (if each block's waves spread over the 4 SIMDs: blocks*W/1024 waves per SIMD)
    const double instr_per_simd = (double)blocks * waves_per_simd / 1024.0 * iters * CHAINS;
    printf("%-34s waves/SIMD %d  %8.3f ms  %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz);
}
int main() {
    float* d;
    hipMalloc(&d, 4096 * sizeof(float));
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f32", w, d, ghz);
        run<1>("v_pk_add_f32", w, d, ghz);
        run<2>("v_fma_f32", w, d, ghz);
        run<3>("v_pk_fma_f32", w, d, ghz);
        run<4>("v_pk_add_f32 op_sel+neg", w, d, ghz);
        run<5>("v_pk_mul_f32 op_sel_hi (broadcast)", w, d, ghz);
    }
    return 0;
}
