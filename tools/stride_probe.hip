// tools/stride_probe.hip -- does a lone pass-2 workgroup camp on one memory channel?
// k_pass2_frame<1024> reads, per workgroup and field, 256 lines of 128 B that lie 32 KiB apart (exchange buffer [jb][i][4]: the four rows of a
// row block at one jb are one line; the next jb is N * 4 * 8 B further).  This probe issues exactly that pattern -- 256 workgroups, one per CU,
// workgroup ab at byte offset 128 ab, 3 fields x 256 lines, 8 B per lane -- with the line stride as a parameter: 32 KiB, and 32 KiB + a pad.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stride_probe tools/stride_probe.hip && /tmp/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void k_read(const float2* __restrict__ e, size_t line_stride_f2, size_t plane_f2, int nfields, float2* sink) {
    const int ab = blockIdx.x, tid = threadIdx.x;
    // lane -> (jb, 8-B piece of the 128-B line): 16 lanes per line, 16 lines per 256-thread instruction
    float2 acc = make_float2(0.f, 0.f);
    for (int f = 0; f < nfields; f++) {
        const float2* base = e + (size_t)f * plane_f2 + (size_t)ab * 16;
        float2 v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = base[(size_t)(q * 16 + tid / 16) * line_stride_f2 + (tid % 16)];
#pragma unroll
        for (int q = 0; q < 16; q++) { acc.x += v[q].x; acc.y += v[q].y; }
    }
    if (acc.x == 12345.f) sink[0] = acc;
}

int main() {
    const int N = 1024, WG = 256, nfields = 3;
    float2* sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pad_bytes : {0, 128, 256, 512, 1024, 2048, 4096, 8192}) {
        const size_t line_stride = (size_t)N * 4 + pad_bytes / 8;           // float2 units
        const size_t plane = line_stride * 256 + 4096;
        float2* e; hipMalloc(&e, plane * nfields * sizeof(float2)); hipMemset(e, 0, plane * nfields * sizeof(float2));
        std::vector<float> ms;
        for (int it = 0; it < 60; it++) {
            hipEventRecord(e0); k_read<<<WG, 256>>>(e, line_stride, plane, nfields, sink); hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); if (it >= 10) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const double bytes = (double)WG * nfields * 256 * 128;
        printf("line stride 32 KiB + %5d B: median %.2f us  min %.2f us   (%.1f MB -> %.2f TB/s at the median)\n", pad_bytes, ms[ms.size() / 2] * 1e3, ms[0] * 1e3,
               bytes / 1e6, bytes / (ms[ms.size() / 2] * 1e-3) / 1e12);
        hipFree(e);
    }
    return 0;
}
