// Practical HBM streaming rates of one MI355X with hand-written kernels (evidence for DESIGN.md's "streaming ceiling"):
// read-only, write-only (plain / non-temporal), copy, and the pass-2 mix (21 B read : 28 B written), 16 B per lane,
// grid-stride over 4 GiB, best of several grid sizes.   hipcc --offload-arch=gfx950 -O3 -o variants/hbm_probe tools/hbm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void k_read(const f4v* __restrict__ a, size_t n, float* out) {
    f4v acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}
template <bool NT>
__global__ void k_write(f4v* __restrict__ b, size_t n) {
    const f4v v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, &b[i]); else b[i] = v;
    }
}
template <bool NT>
__global__ void k_copy(const f4v* __restrict__ a, f4v* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const f4v v = a[i];
        if (NT) __builtin_nontemporal_store(v, &b[i]); else b[i] = v;
    }
}
// 3 reads : 4 writes of 16 B (the pass-2 ratio 21 : 28)
__global__ void k_mix(const f4v* __restrict__ a, f4v* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * blockDim.x) {
        const f4v v0 = a[i], v1 = a[i + n / 4], v2 = a[i + n / 2];
        __builtin_nontemporal_store(v0, &b[i]);
        __builtin_nontemporal_store(v1, &b[i + n / 4]);
        __builtin_nontemporal_store(v2, &b[i + n / 2]);
        __builtin_nontemporal_store(v0 + v1, &b[i + 3 * (n / 4)]);
    }
}
template <class F>
static double time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) f();  // steady clocks
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    f4v *a, *b; float* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64};
    const int blocks[] = {256, 512, 1024};
    struct R { const char* name; double best; int g, t; double bytes; } r[6] = {
        {"read", 0, 0, 0, (double)bytes}, {"write", 0, 0, 0, (double)bytes}, {"write nt", 0, 0, 0, (double)bytes},
        {"copy", 0, 0, 0, 2.0 * bytes}, {"copy nt", 0, 0, 0, 2.0 * bytes}, {"mix 3r:4w nt", 0, 0, 0, 1.75 * bytes}};
    for (int g : grids) for (int t : blocks) {
        double ms[6];
        ms[0] = time_ms([&] { k_read<<<g, t>>>(a, n, out); }, 20);
        ms[1] = time_ms([&] { k_write<false><<<g, t>>>(b, n); }, 20);
        ms[2] = time_ms([&] { k_write<true><<<g, t>>>(b, n); }, 20);
        ms[3] = time_ms([&] { k_copy<false><<<g, t>>>(a, b, n); }, 20);
        ms[4] = time_ms([&] { k_copy<true><<<g, t>>>(a, b, n); }, 20);
        ms[5] = time_ms([&] { k_mix<<<g, t>>>(a, b, n); }, 20);
        for (int k = 0; k < 6; k++) {
            const double tb = r[k].bytes / (ms[k] * 1e-3) / 1e12;
            if (tb > r[k].best) { r[k].best = tb; r[k].g = g; r[k].t = t; }
        }
    }
    for (int k = 0; k < 6; k++) printf("%-14s %.2f TB/s  (grid %d x %d threads, %zu MiB per buffer)\n", r[k].name, r[k].best, r[k].g, r[k].t, bytes >> 20);
    return 0;
}
