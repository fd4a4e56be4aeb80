// Practical HBM streaming rates of one MI355X with hand-written kernels (evidence for DESIGN.md's "streaming ceiling").
// Round 3: the round-2 probe (one 16-B access in flight per lane, plain grid-stride loop) measured copy 5.35-5.5 TB/s while
// /opt/skills/guides/MI355X_MICROARCH.md records 6.29 TB/s for a float4 copy.  This version sweeps what the first one left
// out: U independent accesses in flight per lane (1, 2, 4, 8), 8- vs 16-byte lanes (the pass kernels move float2), persistent
// grids of 1..8 workgroups per CU vs one chunk per workgroup, plain / non-temporal stores and loads, LDS-DMA reads
// (global_load_lds_dwordx4), and the pass-2 mix of 3 reads : 4 writes.
//   hipcc --offload-arch=gfx950 -O3 -o variants/hbm_probe tools/hbm_probe.hip && variants/hbm_probe [MiB per buffer]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

template <class V> __device__ __forceinline__ V ld(const V* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
template <class V> __device__ __forceinline__ void st(V* p, V v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }
template <class V> __device__ __forceinline__ float sum(V v);
template <> __device__ __forceinline__ float sum(f4v v) { return v.x + v.y + v.z + v.w; }
template <> __device__ __forceinline__ float sum(f2v v) { return v.x + v.y; }

// A workgroup walks chunks of U * blockDim elements; inside a chunk lane t owns elements t, t + blockDim, ...: every
// instruction of a wave covers 64 consecutive elements and U of them are in flight before the first use.
template <class V, int U, bool NTL>
__global__ void k_read(const V* __restrict__ a, size_t n, float* out) {
    float acc = 0.f;
    const size_t chunk = (size_t)U * blockDim.x;
    for (size_t c = blockIdx.x; c * chunk < n; c += gridDim.x) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld(&a[c * chunk + (size_t)u * blockDim.x + threadIdx.x], NTL);
#pragma unroll
        for (int u = 0; u < U; u++) acc += sum(v[u]);
    }
    if (acc == 123.456f) out[0] = 1.f;
}
template <class V, int U, bool NTS>
__global__ void k_write(V* __restrict__ b, size_t n) {
    V v;
    for (int i = 0; i < (int)(sizeof(V) / 4); i++) v[i] = 1.f + i;
    const size_t chunk = (size_t)U * blockDim.x;
    for (size_t c = blockIdx.x; c * chunk < n; c += gridDim.x)
#pragma unroll
        for (int u = 0; u < U; u++) st(&b[c * chunk + (size_t)u * blockDim.x + threadIdx.x], v, NTS);
}
template <class V, int U, bool NTL, bool NTS>
__global__ void k_copy(const V* __restrict__ a, V* __restrict__ b, size_t n) {
    const size_t chunk = (size_t)U * blockDim.x;
    for (size_t c = blockIdx.x; c * chunk < n; c += gridDim.x) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld(&a[c * chunk + (size_t)u * blockDim.x + threadIdx.x], NTL);
#pragma unroll
        for (int u = 0; u < U; u++) st(&b[c * chunk + (size_t)u * blockDim.x + threadIdx.x], v[u], NTS);
    }
}
// the pass-2 ratio, 21 B read : 28 B written: 3 read streams and 4 write streams of n/4 elements each
template <class V, int U, bool NTS>
__global__ void k_mix(const V* __restrict__ a, V* __restrict__ b, size_t n) {
    const size_t q = n / 4, chunk = (size_t)U * blockDim.x;
    for (size_t c = blockIdx.x; c * chunk < q; c += gridDim.x) {
        V v[3][U];
#pragma unroll
        for (int s = 0; s < 3; s++)
#pragma unroll
            for (int u = 0; u < U; u++) v[s][u] = a[s * q + c * chunk + (size_t)u * blockDim.x + threadIdx.x];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = c * chunk + (size_t)u * blockDim.x + threadIdx.x;
            st(&b[i], v[0][u], NTS);
            st(&b[q + i], v[1][u], NTS);
            st(&b[2 * q + i], v[2][u], NTS);
            st(&b[3 * q + i], v[0][u] + v[1][u], NTS);
        }
    }
}
// LDS-DMA read: every wave lands U KiB per round in its own LDS slice and never looks at it (aux = 0 default policy, 2 = nt)
template <int U, int AUX>
__global__ void k_read_ldsdma(const f4v* __restrict__ a, size_t n, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x / 64;
    __attribute__((address_space(3))) unsigned char* lds =
        (__attribute__((address_space(3))) unsigned char*)smem + (size_t)wave * U * 1024;
    const size_t chunk = (size_t)U * blockDim.x;
    for (size_t c = blockIdx.x; c * chunk < n; c += gridDim.x) {
#pragma unroll
        for (int u = 0; u < U; u++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)&a[c * chunk + (size_t)u * blockDim.x + threadIdx.x],
                                             (__attribute__((address_space(3))) void*)(lds + u * 1024), 16, 0, AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (n == 12345) out[0] = smem[threadIdx.x];
}

static double time_ms(const std::function<void()>& f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}

struct Variant {
    std::string name;
    double bytes;                                     // bytes moved per launch
    std::function<void(int grid, int block)> launch;  // grid = workgroups
    int u, elem;                                      // elements per lane per chunk, element bytes (for the one-chunk-per-WG grid)
};

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 2048;
    const size_t bytes = mib << 20;
    f4v *a, *b; float* out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    const size_t n16 = bytes / 16, n8 = bytes / 8;
    std::vector<Variant> vs;
#define ADD(NAME, BYTES, U_, ELEM, ...) vs.push_back({NAME, (double)(BYTES), [=](int g, int t) { __VA_ARGS__; }, U_, ELEM})
#define READS(U) \
    ADD("read  16B U" #U, bytes, U, 16, (k_read<f4v, U, false><<<g, t>>>(a, n16, out))); \
    ADD("read  16B U" #U " nt", bytes, U, 16, (k_read<f4v, U, true><<<g, t>>>(a, n16, out))); \
    ADD("read   8B U" #U, bytes, U, 8, (k_read<f2v, U, false><<<g, t>>>((const f2v*)a, n8, out)))
    READS(1); READS(2); READS(4); READS(8);
#define LDSDMA(U) \
    ADD("read  LDS-DMA 16B U" #U, bytes, U, 16, (k_read_ldsdma<U, 0><<<g, t, (t / 64) * U * 1024>>>(a, n16, out))); \
    ADD("read  LDS-DMA 16B U" #U " nt", bytes, U, 16, (k_read_ldsdma<U, 2><<<g, t, (t / 64) * U * 1024>>>(a, n16, out)))
    LDSDMA(2); LDSDMA(4); LDSDMA(8);
#define WRITES(U) \
    ADD("write 16B U" #U, bytes, U, 16, (k_write<f4v, U, false><<<g, t>>>(b, n16))); \
    ADD("write 16B U" #U " nt", bytes, U, 16, (k_write<f4v, U, true><<<g, t>>>(b, n16))); \
    ADD("write  8B U" #U " nt", bytes, U, 8, (k_write<f2v, U, true><<<g, t>>>((f2v*)b, n8)))
    WRITES(1); WRITES(4); WRITES(8);
#define COPIES(U) \
    ADD("copy  16B U" #U, 2.0 * bytes, U, 16, (k_copy<f4v, U, false, false><<<g, t>>>(a, b, n16))); \
    ADD("copy  16B U" #U " nt-store", 2.0 * bytes, U, 16, (k_copy<f4v, U, false, true><<<g, t>>>(a, b, n16))); \
    ADD("copy  16B U" #U " nt-both", 2.0 * bytes, U, 16, (k_copy<f4v, U, true, true><<<g, t>>>(a, b, n16))); \
    ADD("copy   8B U" #U " nt-store", 2.0 * bytes, U, 8, (k_copy<f2v, U, false, true><<<g, t>>>((const f2v*)a, (f2v*)b, n8)))
    COPIES(1); COPIES(2); COPIES(4); COPIES(8);
#define MIXES(U) \
    ADD("mix 3r:4w 16B U" #U " nt-store", 1.75 * bytes, U, 16, (k_mix<f4v, U, true><<<g, t>>>(a, b, n16))); \
    ADD("mix 3r:4w  8B U" #U " nt-store", 1.75 * bytes, U, 8, (k_mix<f2v, U, true><<<g, t>>>((const f2v*)a, (f2v*)b, n8))); \
    ADD("mix 3r:4w  8B U" #U, 1.75 * bytes, U, 8, (k_mix<f2v, U, false><<<g, t>>>((const f2v*)a, (f2v*)b, n8)))
    MIXES(1); MIXES(2); MIXES(4);
    const int per_cu[] = {1, 2, 4, 8, 0};  // workgroups per CU of the persistent grids; 0 = one chunk per workgroup
    const int blocks[] = {256, 512, 1024};
    printf("# %zu MiB per buffer; TB/s = bytes moved / launch time; best launch shape per variant (wg/CU 0 = one chunk per workgroup)\n", mib);
    printf("%-34s %8s %8s %6s | TB/s by (wg/CU x threads): ", "variant", "best", "wg/CU", "thr");
    for (int pc : per_cu) for (int t : blocks) printf("%dx%d ", pc, t);
    printf("\n");
    for (auto& v : vs) {
        double best = 0; int bpc = 0, bt = 0;
        std::string row;
        for (int pc : per_cu) for (int t : blocks) {
            if (pc * t > 2048) { row += "   -  "; continue; }  // beyond 32 waves per CU
            const double elems = v.bytes / (v.name[0] == 'c' ? 2.0 : (v.name[0] == 'm' ? 7.0 : 1.0)) / v.elem;
            const long long chunks = (long long)(elems / ((double)v.u * t));
            const int g = pc ? pc * 256 : (int)(chunks > 0x7fffffffLL ? 0x7fffffff : chunks);
            const double ms = time_ms([&] { v.launch(g, t); }, 8);
            const double tb = v.bytes / (ms * 1e-3) / 1e12;
            char buf[32]; snprintf(buf, sizeof buf, "%5.2f ", tb); row += buf;
            if (tb > best) { best = tb; bpc = pc; bt = t; }
        }
        if (hipGetLastError() != hipSuccess) { printf("%-34s launch error\n", v.name.c_str()); continue; }
        printf("%-34s %8.2f %8d %6d | %s\n", v.name.c_str(), best, bpc, bt, row.c_str());
        fflush(stdout);
    }
    return 0;
}
