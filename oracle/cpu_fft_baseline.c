/*
 * oracle/cpu_fft_baseline.c  --  TEST / MEASUREMENT INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * The CPU baseline "CPU_FFT" of SURVEY.md section 8d: the FFTMesh model (S/FFTMesh.cs:141-280) evaluated through a
 * scalar radix-2 Stockham 2-D transform in float32 on the host cores, rows and columns spread over pthreads.  It is
 * what a competent CPU port of the reference would do instead of its O(N^4) loop, timed by bench.py beside the GPU
 * path (1 thread and all cores).  Valid for commensurate power-of-two grids (unit_width == length / N), where
 *     sum_ij F(i,j) e^{i(k_i x_a + k_j z_b)} = -(-1)^(a+b) * sum_ij F(i,j) pre(i+j) e^{2 pi i (i a + j b)/N},
 *     pre(m) = (-1)^m e^{i pi m / N}                                                    (DESIGN.md section 3).
 * Five separate complex transforms (H, Dx, Dz, Sx, Sz), no Hermitian packing: the straightforward port.
 * Checked against the f64 oracle in tests/test_oracle.py::test_cpu_fft_baseline_matches_oracle.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int32_t N;
    float unit_width, length, wind_x, wind_y, amplitude, choppiness, gravity;
} orc_params;
float orc_dispersion(const orc_params* p, int n, int m); /* fftmesh_oracle.c: S/FFTMesh.cs:141-147, strict f32 */

typedef struct { float re, im; } cpx;
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct job {
    void (*fn)(struct job*, int lo, int hi);
    int lo, hi;
    /* shared state */
    const orc_params* p;
    const float *h0, *h0c;
    float t;
    int N;
    cpx *A, *B;          /* [5][N*N] ping / pong */
    const cpx* tw;       /* e^{+2 pi i k / N}, k < N/2 */
    float* hds;          /* [N*N][2] */
    float *vertices, *normals, *colors;
} job;

/* Persistent worker pool: thread creation would cost more than a whole step on a many-core host (5 stages x 256
 * threads).  Workers sleep on a start barrier; the caller is participant 0. */
static struct {
    int n;                       /* threads incl. the caller */
    pthread_t* th;
    pthread_barrier_t start, done;
    job proto;
    void (*fn)(job*, int, int);
    int count, quit;
} g_pool;
static void run_slice(int k) {
    job j = g_pool.proto;
    const int lo = (int)((int64_t)g_pool.count * k / g_pool.n), hi = (int)((int64_t)g_pool.count * (k + 1) / g_pool.n);
    if (hi > lo) g_pool.fn(&j, lo, hi);
}
static void* worker(void* arg) {
    const int k = (int)(intptr_t)arg;
    for (;;) {
        pthread_barrier_wait(&g_pool.start);
        if (g_pool.quit) return NULL;
        run_slice(k);
        pthread_barrier_wait(&g_pool.done);
    }
}
static void pool_resize(int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (g_pool.n == nthreads) return;
    if (g_pool.n > 1) {  /* retire the old workers */
        g_pool.quit = 1;
        pthread_barrier_wait(&g_pool.start);
        for (int k = 1; k < g_pool.n; k++) pthread_join(g_pool.th[k], NULL);
        pthread_barrier_destroy(&g_pool.start);
        pthread_barrier_destroy(&g_pool.done);
        free(g_pool.th);
        g_pool.quit = 0;
    }
    g_pool.n = nthreads;
    if (nthreads > 1) {
        g_pool.th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
        pthread_barrier_init(&g_pool.start, NULL, nthreads);
        pthread_barrier_init(&g_pool.done, NULL, nthreads);
        for (int k = 1; k < nthreads; k++) pthread_create(&g_pool.th[k], NULL, worker, (void*)(intptr_t)k);
    }
}
static void parallel_for(job* proto, void (*fn)(job*, int, int), int n, int nthreads) {
    pool_resize(nthreads);
    g_pool.proto = *proto;
    g_pool.fn = fn;
    g_pool.count = n;
    if (g_pool.n > 1) pthread_barrier_wait(&g_pool.start);
    run_slice(0);
    if (g_pool.n > 1) pthread_barrier_wait(&g_pool.done);
}

/* in-place-result radix-2 Stockham, unnormalised, e^{+...}: x (length N) -> x, y is scratch */
static void fft1d(cpx* x, cpx* y, int N, const cpx* tw) {
    cpx *src = x, *dst = y;
    for (int half = N / 2, stride = 1; half >= 1; half >>= 1, stride <<= 1) {
        /* butterflies of span `half`: out[2*stride*q + s] , out[2*stride*q + s + stride] */
        for (int q = 0; q < half; q++) {
            const cpx w = tw[q * stride];
            for (int s = 0; s < stride; s++) {
                const cpx a = src[q * stride + s], b = src[(q + half) * stride + s];
                const cpx d = {a.re - b.re, a.im - b.im};
                dst[2 * q * stride + s].re = a.re + b.re;
                dst[2 * q * stride + s].im = a.im + b.im;
                dst[(2 * q + 1) * stride + s].re = d.re * w.re - d.im * w.im;
                dst[(2 * q + 1) * stride + s].im = d.re * w.im + d.im * w.re;
            }
        }
        cpx* tmp = src; src = dst; dst = tmp;
    }
    if (src != x) memcpy(x, src, sizeof(cpx) * N);
}

/* stage A: rows i in [lo,hi): h~(k,t) (S/FFTMesh.cs:178-190) and the five spectra (:211-215), times pre(i+j) */
static void stage_spectra(job* J, int lo, int hi) {
    const int N = J->N;
    const size_t NN = (size_t)N * N;
    const float L = J->p->length;
    for (int i = lo; i < hi; i++) {
        const float kx = 2.0f * 3.1415926536f * ((float)i - N / 2.0f) / L; /* :201 */
        for (int j = 0; j < N; j++) {
            const float kz = 2.0f * 3.1415926536f * ((float)j - N / 2.0f) / L; /* :204 */
            const size_t idx = (size_t)i * N + j;
            const float wt = orc_dispersion(J->p, i, j) * J->t; /* :183 */
            const float c = cosf(wt), s = sinf(wt);
            const float ax = J->h0[2 * idx], ay = J->h0[2 * idx + 1], bx = J->h0c[2 * idx], by = J->h0c[2 * idx + 1];
            float hr = ax * c - ay * s + bx * c + by * s; /* :188 */
            float hi_ = ax * s + ay * c - bx * s + by * c;
            /* pre(i+j) = (-1)^(i+j) e^{i pi (i+j)/N} */
            const int m = i + j;
            const float ang = 3.14159265358979f * (float)m / (float)N;
            float pr = cosf(ang), pi_ = sinf(ang);
            if (m & 1) { pr = -pr; pi_ = -pi_; }
            const float tr = hr * pr - hi_ * pi_, ti = hr * pi_ + hi_ * pr;
            const float kl = sqrtf(kx * kx + kz * kz);
            float ux = 0.f, uz = 0.f;
            if (!(kl < 0.0001f)) { ux = kx / kl; uz = -kz / kl; } /* :213-215 */
            const float mult[5] = {1.f, ux, uz, kx, kz};
            for (int f = 0; f < 5; f++) {
                J->A[f * NN + idx].re = mult[f] * tr;
                J->A[f * NN + idx].im = mult[f] * ti;
            }
        }
    }
}
/* stage B: 1-D transforms along j of rows [lo,hi) of the 5*N row set */
static void stage_rows(job* J, int lo, int hi) {
    const int N = J->N;
    cpx* scratch = (cpx*)malloc(sizeof(cpx) * N);
    for (int r = lo; r < hi; r++) fft1d(J->A + (size_t)r * N, scratch, N, J->tw);
    free(scratch);
}
/* stage C: 1-D transforms along i, 8 columns at a time (gather / transform / scatter) over the 5*N/8 column groups */
static void stage_cols(job* J, int lo, int hi) {
    const int N = J->N;
    const size_t NN = (size_t)N * N;
    cpx* buf = (cpx*)malloc(sizeof(cpx) * N * 9);
    for (int g = lo; g < hi; g++) {
        const int f = g / (N / 8), j0 = (g % (N / 8)) * 8;
        cpx* F = J->A + f * NN;
        for (int i = 0; i < N; i++)
            for (int c = 0; c < 8; c++) buf[(size_t)c * N + i] = F[(size_t)i * N + j0 + c];
        for (int c = 0; c < 8; c++) fft1d(buf + (size_t)c * N, buf + (size_t)8 * N, N, J->tw);
        for (int i = 0; i < N; i++)
            for (int c = 0; c < 8; c++) F[(size_t)i * N + j0 + c] = buf[(size_t)c * N + i];
    }
    free(buf);
}
static inline float rest_pos(const orc_params* p, int a) { /* S/FFTMesh.cs:107-112 */
    return (float)(a - p->N / 2) * p->unit_width + ((p->N % 2 == 0) ? p->unit_width / 2.0f : 0.0f);
}
/* stage D1: rows a in [lo,hi): vertices, normals, hds (S/FFTMesh.cs:218-219, 243-247) */
static void stage_vertices(job* J, int lo, int hi) {
    const int N = J->N;
    const size_t NN = (size_t)N * N;
    for (int a = lo; a < hi; a++)
        for (int b = 0; b < N; b++) {
            const size_t idx = (size_t)a * N + b;
            const float sg = ((a + b) & 1) ? 1.f : -1.f; /* -(-1)^(a+b) */
            const float h = sg * J->A[0 * NN + idx].re, dx = sg * J->A[1 * NN + idx].im, dz = sg * J->A[2 * NN + idx].im;
            const float sx = sg * J->A[3 * NN + idx].im, sz = sg * J->A[4 * NN + idx].im;
            const float inv = 1.0f / sqrtf(sx * sx + 1.0f + sz * sz); /* :218 */
            J->normals[3 * idx] = sx * inv; J->normals[3 * idx + 1] = inv; J->normals[3 * idx + 2] = sz * inv;
            J->vertices[3 * idx] = rest_pos(J->p, a) - dx * J->p->choppiness;      /* :245 */
            J->vertices[3 * idx + 1] = h;                                            /* :243 */
            J->vertices[3 * idx + 2] = rest_pos(J->p, b) - dz * J->p->choppiness;  /* :244 */
            J->hds[2 * idx] = dx; J->hds[2 * idx + 1] = dz;                          /* :247 */
        }
}
/* stage D2: Jacobian / whitecap (S/FFTMesh.cs:251-276) */
static void stage_white(job* J, int lo, int hi) {
    const int N = J->N;
    const float* hds = J->hds;
    for (int i = lo; i < hi; i++)
        for (int j = 0; j < N; j++) {
            const size_t index = (size_t)i * N + j;
            float ax = 0, ay = 0, bx = 0, by = 0;
            if (i != N - 1) { ax = 0.5f * (hds[2 * index] - hds[2 * (index + N)]); ay = 0.5f * (hds[2 * index + 1] - hds[2 * (index + N) + 1]); }
            if (j != N - 1) { bx = 0.5f * (hds[2 * index] - hds[2 * (index + 1)]); by = 0.5f * (hds[2 * index + 1] - hds[2 * (index + 1) + 1]); }
            const float jac = (1 + ax) * (1 + by) - ay * bx;
            const float n0 = fabsf(J->normals[3 * index]) * 0.3f, n1 = fabsf(J->normals[3 * index + 2]) * 0.3f;
            float turb = 1.0f - jac + sqrtf(n0 * n0 + n1 * n1);
            turb = turb < 0 ? 0 : (turb > 1 ? 1 : turb);
            const float xx = -2.0f * turb * turb * turb + 3.0f * turb * turb;
            float* c = J->colors + 4 * index;
            c[0] = c[1] = c[2] = c[3] = xx;
        }
}

/* one EvaluateWaves(t).  Returns 0, 1 (N not a power of two >= 8), 2 (grid not commensurate), 3 (out of memory). */
int orc_cpu_fft_step_f32(const orc_params* p, const float* h0, const float* h0c, float t, int nthreads, float* vertices,
                         float* normals, float* colors) {
    const int N = p->N;
    if (N < 8 || (N & (N - 1))) return 1;
    if (fabsf(p->unit_width * (float)N - p->length) > 1e-6f * p->length) return 2;
    const size_t NN = (size_t)N * N;
    job J;
    memset(&J, 0, sizeof(J));
    J.p = p; J.h0 = h0; J.h0c = h0c; J.t = t; J.N = N;
    J.vertices = vertices; J.normals = normals; J.colors = colors;
    /* workspace kept across calls (48 MiB at 1024^2: mapping and first-touching it every step costs more than the step) */
    static cpx *ws_tw = NULL, *ws_A = NULL;
    static float* ws_hds = NULL;
    static int ws_N = 0;
    if (ws_N != N) {
        free(ws_tw); free(ws_A); free(ws_hds);
        ws_tw = (cpx*)malloc(sizeof(cpx) * (N / 2));
        ws_A = (cpx*)malloc(sizeof(cpx) * 5 * NN);
        ws_hds = (float*)malloc(sizeof(float) * 2 * NN);
        ws_N = (ws_tw && ws_A && ws_hds) ? N : 0;
        if (!ws_N) { free(ws_tw); free(ws_A); free(ws_hds); ws_tw = ws_A = NULL; ws_hds = NULL; return 3; }
        for (int k = 0; k < N / 2; k++) {
            ws_tw[k].re = (float)cos(2.0 * M_PI * k / N);
            ws_tw[k].im = (float)sin(2.0 * M_PI * k / N);
        }
    }
    cpx* tw = ws_tw;
    J.A = ws_A;
    J.hds = ws_hds;
    J.tw = tw;
    double tm[6];
    tm[0] = now_s();
    parallel_for(&J, stage_spectra, N, nthreads);  tm[1] = now_s();
    parallel_for(&J, stage_rows, 5 * N, nthreads); tm[2] = now_s();
    parallel_for(&J, stage_cols, 5 * (N / 8), nthreads); tm[3] = now_s();
    parallel_for(&J, stage_vertices, N, nthreads); tm[4] = now_s();
    parallel_for(&J, stage_white, N, nthreads);    tm[5] = now_s();
    if (getenv("ORC_CPU_FFT_VERBOSE"))
        fprintf(stderr, "cpu_fft N=%d threads=%d: spectra %.1f rows %.1f cols %.1f vertices %.1f white %.1f ms\n", N, nthreads,
                1e3 * (tm[1] - tm[0]), 1e3 * (tm[2] - tm[1]), 1e3 * (tm[3] - tm[2]), 1e3 * (tm[4] - tm[3]), 1e3 * (tm[5] - tm[4]));
    return 0;
}
