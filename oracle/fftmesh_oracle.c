/*
 * oracle/fftmesh_oracle.c  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the reference's CPU "FFT Mesh" ocean path
 *   /root/reference/Assets/Mistral Water/Scripts/FFTMesh.cs   (cited below as S/FFTMesh.cs:LINE)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product library (libmistral_water.so) never links or calls it.
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures
 *  and cannot be compiled or run here (C# + closed-source
 * UnityEngine.dll, no mono/dotnet).  This restatement is therefore checked only
 * against (i) the closed-form anchor values of SURVEY.md section 8a, (ii) analytic
 * cases (zero spectrum, single mode) and (iii) its own independent variants
 * (literal O(N^4) f32  vs  separable O(N^3) f64  vs  numpy FFT f64).
 *
 * Three evaluators are provided:
 *   orc_eval_literal_f32   line-by-line S/FFTMesh.cs:192-280 in IEEE float32 with
 *                          UnityEngine.Mathf semantics ((float)libm((double)x)); O(N^4).
 *   orc_eval_f64           same model, every accumulation and transcendental in double,
 *                          separable O(N^3); works for non-commensurate grids too.
 *   orc_htilde_fields_f64 / orc_assemble_f64
 *                          the two halves of orc_eval_f64 around the 2-D transform, so
 *                          that oracle_np.py can put a numpy FFT in the middle for N=1024+.
 * In all of them the *discontinuous / amplified* spectral scalars -- the quantised
 * dispersion floor() (S/FFTMesh.cs:146) and the product omega*t (S/FFTMesh.cs:183) --
 * are evaluated in strict float32 exactly as the reference's float code does, because a
 * flipped floor() changes a phase by 2*pi*t/length, which no tolerance can absorb.
 *
 * Build:  gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fPIC -shared (see oracle/Makefile)
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* S/FFTMesh.cs:50-54 */
#define ORC_PI_F 3.1415926536f
#define ORC_EPS_F 0.0001f

typedef struct {
    int32_t N;          /* resolution  S/FFTMesh.cs:13 */
    float unit_width;   /* unitWidth   :15 */
    float length;       /* length      :19 */
    float wind_x;       /* wind.x      :21 */
    float wind_y;       /* wind.y      :21 */
    float amplitude;    /* amplitude   :23 */
    float choppiness;   /* choppiness  :9  */
    float gravity;      /* G = 9.81f   :52 (a constant in the reference; a parameter here) */
} orc_params;

/* ---- UnityEngine.Mathf: double libm rounded once to float [unity] ------------------- */
static inline float mf_sqrt(float x) { return (float)sqrt((double)x); }
static inline float mf_sin(float x) { return (float)sin((double)x); }
static inline float mf_cos(float x) { return (float)cos((double)x); }
static inline float mf_exp(float x) { return (float)exp((double)x); }
static inline float mf_log(float x) { return (float)log((double)x); }
static inline float mf_floor(float x) { return (float)floor((double)x); }
static inline float mf_clamp01(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }
/* Mathf.SmoothStep(from,to,t): t is clamped, NOT rescaled by (from,to) [unity] */
static inline float mf_smoothstep(float from, float to, float t) {
    t = mf_clamp01(t);
    t = -2.0f * t * t * t + 3.0f * t * t;
    return to * t + from * (1.0f - t);
}

/* ---- the build's own counter-based RNG (Unity's Random.value is closed source) ------ */
/* uniform in (0,1]: never 0, so Log(z1) is finite (the reference is unguarded, :173).  */
static inline uint64_t orc_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
float orc_uniform(uint64_t seed, uint64_t counter) {
    uint64_t bits = orc_mix64(seed * 0xD1342543DE82EF95ull + counter);
    return (float)((uint32_t)(bits >> 40) + 1u) * (1.0f / 16777216.0f);
}

/* ---- S/FFTMesh.cs:141-147  Dispersion(n,m): quantised deep-water omega, strict f32 ---- */
float orc_dispersion(const orc_params* p, int n, int m) {
    float w = 2 * ORC_PI_F / p->length;
    float kx = ORC_PI_F * (float)(2 * n - p->N) / p->length;
    float kz = ORC_PI_F * (float)(2 * m - p->N) / p->length;
    float kx2 = kx * kx, kz2 = kz * kz;
    float s = kx2 + kz2;
    return mf_floor(mf_sqrt(p->gravity * mf_sqrt(s)) / w) * w;
}

/* Dispersion(n,m) * t over the whole grid, out[n * N + m] -- the argument of :184-185 exactly as htilde() forms it (:183, one
 * float32 product).  The GPU tests compare the device's omega*t with THIS over every grid point (index work: bit for bit).   */
void orc_dispersion_grid(const orc_params* p, float t, float* out) {
    for (int n = 0; n < p->N; n++)
        for (int m = 0; m < p->N; m++) out[(size_t)n * p->N + m] = orc_dispersion(p, n, m) * t; /* :183 */
}

/* ---- S/FFTMesh.cs:149-166  Phillips(n,m) ---------------------------------------------- */
float orc_phillips(const orc_params* p, int n, int m) {
    float kx = (float)(2 * n - p->N) / p->length * ORC_PI_F; /* :151 */
    float kz = (float)(2 * m - p->N) / p->length * ORC_PI_F;
    float k_length = mf_sqrt(kx * kx + kz * kz); /* Vector2.magnitude :152 */
    if (k_length < ORC_EPS_F) return 0.0f;       /* :153 */
    float k_length2 = k_length * k_length;
    float k_length4 = k_length2 * k_length2;
    /* Vector2.normalized: v / magnitude when magnitude > 1e-5, else zero [unity] */
    float knx = 0.f, knz = 0.f;
    if (k_length > 1e-5f) { knx = kx / k_length; knz = kz / k_length; }
    float w_length = mf_sqrt(p->wind_x * p->wind_x + p->wind_y * p->wind_y); /* :160 */
    float wnx = 0.f, wny = 0.f;
    if (w_length > 1e-5f) { wnx = p->wind_x / w_length; wny = p->wind_y / w_length; }
    float kDotW = knx * wnx + knz * wny; /* :158 */
    float kDotW2 = kDotW * kDotW;
    float l = w_length * w_length / p->gravity; /* :161 */
    float l2 = l * l;
    float damping = 0.001f; /* :163 */
    float L2 = l2 * damping * damping;
    return p->amplitude * mf_exp(-1.f / (k_length2 * l2)) / k_length4 * kDotW2 * mf_exp(-k_length2 * L2); /* :165 */
}

/* ---- S/FFTMesh.cs:168-176  htilde0(n,m) with the two uniforms supplied ----------------- */
void orc_htilde0(const orc_params* p, int n, int m, float z1, float z2, float out[2]) {
    float rad = mf_sqrt(-2.f * mf_log(z1));
    float rx = rad * mf_cos(2 * ORC_PI_F * z2);
    float ry = rad * mf_sin(2 * ORC_PI_F * z2);
    float sc = mf_sqrt(orc_phillips(p, n, m) / 2.f);
    out[0] = rx * sc;
    out[1] = ry * sc;
}

/* ---- S/FFTMesh.cs:101-116  spectrum fill order: 4 draws per grid point, row-major ------ */
void orc_generate_spectrum(const orc_params* p, uint64_t seed, float* h0, float* h0conj) {
    int N = p->N;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            uint64_t idx = (uint64_t)i * N + j; /* :110 */
            float a[2], b[2];
            orc_htilde0(p, i, j, orc_uniform(seed, 4 * idx + 0), orc_uniform(seed, 4 * idx + 1), a); /* :114 */
            orc_htilde0(p, N - i, N - j, orc_uniform(seed, 4 * idx + 2), orc_uniform(seed, 4 * idx + 3), b); /* :115 */
            h0[2 * idx] = a[0];
            h0[2 * idx + 1] = a[1];
            h0conj[2 * idx] = b[0];
            h0conj[2 * idx + 1] = -b[1]; /* :116 */
        }
}

/* ---- S/FFTMesh.cs:101-139  GenerateMesh: rest positions, normals, uvs, indices --------- */
/* returns the number of indices written ( = (N-1)^2*6 ) */
int orc_rest_mesh(const orc_params* p, float* vertices, float* normals, float* uvs, int32_t* indices) {
    int N = p->N;
    int indiceCount = 0;
    int half = N / 2; /* :104 */
    for (int i = 0; i < N; i++) {
        float hpos = (float)(i - half) * p->unit_width; /* :107 */
        for (int j = 0; j < N; j++) {
            int cur = i * N + j;
            float vpos = (float)(j - half) * p->unit_width; /* :111 */
            float off = (N % 2 == 0) ? p->unit_width / 2.f : 0.f;
            if (vertices) {
                vertices[3 * cur + 0] = hpos + off; /* :112 */
                vertices[3 * cur + 1] = 0.f;
                vertices[3 * cur + 2] = vpos + off;
            }
            if (normals) { normals[3 * cur] = 0.f; normals[3 * cur + 1] = 1.f; normals[3 * cur + 2] = 0.f; }
            if (uvs) {
                uvs[2 * cur + 0] = (float)i * 1.0f / (float)(N - 1); /* :117 */
                uvs[2 * cur + 1] = (float)j * 1.0f / (float)(N - 1);
            }
            if (j == N - 1) continue; /* :118 */
            if (i != N - 1) {         /* :120-125 */
                if (indices) { indices[indiceCount] = cur; indices[indiceCount + 1] = cur + 1; indices[indiceCount + 2] = cur + N; }
                indiceCount += 3;
            }
            if (i != 0) { /* :126-131 */
                if (indices) { indices[indiceCount] = cur; indices[indiceCount + 1] = cur - N + 1; indices[indiceCount + 2] = cur + 1; }
                indiceCount += 3;
            }
        }
    }
    return indiceCount;
}

/* ---- S/FFTMesh.cs:178-190  htilde(t,n,m), strict f32 ----------------------------------- */
static inline void htilde_f32(const orc_params* p, const float* h0, const float* h0c, float t, int n, int m, float out[2]) {
    int index = n * p->N + m;
    float h0x = h0[2 * index], h0y = h0[2 * index + 1];
    float cx = h0c[2 * index], cy = h0c[2 * index + 1];
    float omegat = orc_dispersion(p, n, m) * t; /* :183 */
    float _cos = mf_cos(omegat), _sin = mf_sin(omegat);
    float c0x = _cos, c0y = _sin, c1x = _cos, c1y = -_sin;
    out[0] = h0x * c0x - h0y * c0y + cx * c1x - cy * c1y; /* :188 */
    out[1] = h0x * c0y + h0y * c0x + cx * c1y + cy * c1x;
}

/* ---- S/FFTMesh.cs:192-220  Displacement(x,t,out nor), strict f32, O(N^2) per vertex ---- */
void orc_displacement_f32(const orc_params* p, const float* h0, const float* h0c, float x, float z, float t,
                          float hd[3], float nor[3]) {
    int N = p->N;
    float hx = 0.f, hy = 0.f, dx = 0.f, dy = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
    for (int i = 0; i < N; i++) {
        float kx = 2 * ORC_PI_F * ((float)i - (float)N / 2.0f) / p->length; /* :201 */
        for (int j = 0; j < N; j++) {
            float kz = 2 * ORC_PI_F * ((float)j - (float)N / 2.0f) / p->length; /* :204 */
            float k_length = mf_sqrt(kx * kx + kz * kz);                        /* :206 */
            float kDotX = kx * x + kz * z;                                      /* :207 */
            float cx = mf_cos(kDotX), cy = mf_sin(kDotX);                       /* :208 */
            float tmp[2];
            htilde_f32(p, h0, h0c, t, i, j, tmp); /* :209 */
            float hcx = tmp[0] * cx - tmp[1] * cy, hcy = tmp[0] * cy + tmp[1] * cx; /* :210 */
            hx += hcx; hy += hcy;                                                /* :211 */
            nx += -kx * hcy; ny += 0.f; nz += -kz * hcy;                         /* :212 */
            if (k_length < ORC_EPS_F) continue;                                  /* :213 */
            dx += kx / k_length * hcy;                                           /* :215 */
            dy += -kz / k_length * hcy;                                          /* :215 note the sign */
        }
    }
    /* :218 nor = Vector3.Normalize(Vector3.up - n) */
    float ux = 0.f - nx, uy = 1.f - ny, uz = 0.f - nz;
    float mag = mf_sqrt(ux * ux + uy * uy + uz * uz);
    if (mag > 1e-5f) { nor[0] = ux / mag; nor[1] = uy / mag; nor[2] = uz / mag; }
    else { nor[0] = nor[1] = nor[2] = 0.f; }
    (void)hy;
    hd[0] = dx; hd[1] = hx; hd[2] = dy; /* :219 */
}

/* ---- S/FFTMesh.cs:251-276  Jacobian / whitecap colour, strict f32 ---------------------- */
static void whitecap_f32(int N, const float* hds, const float* normals, float* colors /* N*N*4 */) {
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            int index = i * N + j;
            float dDdx_x = 0.f, dDdx_y = 0.f, dDdy_x = 0.f, dDdy_y = 0.f;
            if (i != N - 1) { /* :260-263 */
                dDdx_x = 0.5f * (hds[2 * index] - hds[2 * (index + N)]);
                dDdx_y = 0.5f * (hds[2 * index + 1] - hds[2 * (index + N) + 1]);
            }
            if (j != N - 1) { /* :264-267 */
                dDdy_x = 0.5f * (hds[2 * index] - hds[2 * (index + 1)]);
                dDdy_y = 0.5f * (hds[2 * index + 1] - hds[2 * (index + 1) + 1]);
            }
            float jacobian = (1 + dDdx_x) * (1 + dDdy_y) - dDdx_y * dDdy_x; /* :268 */
            float nzx = fabsf(normals[3 * index]) * 0.3f, nzy = fabsf(normals[3 * index + 2]) * 0.3f; /* :269 */
            float turb = fmaxf(1.f - jacobian + mf_sqrt(nzx * nzx + nzy * nzy), 0.f); /* :270 */
            float xx = mf_smoothstep(0.f, 1.f, turb); /* :273 (271-272 are dead stores) */
            colors[4 * index] = colors[4 * index + 1] = colors[4 * index + 2] = colors[4 * index + 3] = xx; /* :274 */
        }
}

/* ---- S/FFTMesh.cs:224-280  EvaluateWaves(t), literal O(N^4), strict f32 ---------------- */
void orc_eval_literal_f32(const orc_params* p, const float* h0, const float* h0c, float t,
                          float* vertices_out, float* normals_out, float* colors_out) {
    int N = p->N;
    float* rest = (float*)malloc(sizeof(float) * 3 * N * N);
    float* hds = (float*)malloc(sizeof(float) * 2 * N * N);
    orc_rest_mesh(p, rest, NULL, NULL, NULL);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            int index = i * N + j;
            float hd[3], nor[3];
            orc_displacement_f32(p, h0, h0c, rest[3 * index], rest[3 * index + 2], t, hd, nor); /* :242 */
            vertices_out[3 * index + 1] = hd[1];                                       /* :243 */
            vertices_out[3 * index + 2] = rest[3 * index + 2] - hd[2] * p->choppiness; /* :244 */
            vertices_out[3 * index + 0] = rest[3 * index + 0] - hd[0] * p->choppiness; /* :245 */
            normals_out[3 * index] = nor[0]; normals_out[3 * index + 1] = nor[1]; normals_out[3 * index + 2] = nor[2];
            hds[2 * index] = hd[0]; hds[2 * index + 1] = hd[2]; /* :247 un-scaled by choppiness */
        }
    whitecap_f32(N, hds, normals_out, colors_out);
    free(rest); free(hds);
}

/* literal f32 Displacement at a caller-chosen subset of vertices (spot checks at large N and the
 * bench's bounded cpu_baseline sample).  Outputs: hd[3*k] = (d.x, h, d.z), nor[3*k].            */
void orc_displacement_subset_f32(const orc_params* p, const float* h0, const float* h0c, float t,
                                 const int32_t* vertex_idx, int count, float* hd_out, float* nor_out) {
    int N = p->N;
    int half = N / 2;
    float off = (N % 2 == 0) ? p->unit_width / 2.f : 0.f;
    for (int k = 0; k < count; k++) {
        int i = vertex_idx[k] / N, j = vertex_idx[k] % N;
        float x = (float)(i - half) * p->unit_width + off, z = (float)(j - half) * p->unit_width + off;
        orc_displacement_f32(p, h0, h0c, x, z, t, hd_out + 3 * k, nor_out + 3 * k);
    }
}

/* =====================================  f64 path  ======================================= */

/* h~(k,t) and the five spectra whose transforms give the five real output fields.
 * fields layout: [5][N*N][2] doubles, field order H, Dx, Dz, Sx, Sz where (S/FFTMesh.cs:211-215)
 *   H  = h~                      -> height = Re F(H)
 *   Dx = (kx/|k|) h~ (0 if |k|<EPS)   -> d.x  = Im F(Dx)
 *   Dz = (-kz/|k|) h~                 -> d.y  = Im F(Dz)      (the sign quirk of :215)
 *   Sx = kx h~ , Sz = kz h~           -> n = -(Im F(Sx), 0, Im F(Sz)),  nor = normalize(up - n)
 * k uses exact pi in double; omega*t follows the f32 path bit for bit (see file header).      */
void orc_htilde_fields_f64(const orc_params* p, const float* h0, const float* h0c, float t, double* fields) {
    int N = p->N;
    size_t NN = (size_t)N * N;
    for (int i = 0; i < N; i++) {
        double kx = 2.0 * M_PI * ((double)i - N / 2.0) / (double)p->length; /* :201 */
        for (int j = 0; j < N; j++) {
            double kz = 2.0 * M_PI * ((double)j - N / 2.0) / (double)p->length; /* :204 */
            size_t idx = (size_t)i * N + j;
            float omegat = orc_dispersion(p, i, j) * t; /* :183 strict f32 */
            double c = cos((double)omegat), s = sin((double)omegat);
            double ax = h0[2 * idx], ay = h0[2 * idx + 1], bx = h0c[2 * idx], by = h0c[2 * idx + 1];
            double hr = ax * c - ay * s + bx * c + by * s; /* :188 with c1 = (c,-s) */
            double hi = ax * s + ay * c - bx * s + by * c;
            double kl = sqrt(kx * kx + kz * kz);
            double ux = 0.0, uz = 0.0;
            if (!(kl < (double)ORC_EPS_F)) { ux = kx / kl; uz = -kz / kl; } /* :213-215 */
            double* f;
            f = fields + 2 * (0 * NN + idx); f[0] = hr; f[1] = hi;
            f = fields + 2 * (1 * NN + idx); f[0] = ux * hr; f[1] = ux * hi;
            f = fields + 2 * (2 * NN + idx); f[0] = uz * hr; f[1] = uz * hi;
            f = fields + 2 * (3 * NN + idx); f[0] = kx * hr; f[1] = kx * hi;
            f = fields + 2 * (4 * NN + idx); f[0] = kz * hr; f[1] = kz * hi;
        }
    }
}

/* rest position of grid line a, in double, S/FFTMesh.cs:107-112 */
static inline double rest_pos(const orc_params* p, int a) {
    return (double)(a - p->N / 2) * (double)p->unit_width + ((p->N % 2 == 0) ? (double)p->unit_width / 2.0 : 0.0);
}

/* separable direct transform  out_f(a,b) = sum_i sum_j F_f(i,j) e^{i(kx_i x_a + kz_j z_b)},
 * O(N^3); valid for any unit_width / length (commensurate or not).  spatial: [5][N*N][2].     */
void orc_transform_direct_f64(const orc_params* p, const double* fields, double* spatial) {
    int N = p->N;
    size_t NN = (size_t)N * N;
    double* E = (double*)malloc(sizeof(double) * 2 * NN); /* E[j][b] = e^{i k_j pos_b} (same both axes) */
    for (int j = 0; j < N; j++) {
        double k = 2.0 * M_PI * ((double)j - N / 2.0) / (double)p->length;
        for (int b = 0; b < N; b++) {
            double ph = k * rest_pos(p, b);
            E[2 * ((size_t)j * N + b)] = cos(ph);
            E[2 * ((size_t)j * N + b) + 1] = sin(ph);
        }
    }
    double* tmp = (double*)malloc(sizeof(double) * 2 * NN);
    for (int f = 0; f < 5; f++) {
        const double* F = fields + 2 * f * NN;
        double* O = spatial + 2 * f * NN;
        memset(tmp, 0, sizeof(double) * 2 * NN);
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                double fr = F[2 * ((size_t)i * N + j)], fi = F[2 * ((size_t)i * N + j) + 1];
                if (fr == 0.0 && fi == 0.0) continue;
                const double* e = E + 2 * (size_t)j * N;
                double* trow = tmp + 2 * (size_t)i * N;
                for (int b = 0; b < N; b++) {
                    trow[2 * b] += fr * e[2 * b] - fi * e[2 * b + 1];
                    trow[2 * b + 1] += fr * e[2 * b + 1] + fi * e[2 * b];
                }
            }
        memset(O, 0, sizeof(double) * 2 * NN);
        for (int i = 0; i < N; i++)
            for (int a = 0; a < N; a++) {
                double er = E[2 * ((size_t)i * N + a)], ei = E[2 * ((size_t)i * N + a) + 1];
                const double* trow = tmp + 2 * (size_t)i * N;
                double* orow = O + 2 * (size_t)a * N;
                for (int b = 0; b < N; b++) {
                    orow[2 * b] += trow[2 * b] * er - trow[2 * b + 1] * ei;
                    orow[2 * b + 1] += trow[2 * b] * ei + trow[2 * b + 1] * er;
                }
            }
    }
    free(E); free(tmp);
}

/* S/FFTMesh.cs:218-219, 243-247, 251-276 in double: spatial fields -> vertices/normals/colours.
 * spatial: [5][N*N][2] complex (H,Dx,Dz,Sx,Sz); hds_out (optional) = (d.x, d.z) per vertex.    */
void orc_assemble_f64(const orc_params* p, const double* spatial, double* vertices, double* normals,
                      double* colors, double* hds_out) {
    int N = p->N;
    size_t NN = (size_t)N * N;
    double* hds = hds_out ? hds_out : (double*)malloc(sizeof(double) * 2 * NN);
    for (int a = 0; a < N; a++)
        for (int b = 0; b < N; b++) {
            size_t idx = (size_t)a * N + b;
            double h = spatial[2 * (0 * NN + idx)];         /* Re  :219 h.x */
            double dx = spatial[2 * (1 * NN + idx) + 1];    /* Im  :215 */
            double dz = spatial[2 * (2 * NN + idx) + 1];
            double sx = spatial[2 * (3 * NN + idx) + 1];    /* n.x = -sx  :212 */
            double sz = spatial[2 * (4 * NN + idx) + 1];
            double ux = sx, uy = 1.0, uz = sz;              /* up - n  :218 */
            double mag = sqrt(ux * ux + uy * uy + uz * uz);
            normals[3 * idx] = ux / mag; normals[3 * idx + 1] = uy / mag; normals[3 * idx + 2] = uz / mag;
            vertices[3 * idx + 1] = h;                                                  /* :243 */
            vertices[3 * idx + 2] = rest_pos(p, b) - dz * (double)p->choppiness;        /* :244 */
            vertices[3 * idx + 0] = rest_pos(p, a) - dx * (double)p->choppiness;        /* :245 */
            hds[2 * idx] = dx; hds[2 * idx + 1] = dz;                                   /* :247 */
        }
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            size_t index = (size_t)i * N + j;
            double ax = 0, ay = 0, bx = 0, by = 0;
            if (i != N - 1) { ax = 0.5 * (hds[2 * index] - hds[2 * (index + N)]); ay = 0.5 * (hds[2 * index + 1] - hds[2 * (index + N) + 1]); }
            if (j != N - 1) { bx = 0.5 * (hds[2 * index] - hds[2 * (index + 1)]); by = 0.5 * (hds[2 * index + 1] - hds[2 * (index + 1) + 1]); }
            double jac = (1 + ax) * (1 + by) - ay * bx;                                      /* :268 */
            double n0 = fabs(normals[3 * index]) * 0.3, n1 = fabs(normals[3 * index + 2]) * 0.3; /* :269 */
            double turb = fmax(1.0 - jac + sqrt(n0 * n0 + n1 * n1), 0.0);                    /* :270 */
            double tt = turb < 0 ? 0 : (turb > 1 ? 1 : turb);
            double xx = -2.0 * tt * tt * tt + 3.0 * tt * tt;                                 /* :273 */
            colors[4 * index] = colors[4 * index + 1] = colors[4 * index + 2] = colors[4 * index + 3] = xx;
        }
    if (!hds_out) free(hds);
}

/* full f64 evaluation, separable O(N^3) */
void orc_eval_f64(const orc_params* p, const float* h0, const float* h0c, float t,
                  double* vertices, double* normals, double* colors) {
    size_t NN = (size_t)p->N * p->N;
    double* fields = (double*)malloc(sizeof(double) * 10 * NN);
    double* spatial = (double*)malloc(sizeof(double) * 10 * NN);
    orc_htilde_fields_f64(p, h0, h0c, t, fields);
    orc_transform_direct_f64(p, fields, spatial);
    orc_assemble_f64(p, spatial, vertices, normals, colors, NULL);
    free(fields); free(spatial);
}

/* strict-f32 whitecap on caller-supplied hds/normals: bit-exact check of the edge handling */
void orc_whitecap_f32(int32_t N, const float* hds, const float* normals, float* colors) { whitecap_f32(N, hds, normals, colors); }
