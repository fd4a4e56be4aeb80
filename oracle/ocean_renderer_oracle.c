/*
 * oracle/ocean_renderer_oracle.c  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the reference's fragment-shader ocean pipeline (semantics "B"):
 *   S/OceanRenderer.cs:116-170,209-316  (host scheduling: 1 + 45 full-screen passes per frame)
 *   F/FFTCommon.cginc, F/InitialSpectrum.shader, F/Dispersion.shader, F/Spectrum.shader,
 *   F/SpectrumHeight.shader, F/Stockham.shader, F/OceanNormal.shader, F/WhiteCap.shader
 * Textures are M x M with M = 8*resolution (S/OceanRenderer.cs:136); texel (px,py) lives at index py*M + px and
 * its texcoord is ((px+.5)/M, (py+.5)/M).
 *
 * PARITY UNPINNED (see fftmesh_oracle.c): the shaders ran inside Unity's renderer; nothing here can be checked
 * against an execution of the reference.  Deliberate, documented departures:
 *   - F/FFTCommon.cginc:37-41 UVRandom (a GPU-sin hash, not reproducible across devices) is replaced by the
 *     build's counter RNG; the clamp to [0.01,1] (:92-93) is kept.  _RandomSeed1/2 (Random.value*10,
 *     S/OceanRenderer.cs:147-148) are therefore unused.
 *   - HLSL fmod (F/FFTCommon.cginc:103) is taken as the exact IEEE remainder (fmodf).
 *   - render-texture addressing at the borders is Clamp (Unity's default wrap mode [unity]; not set in source).
 * The stateful phase (R32F ping-pong, F/Dispersion.shader:37-40) is float32 and is advanced in strict float32
 * in both evaluators, like omega*t in the FFTMesh oracle; everything else is double in orr_step_f64.
 *
 * The transform is performed by the LITERAL pass schedule of S/OceanRenderer.cs:229-262 over F/Stockham.shader:31-57
 * (2*log2(M) gather passes), which pins the survey's probe "pass sequence == forward unnormalised DFT".
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORR_PI_F 3.1415926536f /* F/FFTCommon.cginc:7 */
#define ORR_EPS_F 0.0001f      /* :8 */

typedef struct {
    int32_t resolution; /* mesh resolution; textures are 8x */
    float length, wind_x, wind_y, amplitude, choppiness, gravity, mult;
} orr_params;

float orc_uniform(uint64_t seed, uint64_t counter); /* fftmesh_oracle.c */

/* F/FFTCommon.cginc:58-67 GetWave: FFT-order k from the half-texel-centred coordinate */
static void get_wave_f(float n, float m, float len, float res, float* kx, float* kz) {
    n -= 0.5f; m -= 0.5f;
    n = (n < res * 0.5f) ? n : n - res;
    m = (m < res * 0.5f) ? m : m - res;
    *kx = 2 * ORR_PI_F * n / len;
    *kz = 2 * ORR_PI_F * m / len;
}

/* F/FFTCommon.cginc:69-85 Phillips, damping 0.01 */
float orr_phillips(const orr_params* p, float n, float m, int M) {
    float kx, kz;
    get_wave_f(n, m, p->length, (float)M, &kx, &kz);
    float klen = (float)sqrt((double)(kx * kx + kz * kz));
    float klen2 = klen * klen, klen4 = klen2 * klen2;
    if (klen < ORR_EPS_F) return 0.f;
    float wlen = (float)sqrt((double)(p->wind_x * p->wind_x + p->wind_y * p->wind_y));
    float kDotW = (kx / klen) * (p->wind_x / wlen) + (kz / klen) * (p->wind_y / wlen);
    float kDotW2 = kDotW * kDotW;
    float l = wlen * wlen / p->gravity, l2 = l * l;
    float damping = 0.01f, L2 = l2 * damping * damping;
    float amp = p->amplitude / 10000.f; /* S/OceanRenderer.cs:149 */
    return amp * (float)exp((double)(-1.f / (klen2 * l2))) / klen4 * kDotW2 * (float)exp((double)(-klen2 * L2));
}

/* F/FFTCommon.cginc:87-99 hTilde0 with the build RNG in place of UVRandom */
static void htilde0(float u1, float u2, float phi, float out[2]) {
    float r1 = u1 < 0.01f ? 0.01f : (u1 > 1.f ? 1.f : u1); /* :92-93 */
    float r2 = u2 < 0.01f ? 0.01f : (u2 > 1.f ? 1.f : u2);
    float x = (float)sqrt((double)(-2.f * (float)log((double)r1)));
    float y = 2 * ORR_PI_F * r2;
    float sc = (float)sqrt((double)(phi / 2.f));
    out[0] = x * (float)cos((double)y) * sc;
    out[1] = x * (float)sin((double)y) * sc;
}

/* F/InitialSpectrum.shader:42-54 -> RGBA = (h0.xy, conj(h0').xy), index py*M + px */
void orr_initial_spectrum(const orr_params* p, uint64_t seed, float* init4) {
    int M = p->resolution * 8;
    for (int py = 0; py < M; py++)
        for (int px = 0; px < M; px++) {
            uint64_t idx = (uint64_t)py * M + px;
            float n = (float)px + 0.5f, m = (float)py + 0.5f; /* texcoord * _Resolution */
            float phi1 = orr_phillips(p, n, m, M);
            float phi2 = orr_phillips(p, (float)M - n, (float)M - m, M); /* :48 -- texel M-1-px, the off-by-one mirror */
            float a[2], b[2];
            htilde0(orc_uniform(seed, 4 * idx + 0), orc_uniform(seed, 4 * idx + 1), phi1, a);
            htilde0(orc_uniform(seed, 4 * idx + 2), orc_uniform(seed, 4 * idx + 3), phi2, b);
            init4[4 * idx] = a[0]; init4[4 * idx + 1] = a[1];
            init4[4 * idx + 2] = b[0]; init4[4 * idx + 3] = -b[1]; /* Conj, :51 */
        }
}

/* F/FFTCommon.cginc:106-114 CalcDispersion (capillary form) and :101-104 GetDispersion; strict float32 */
float orr_phase_advance(const orr_params* p, int M, int px, int py, float old_phase, float dt) {
    float kx, kz;
    get_wave_f((float)px + 0.5f, (float)py + 0.5f, p->length, (float)M, &kx, &kz);
    float s = kx * kx + kz * kz;
    float wlen = (float)sqrt((double)s);
    float q = wlen * wlen / 370.f / 370.f;
    float inner = p->gravity * wlen * (1.f + q);
    float dphi = (float)sqrt((double)inner) * dt;
    float sum = old_phase + dphi;
    return fmodf(sum, 2 * ORR_PI_F);
}

/* one F/Stockham.shader pass over two packed complex signals (RGBA = A.xy, B.xy); horizontal != 0: along px */
static void stockham_pass(int M, int horizontal, double S, const double* in4, double* out4) {
    for (int py = 0; py < M; py++)
        for (int px = 0; px < M; px++) {
            double index = horizontal ? (double)px : (double)py;                          /* :36/:38 */
            double even = floor(index / S) * (S * 0.5) + fmod(index, S * 0.5);              /* :41 (+0.5 is the texel centre) */
            int e = (int)even, o = (int)(even + M * 0.5);
            const double* pe = horizontal ? in4 + 4 * ((size_t)py * M + e) : in4 + 4 * ((size_t)e * M + px);
            const double* po = horizontal ? in4 + 4 * ((size_t)py * M + o) : in4 + 4 * ((size_t)o * M + px);
            double tw = -2.0 * M_PI * (index / S);                                          /* :51, F/FFTCommon.cginc:116-119 */
            double c = cos(tw), s = sin(tw);
            double* q = out4 + 4 * ((size_t)py * M + px);
            q[0] = pe[0] + c * po[0] - s * po[1]; q[1] = pe[1] + c * po[1] + s * po[0];     /* :53 */
            q[2] = pe[2] + c * po[2] - s * po[3]; q[3] = pe[3] + c * po[3] + s * po[2];     /* :54 */
        }
}

/* S/OceanRenderer.cs:229-262: iterations = 2*ceil(log2 M); S = 2^((i mod log2M)+1); horizontal first */
static void stockham_2d(int M, double* a, double* b) {
    int lg = 0;
    while ((1 << lg) < M) lg++;
    int iterations = 2 * lg;
    double *src = a, *dst = b;
    for (int i = 0; i < iterations; i++) {
        double S = pow(2.0, (double)((i % (iterations / 2)) + 1)); /* :235 */
        stockham_pass(M, i < iterations / 2, S, src, dst);         /* :256-260 */
        double* t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, sizeof(double) * 4 * (size_t)M * M);
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* One OceanRenderer.GenerateTexture() (S/OceanRenderer.cs:216-316).  phase: float32 state [M*M], updated in
 * place.  Outputs (double): height [M*M] = heightTexture.r, disp [M*M*2] = displacementTexture.rb,
 * normal [M*M*3] = normalTexture.rgb, white [M*M] = whiteTexture.r; disp_g (optional) [M*M] = displacementTexture.g. */
void orr_step_f64(const orr_params* p, const float* init4, float* phase, float delta_time, double* height, double* disp,
                  double* normal, double* white, double* disp_g) {
    int M = p->resolution * 8;
    size_t MM = (size_t)M * M;
    double* spec = (double*)malloc(sizeof(double) * 4 * MM);
    double* tmp = (double*)malloc(sizeof(double) * 4 * MM);
    double* dtex = (double*)malloc(sizeof(double) * 4 * MM);
    double* htex = (double*)malloc(sizeof(double) * 4 * MM);
    float dt = delta_time * p->mult; /* :223 */
    /* Dispersion pass (:220-224) then Spectrum pass (:226-227, F/Spectrum.shader:34-51) */
    for (int py = 0; py < M; py++)
        for (int px = 0; px < M; px++) {
            size_t idx = (size_t)py * M + px;
            phase[idx] = orr_phase_advance(p, M, px, py, phase[idx], dt);
            double ph = (double)phase[idx], c = cos(ph), s = sin(ph);
            float kxf, kzf;
            get_wave_f((float)px + 0.5f, (float)py + 0.5f, p->length, (float)M, &kxf, &kzf);
            double kx = kxf, kz = kzf, w = sqrt(kx * kx + kz * kz);
            double ax = init4[4 * idx], ay = init4[4 * idx + 1], bx = init4[4 * idx + 2], by = init4[4 * idx + 3];
            double hr = ax * c - ay * s + bx * c + by * s;  /* h0*pv + h0conj*Conj(pv), :45 */
            double hi = ax * s + ay * c - bx * s + by * c;
            if (w < 0.0001) w = 0.0001;                     /* :47 */
            double fx = kx / w * (double)p->choppiness, fz = kz / w * (double)p->choppiness;
            /* hx = -MultByI(h * kx/w) * chop = (Im, -Re) * f   :48-49 */
            spec[4 * idx] = hi * fx; spec[4 * idx + 1] = -hr * fx;
            spec[4 * idx + 2] = hi * fz; spec[4 * idx + 3] = -hr * fz;
            htex[4 * idx] = hr; htex[4 * idx + 1] = hi; htex[4 * idx + 2] = hr; htex[4 * idx + 3] = hi; /* F/SpectrumHeight.shader:46 */
        }
    memcpy(dtex, spec, sizeof(double) * 4 * MM);
    stockham_2d(M, dtex, tmp); /* -> displacementTexture */
    stockham_2d(M, htex, tmp); /* -> heightTexture */
    /* F/OceanNormal.shader:39-56 */
    double ts = (double)p->length / (double)M;
    for (int py = 0; py < M; py++)
        for (int px = 0; px < M; px++) {
            size_t idx = (size_t)py * M + px;
            double cx = dtex[4 * idx], cy = dtex[4 * idx + 1], cz = dtex[4 * idx + 2]; /* center = D.rgb  (:44, the quirk) */
            int xr = clampi(px + 1, 0, M - 1), xl = clampi(px - 1, 0, M - 1), yt = clampi(py - 1, 0, M - 1), yb = clampi(py + 1, 0, M - 1);
#define GETVEC(X, Y, o) { size_t j_ = (size_t)(Y) * M + (X); o[0] = dtex[4 * j_]; o[1] = htex[4 * j_]; o[2] = dtex[4 * j_ + 2]; }
            double r[3], l[3], t[3], b[3];
            GETVEC(xr, py, r); GETVEC(xl, py, l); GETVEC(px, yt, t); GETVEC(px, yb, b);
            double right[3] = {ts + r[0] - cx, r[1] - cy, r[2] - cz};      /* :45 */
            double left[3] = {-ts + l[0] - cx, l[1] - cy, l[2] - cz};      /* :46 */
            double top[3] = {t[0] - cx, t[1] - cy, -ts + t[2] - cz};       /* :47 */
            double bot[3] = {b[0] - cx, b[1] - cy, ts + b[2] - cz};        /* :48 */
#define CROSS(a, b, o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
            double c1[3], c2[3], c3[3], c4[3];
            CROSS(right, top, c1); CROSS(top, left, c2); CROSS(left, bot, c3); CROSS(bot, right, c4); /* :50-53 */
            double nx = c1[0] + c2[0] + c3[0] + c4[0], ny = c1[1] + c2[1] + c3[1] + c4[1], nz = c1[2] + c2[2] + c3[2] + c4[2];
            double mag = sqrt(nx * nx + ny * ny + nz * nz);
            normal[3 * idx] = nx / mag; normal[3 * idx + 1] = ny / mag; normal[3 * idx + 2] = nz / mag; /* :55 */
            height[idx] = htex[4 * idx];
            disp[2 * idx] = dtex[4 * idx]; disp[2 * idx + 1] = dtex[4 * idx + 2];
            if (disp_g) disp_g[idx] = dtex[4 * idx + 1];
        }
    /* F/WhiteCap.shader:33-45: texelSize = 1/_Length with _Length = resolution (S/OceanRenderer.cs:306) = 8 texels */
    for (int py = 0; py < M; py++)
        for (int px = 0; px < M; px++) {
            size_t idx = (size_t)py * M + px;
            int ym = clampi(py - 8, 0, M - 1), yp = clampi(py + 8, 0, M - 1), xm = clampi(px - 8, 0, M - 1), xp = clampi(px + 8, 0, M - 1);
            size_t iym = (size_t)ym * M + px, iyp = (size_t)yp * M + px, ixm = (size_t)py * M + xm, ixp = (size_t)py * M + xp;
            double dDdy_x = -0.5 * (dtex[4 * iym] - dtex[4 * iyp]) / 8.0, dDdy_y = -0.5 * (dtex[4 * iym + 2] - dtex[4 * iyp + 2]) / 8.0; /* :36 */
            double dDdx_x = -0.5 * (dtex[4 * ixm] - dtex[4 * ixp]) / 8.0, dDdx_y = -0.5 * (dtex[4 * ixm + 2] - dtex[4 * ixp + 2]) / 8.0; /* :37 */
            double n0 = 0.3 * normal[3 * idx], n1 = 0.3 * normal[3 * idx + 2];  /* :38 */
            double jac = (1 + dDdx_x) * (1 + dDdy_y) - dDdx_y * dDdy_x;          /* :39 */
            double turb = fmax(0.0, 1 - jac + sqrt(n0 * n0 + n1 * n1));          /* :40 */
            double tt = turb < 0 ? 0 : (turb > 1 ? 1 : turb);
            white[idx] = tt * tt * (3.0 - 2.0 * tt);                             /* smoothstep(0,1,turb) :43 */
        }
    free(spec); free(tmp); free(dtex); free(htex);
}

/* numpy-friendly helper for the 1024^2 check: the per-frame spectra only (f64), so that oracle.py can put
 * np.fft.fft2 in place of the 40 literal passes.  spec_d [M*M*4] = (hx, hz), spec_h [M*M*2] = h. */
void orr_spectra_f64(const orr_params* p, const float* init4, float* phase, float delta_time, double* spec_d, double* spec_h) {
    int M = p->resolution * 8;
    float dt = delta_time * p->mult;
    for (int py = 0; py < M; py++)
        for (int px = 0; px < M; px++) {
            size_t idx = (size_t)py * M + px;
            phase[idx] = orr_phase_advance(p, M, px, py, phase[idx], dt);
            double ph = (double)phase[idx], c = cos(ph), s = sin(ph);
            float kxf, kzf;
            get_wave_f((float)px + 0.5f, (float)py + 0.5f, p->length, (float)M, &kxf, &kzf);
            double kx = kxf, kz = kzf, w = sqrt(kx * kx + kz * kz);
            double ax = init4[4 * idx], ay = init4[4 * idx + 1], bx = init4[4 * idx + 2], by = init4[4 * idx + 3];
            double hr = ax * c - ay * s + bx * c + by * s, hi = ax * s + ay * c - bx * s + by * c;
            if (w < 0.0001) w = 0.0001;
            double fx = kx / w * (double)p->choppiness, fz = kz / w * (double)p->choppiness;
            spec_d[4 * idx] = hi * fx; spec_d[4 * idx + 1] = -hr * fx; spec_d[4 * idx + 2] = hi * fz; spec_d[4 * idx + 3] = -hr * fz;
            spec_h[2 * idx] = hr; spec_h[2 * idx + 1] = hi;
        }
}

/* normal + whitecap passes on caller-supplied transformed textures (f64): dtex [M*M*4], hre [M*M] */
void orr_normal_white_f64(const orr_params* p, const double* dtex, const double* hre, double* normal, double* white) {
    int M = p->resolution * 8;
    double ts = (double)p->length / (double)M;
    for (int py = 0; py < M; py++)
        for (int px = 0; px < M; px++) {
            size_t idx = (size_t)py * M + px;
            double cx = dtex[4 * idx], cy = dtex[4 * idx + 1], cz = dtex[4 * idx + 2];
            int xr = clampi(px + 1, 0, M - 1), xl = clampi(px - 1, 0, M - 1), yt = clampi(py - 1, 0, M - 1), yb = clampi(py + 1, 0, M - 1);
#define GETV2(X, Y, o) { size_t j_ = (size_t)(Y) * M + (X); o[0] = dtex[4 * j_]; o[1] = hre[j_]; o[2] = dtex[4 * j_ + 2]; }
            double r[3], l[3], t[3], b[3];
            GETV2(xr, py, r); GETV2(xl, py, l); GETV2(px, yt, t); GETV2(px, yb, b);
            double right[3] = {ts + r[0] - cx, r[1] - cy, r[2] - cz}, left[3] = {-ts + l[0] - cx, l[1] - cy, l[2] - cz};
            double top[3] = {t[0] - cx, t[1] - cy, -ts + t[2] - cz}, bot[3] = {b[0] - cx, b[1] - cy, ts + b[2] - cz};
            double c1[3], c2[3], c3[3], c4[3];
            CROSS(right, top, c1); CROSS(top, left, c2); CROSS(left, bot, c3); CROSS(bot, right, c4);
            double nx = c1[0] + c2[0] + c3[0] + c4[0], ny = c1[1] + c2[1] + c3[1] + c4[1], nz = c1[2] + c2[2] + c3[2] + c4[2];
            double mag = sqrt(nx * nx + ny * ny + nz * nz);
            normal[3 * idx] = nx / mag; normal[3 * idx + 1] = ny / mag; normal[3 * idx + 2] = nz / mag;
        }
    for (int py = 0; py < M; py++)
        for (int px = 0; px < M; px++) {
            size_t idx = (size_t)py * M + px;
            int ym = clampi(py - 8, 0, M - 1), yp = clampi(py + 8, 0, M - 1), xm = clampi(px - 8, 0, M - 1), xp = clampi(px + 8, 0, M - 1);
            size_t iym = (size_t)ym * M + px, iyp = (size_t)yp * M + px, ixm = (size_t)py * M + xm, ixp = (size_t)py * M + xp;
            double dDdy_x = -0.5 * (dtex[4 * iym] - dtex[4 * iyp]) / 8.0, dDdy_y = -0.5 * (dtex[4 * iym + 2] - dtex[4 * iyp + 2]) / 8.0;
            double dDdx_x = -0.5 * (dtex[4 * ixm] - dtex[4 * ixp]) / 8.0, dDdx_y = -0.5 * (dtex[4 * ixm + 2] - dtex[4 * ixp + 2]) / 8.0;
            double n0 = 0.3 * normal[3 * idx], n1 = 0.3 * normal[3 * idx + 2];
            double jac = (1 + dDdx_x) * (1 + dDdy_y) - dDdx_y * dDdy_x;
            double turb = fmax(0.0, 1 - jac + sqrt(n0 * n0 + n1 * n1));
            double tt = turb < 0 ? 0 : (turb > 1 ? 1 : turb);
            white[idx] = tt * tt * (3.0 - 2.0 * tt);
        }
}

/* ---- the ocean material's vertex stage on the OceanRenderer mesh ------------------------------------------------
 * S/OceanRenderer.cs:172-207 builds a res x res grid (res = resolution/8 after :169) with
 *   vertex(i,j) = ((i - res/2) uw + (res even ? uw/2 : 0), 0, (j - res/2) uw + ...),  uv = (i/(res-1), j/(res-1)).
 * W/TestOcean.shader:61-79 (and W/MistralWaterCommon.cginc:19-25,54-56) then displace it from the four textures:
 *   v.y += _Height(uv).r / 8;  v.xz += _Anim(uv).rb / 8;  normal = normalize(_Bump(uv).rgb);  color = _White(uv).r
 * tex2Dlod at lod 0 on a bilinear, clamp-addressed texture [unity defaults]: texel centres at (p + .5)/M.
 * Inputs (double): height [M*M], disp_rb [M*M*2], normal [M*M*3], white [M*M], texel (px,py) at py*M + px.          */
static double bilinear(const double* tex, int M, int stride, int comp, double u, double v) {
    double x = u * M - 0.5, y = v * M - 0.5;
    double fx = floor(x), fy = floor(y), wx = x - fx, wy = y - fy;
    int x0 = clampi((int)fx, 0, M - 1), x1 = clampi((int)fx + 1, 0, M - 1);
    int y0 = clampi((int)fy, 0, M - 1), y1 = clampi((int)fy + 1, 0, M - 1);
    double a00 = tex[((size_t)y0 * M + x0) * stride + comp], a10 = tex[((size_t)y0 * M + x1) * stride + comp];
    double a01 = tex[((size_t)y1 * M + x0) * stride + comp], a11 = tex[((size_t)y1 * M + x1) * stride + comp];
    double a0 = a00 + (a10 - a00) * wx, a1 = a01 + (a11 - a01) * wx;
    return a0 + (a1 - a0) * wy;
}
void orr_mesh_vertex_stage_f64(int M, int res, float unit_width, const double* height, const double* disp_rb,
                               const double* normal, const double* white, double* out_v, double* out_n, double* out_c) {
    int half = res / 2;
    for (int i = 0; i < res; i++)
        for (int j = 0; j < res; j++) {
            size_t cur = (size_t)i * res + j;
            /* uv in float32 as the C# stores it (Vector2 of floats, :184) */
            float uf = (float)i * 1.0f / (float)(res - 1), vf = (float)j * 1.0f / (float)(res - 1);
            double u = uf, v = vf;
            float hx = (float)(i - half) * unit_width + (res % 2 == 0 ? unit_width / 2.0f : 0.0f); /* :178-183 */
            float hz = (float)(j - half) * unit_width + (res % 2 == 0 ? unit_width / 2.0f : 0.0f);
            out_v[3 * cur] = (double)hx + bilinear(disp_rb, M, 2, 0, u, v) / 8.0;
            out_v[3 * cur + 1] = bilinear(height, M, 1, 0, u, v) / 8.0;
            out_v[3 * cur + 2] = (double)hz + bilinear(disp_rb, M, 2, 1, u, v) / 8.0;
            double n[3];
            for (int k = 0; k < 3; k++) n[k] = bilinear(normal, M, 3, k, u, v);
            double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            out_n[3 * cur] = n[0] / len; out_n[3 * cur + 1] = n[1] / len; out_n[3 * cur + 2] = n[2] / len;
            out_c[cur] = bilinear(white, M, 1, 0, u, v);
        }
}
