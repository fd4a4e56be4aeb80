/*
 * oracle/pond_oracle.c  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * CPU restatement, in double precision, of the pond material's vertex-stage Displacement()
 *   W/MistralWaterLib.cginc:154-180  with the three displacement functions of the library:
 *   Wave() :127-152, Gerstner() :71-99, GerstnerLevelOne() :101-125.
 * unity_ObjectToWorld / unity_WorldToObject are the identity (object space == world space); _Time.y = t.
 * PARITY UNPINNED: the reference holds no vectors for these functions; the restatement follows the shader text.
 */
#include <math.h>
#include <stdint.h>

typedef struct {
    int32_t mode; /* 0 Wave, 1 Gerstner, 2 GerstnerLevelOne */
    float amplitude, frequency, speed, steepness, smoothing;
    float wspeed[4], dir_ab[4], dir_cd[4];
} orc_pond_params;

static void wave(const orc_pond_params* P, double t, double x, double y, double z, double* o, double* n) {
    double v0[3] = {x, y, z}, v1[3] = {x + 0.05, y, z}, v2[3] = {x, y, z + 0.05}; /* :129-131 */
    double speed = (double)P->speed * t;                                            /* :133 */
    double amplitude = (double)P->amplitude * 0.01;                                 /* :134 */
    double f = P->frequency;
    v0[1] += sin(speed + v0[0] * f) * amplitude; /* :136-138 */
    v1[1] += sin(speed + v1[0] * f) * amplitude;
    v2[1] += sin(speed + v2[0] * f) * amplitude;
    v0[1] -= cos(speed + v0[2] * f) * amplitude; /* :140-142 */
    v1[1] -= cos(speed + v1[2] * f) * amplitude;
    v2[1] -= cos(speed + v2[2] * f) * amplitude;
    v1[1] -= (v1[1] - v0[1]) * (1.0 - (double)P->smoothing); /* :144-145 */
    v2[1] -= (v2[1] - v0[1]) * (1.0 - (double)P->smoothing);
    double a[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]}, b[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; /* :147 */
    double len = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    n[0] = c[0] / len; n[1] = c[1] / len; n[2] = c[2] / len; /* :150 */
    o[0] = x; o[1] = y + v0[1]; o[2] = z;                    /* offsets = v0 (:151); v.vertex.y += offsets.y (:163) */
}

static void gerstner(const orc_pond_params* P, double t, double x, double y, double z, double* o, double* n) {
    double amplitude = (double)P->amplitude * 0.01; /* :172 */
    double d[4][2] = {{P->dir_ab[0], P->dir_ab[1]}, {P->dir_ab[2], P->dir_ab[3]}, {P->dir_cd[0], P->dir_cd[1]}, {P->dir_cd[2], P->dir_cd[3]}};
    double ox = 0, oy = 0, oz = 0;
    for (int i = 0; i < 4; i++) {
        double th = (double)P->frequency * (d[i][0] * x + d[i][1] * z) + t * (double)P->wspeed[i]; /* :80-81 */
        ox += cos(th) * (double)P->steepness * amplitude * d[i][0];                                 /* :77-78,86 */
        oz += cos(th) * (double)P->steepness * amplitude * d[i][1];                                 /* :87 */
        oy += sin(th) * amplitude;                                                                  /* :88 */
    }
    o[0] = x + ox; o[1] = y + oy; o[2] = z + oz; /* :176 */
    n[0] = 0; n[1] = 1; n[2] = 0;                /* :98 */
}

static void level_one(const orc_pond_params* P, double t, double x, double y, double z, double* o, double* n) {
    static const float amps[5] = {0.7f, 0.6f, 0.6f, 0.7f, 0.9f};                 /* :105-109 (float literals) */
    static const float steeps[5] = {0.95f, 0.615f, 0.821f, 0.462f, 0.611f};
    static const float speeds[5] = {-2.112f, 0.6124f, -0.878f, -3.6234f, 1.f};
    static const float dir[5][2] = {{1.f, -0.2f}, {-0.9f, 1.f}, {0.2f, 0.2f}, {-1.0f, 0.77f}, {0.99f, -1.145f}};
    static const float fs[5] = {0.954f, 1.52f, 0.44f, 0.21f, 0.8f};
    double ox = 0, oy = 0, oz = 0, A = P->amplitude, F = P->frequency, S = P->steepness;
    for (int i = 0; i < 5; i++) { /* :112-117 */
        double th = F * fs[i] * (x * dir[i][0] + z * dir[i][1]) + (double)speeds[i] * F * fs[i] * t;
        ox += S * A * steeps[i] * amps[i] * dir[i][0] * cos(th);
        oz += S * A * steeps[i] * amps[i] * dir[i][1] * cos(th);
        oy += A * amps[i] * sin(th);
    }
    o[0] = x + ox; o[1] = y + oy; o[2] = z + oz;
    n[0] = 0; n[1] = 1; n[2] = 0; /* :121 */
}

/* returns 0, or 1 for an unknown mode */
int orc_pond_displace_f64(const orc_pond_params* P, const float* pos_xyz, int64_t nverts, float t, double* out_xyz,
                          double* out_normal_xyz) {
    for (int64_t v = 0; v < nverts; v++) {
        double x = pos_xyz[3 * v], y = pos_xyz[3 * v + 1], z = pos_xyz[3 * v + 2];
        double* o = out_xyz + 3 * v;
        double* n = out_normal_xyz + 3 * v;
        if (P->mode == 0) wave(P, t, x, y, z, o, n);
        else if (P->mode == 1) gerstner(P, t, x, y, z, o, n);
        else if (P->mode == 2) level_one(P, t, x, y, z, o, n);
        else return 1;
    }
    return 0;
}
