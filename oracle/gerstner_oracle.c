/*
 * oracle/gerstner_oracle.c  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the pond's Gerstner vertex displacement (BASELINE config 5):
 *   W/MistralWaterLib.cginc:71-99  Gerstner()  as called from  :154-180 Displacement()  (`half` = f32 on desktop).
 * The reference sums exactly 4 waves (directionAB.xy/.zw, directionCD.xy/.zw, speed.xyzw); nwaves generalises that
 * (two parameter sets = the "8 waves" of BASELINE config 5).  PARITY UNPINNED (no vectors in the reference).
 */
#include <math.h>
#include <stdint.h>

/* waves: nwaves x {dir.x, dir.y, speed}; amplitude is _Amplitude * 0.01 (:172) */
void orc_gerstner_f64(const float* pos_xyz, int64_t nverts, const float* waves, int nwaves, float amplitude, float frequency,
                      float steepness, float t, double* out_xyz) {
    for (int64_t v = 0; v < nverts; v++) {
        double x = pos_xyz[3 * v], y = pos_xyz[3 * v + 1], z = pos_xyz[3 * v + 2]; /* sVertex.xz = world x,z (:170) */
        double ox = 0, oy = 0, oz = 0;
        for (int i = 0; i < nwaves; i++) {
            double dx = waves[3 * i], dy = waves[3 * i + 1], sp = waves[3 * i + 2];
            double th = (double)frequency * (dx * x + dy * z) + (double)t * sp;  /* :80-84 */
            ox += cos(th) * (double)steepness * (double)amplitude * dx;          /* :77,86 */
            oz += cos(th) * (double)steepness * (double)amplitude * dy;          /* :78,87 */
            oy += sin(th);                                                       /* :88 */
        }
        out_xyz[3 * v] = x + ox;                       /* v.vertex.xyz += offsets (:176) */
        out_xyz[3 * v + 1] = y + (double)amplitude * oy;
        out_xyz[3 * v + 2] = z + oz;
    }
}

/* The same in the shader's own arithmetic (float32; sinf/cosf): the "CPU_GERSTNER" baseline of BASELINE.md section 4 --
 * what a CPU port of W/MistralWaterLib.cginc:71-99 costs per vertex.  Vertices [v0, v1) so that callers can split a lattice
 * across host threads. */
void orc_gerstner_f32_range(const float* pos_xyz, int64_t v0, int64_t v1, const float* waves, int nwaves, float amplitude,
                            float frequency, float steepness, float t, float* out_xyz) {
    for (int64_t v = v0; v < v1; v++) {
        float x = pos_xyz[3 * v], y = pos_xyz[3 * v + 1], z = pos_xyz[3 * v + 2];
        float ox = 0.f, oy = 0.f, oz = 0.f;
        for (int i = 0; i < nwaves; i++) {
            float dx = waves[3 * i], dy = waves[3 * i + 1], sp = waves[3 * i + 2];
            float th = frequency * (dx * x + dy * z) + t * sp;
            float c = cosf(th), s = sinf(th);
            ox += c * steepness * amplitude * dx;
            oz += c * steepness * amplitude * dy;
            oy += s;
        }
        out_xyz[3 * v] = x + ox;
        out_xyz[3 * v + 1] = y + amplitude * oy;
        out_xyz[3 * v + 2] = z + oz;
    }
}
