// host_demo.cpp -- drives the C++ FFTMesh mirror for a few frames (needs an MI355X at run time).
#include <cstdio>
#include "FFTMesh.hpp"

int main() {
    using namespace mistral_water;
    try {
        FFTMesh m;
        m.resolution = 256; m.unitWidth = 1.f; m.length = 256.f; m.wind = {14.45f, 12.f}; m.amplitude = 2.4e-7f;
        m.choppiness = 0.46f;
        m.Awake();
        for (int f = 0; f < 3; f++) m.Update(1.f / 60.f);
        double hmax = 0;
        for (auto& v : m.mesh.vertices) hmax = hmax > (v.y < 0 ? -v.y : v.y) ? hmax : (v.y < 0 ? -v.y : v.y);
        std::printf("FFTMesh 256^2: timer = %.9g, max|height| = %.9g, colour[0] = %.9g\n", m.timer(), hmax, m.mesh.colors[0].r);
        OceanRenderer r;
        r.resolution = 16; r.length = 60.f; r.amplitude = 0.41f; r.choppiness = 0.46f; r.mult = 1.5f; r.wind = {14.45f, 12.f};
        r.Awake();
        r.Update(1.f / 60.f);
        std::vector<float> H, A, B, W;
        r.GenerateTextureRGBA(1.f / 60.f, H, A, B, W);
        std::vector<Vector3> dv, dn;
        std::vector<float> foam;
        r.DisplaceMesh(dv, dn, foam);
        std::printf("OceanRenderer 128^2: height.r[0] = %.5f, bump.a[0] = %.1f, vertex[0].y = %.5f\n", H[0], B[3], dv[0].y);
        PondMaterial pond;
        std::vector<Vector3> grid(1000), moved, nrm;
        for (int k = 0; k < 1000; k++) grid[k] = {0.1f * (k % 40), 0.f, 0.1f * (k / 40)};
        pond.Displacement(grid, 2.f, moved, &nrm);
        std::printf("pond Gerstner: vertex[7] = (%.4f, %.4f, %.4f)\n", moved[7].x, moved[7].y, moved[7].z);
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
