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
        std::printf("FFTMesh 256^2: timer = %.4f, max|height| = %.4f, colour[0] = %.4f\n", m.timer(), hmax, m.mesh.colors[0].r);
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
