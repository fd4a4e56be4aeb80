// FFTMesh.hpp / OceanRenderer -- C++ mirror of the reference's two MonoBehaviours over the C ABI.
//
// The reference's host language is C# (Unity); the build image has no C# toolchain, so the compiled host side
// above include/mistral_water.h is C++ (and Python for the tests).  Public field names and the Awake()/Update()
// lifecycle follow S/FFTMesh.cs:9-23,60-84 and S/OceanRenderer.cs:10-19,76-110; everything the reference
// computes in private methods is delegated to libmistral_water.so (HIP, MI355X).  Header-only.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mistral_water.h"

namespace mistral_water {

struct Vector2 { float x = 0.f, y = 0.f; };
struct Vector3 { float x = 0.f, y = 0.f, z = 0.f; };
struct Color { float r = 0.f, g = 0.f, b = 0.f, a = 0.f; };
static_assert(sizeof(Vector3) == 12 && sizeof(Color) == 16 && sizeof(Vector2) == 8, "blittable Unity structs");

// the UnityEngine.Mesh members the reference assigns (S/FFTMesh.cs:134-138,277-279)
struct Mesh {
    std::vector<Vector3> vertices, normals;
    std::vector<Vector2> uv;
    std::vector<Color> colors;
    std::vector<int32_t> indices;
};

inline void check(mw_status s) {
    if (s != MW_OK) throw std::runtime_error(std::string("mistral_water: ") + mw_last_error());
}

class FFTMesh {
public:
    // ---- public Inspector fields, S/FFTMesh.cs:9-23 -------------------------------------------------
    float choppiness = 1.f;
    float tDivision = 1.f;
    int resolution = 50;
    float unitWidth = 1.f;
    bool generate = false;
    float length = 1.f;
    Vector2 wind{1.f, 1.f};
    float amplitude = 1.f;
    // ---- additions: the constants the reference hard-codes, and what Unity supplies implicitly --------
    float gravity = 9.81f;  // G, S/FFTMesh.cs:52
    uint64_t seed = 1;      // the reference never seeds UnityEngine.Random
    bool fixedSeed = false; // true: every regeneration reproduces the same sea (tests); false: a new one, like GenerateMesh
    int device = 0;
    Mesh mesh;

    ~FFTMesh() { mw_ocean_destroy(ocean_); }

    void Awake() {  // S/FFTMesh.cs:75-84
        SetParamsAndGenerateMesh();
    }
    void Update(float deltaTime) {  // S/FFTMesh.cs:60-73
        if (generate) {
            timer_ = 0.f;
            SetParamsAndGenerateMesh();
            generate = false;
        }
        timer_ += deltaTime / tDivision;
        EvaluateWaves(timer_);
    }
    void EvaluateWaves(float t) {  // S/FFTMesh.cs:224-280
        check(mw_ocean_set_choppiness(ocean_, choppiness));
        check(mw_ocean_evaluate(ocean_, t, &mesh.vertices[0].x, &mesh.normals[0].x, &mesh.colors[0].r));
    }
    float timer() const { return timer_; }
    mw_ocean* handle() { return ocean_; }

private:
    void SetParamsAndGenerateMesh() {  // S/FFTMesh.cs:90-139
        mw_ocean_destroy(ocean_);
        ocean_ = nullptr;
        mw_params p;
        mw_params_default(&p, MW_SEM_FFTMESH);
        p.resolution = resolution; p.unit_width = unitWidth; p.length = length; p.wind_x = wind.x; p.wind_y = wind.y;
        p.amplitude = amplitude; p.choppiness = choppiness; p.gravity = gravity; p.t_division = tDivision;
        // GenerateMesh draws fresh UnityEngine.Random values every time it runs (S/FFTMesh.cs:114-116): each regeneration
        // is a NEW sea state unless fixedSeed asks for reproducibility
        p.seed = fixedSeed ? seed : seed + generation_;
        generation_++;
        p.device = device;
        check(mw_ocean_create(&p, &ocean_));
        const size_t nn = (size_t)resolution * resolution;
        mesh.vertices.resize(nn); mesh.normals.resize(nn); mesh.uv.resize(nn); mesh.colors.resize(nn);
        mesh.indices.resize((size_t)mw_ocean_index_count(ocean_));
        check(mw_ocean_rest_mesh(ocean_, &mesh.vertices[0].x, &mesh.normals[0].x, &mesh.uv[0].x, mesh.indices.data()));
    }
    mw_ocean* ocean_ = nullptr;
    float timer_ = 0.f;
    uint64_t generation_ = 0;
};

class OceanRenderer {
public:
    // ---- public Inspector fields, S/OceanRenderer.cs:10-19 -------------------------------------------
    float mult = 2.f;
    float unitWidth = 1.f;
    int resolution = 256;
    float length = 256.f;
    float choppiness = 1.5f;
    float amplitude = 1.f;
    Vector2 wind;
    float gravity = 9.81f;
    uint64_t seed = 1;
    int device = 0;
    Mesh mesh;
    // the four result textures bound to the ocean material (S/OceanRenderer.cs:310-313), M = 8*resolution
    std::vector<float> heightTexture, displacementTexture, normalTexture, whiteTexture;

    ~OceanRenderer() { mw_ocean_destroy(ocean_); }

    void Awake() {  // S/OceanRenderer.cs:76-89: SetParams, GenerateMesh, RenderInitial
        Create();
        const size_t nn = (size_t)resolution * resolution;
        mesh.vertices.resize(nn); mesh.normals.resize(nn); mesh.uv.resize(nn);
        mesh.indices.resize((size_t)mw_ocean_index_count(ocean_));
        check(mw_ocean_rest_mesh(ocean_, &mesh.vertices[0].x, &mesh.normals[0].x, &mesh.uv[0].x, mesh.indices.data()));
    }
    void Update(float deltaTime) {  // S/OceanRenderer.cs:91-110
        GenerateTexture(deltaTime);                            // with the values the materials carried into this frame (:93)
        check(mw_ocean_set_choppiness(ocean_, choppiness));    // spectrumMat._Choppiness for the NEXT frame (:96)
        if (oldLength_ != length || oldWind_.x != wind.x || oldWind_.y != wind.y || oldAmplitude_ != amplitude) {
            // RenderInitial again with the same seeds (:98-109): the phase textures keep running
            check(mw_ocean_reinit_spectrum(ocean_, length, wind.x, wind.y, amplitude, seed));
            oldLength_ = length; oldWind_ = wind; oldAmplitude_ = amplitude;
        }
    }
    void GenerateTexture(float deltaTime) {  // S/OceanRenderer.cs:216-316
        const size_t mm = (size_t)M_ * M_;
        heightTexture.resize(mm); displacementTexture.resize(2 * mm); normalTexture.resize(3 * mm); whiteTexture.resize(mm);
        check(mw_ocean_generate_texture(ocean_, deltaTime, heightTexture.data(), displacementTexture.data(),
                                        normalTexture.data(), whiteTexture.data()));
    }
    // The same frame as the four ARGBFloat render targets in the shaders' channel layout (S/OceanRenderer.cs:143-146):
    // what oceanMat.SetTexture("_Height"/"_Anim"/"_Bump"/"_White") binds (:310-313).  4 floats per texel each.
    void GenerateTextureRGBA(float deltaTime, std::vector<float>& height, std::vector<float>& anim, std::vector<float>& bump,
                             std::vector<float>& white) {
        const size_t mm = (size_t)M_ * M_ * 4;
        height.resize(mm); anim.resize(mm); bump.resize(mm); white.resize(mm);
        check(mw_ocean_generate_texture_rgba(ocean_, deltaTime, height.data(), anim.data(), bump.data(), white.data()));
    }
    // The ocean material's vertex stage (W/TestOcean.shader:61-79) applied to `mesh` from the latest frame's textures:
    // displaced vertices, per-vertex normals and foam factor, for a consumer that does not bind textures.
    void DisplaceMesh(std::vector<Vector3>& vertices, std::vector<Vector3>& normals, std::vector<float>& foam) {
        const size_t nn = (size_t)resolution * resolution;
        vertices.resize(nn); normals.resize(nn); foam.resize(nn);
        check(mw_ocean_displace_mesh(ocean_, &vertices[0].x, &normals[0].x, foam.data()));
    }

private:
    void Create() {
        mw_ocean_destroy(ocean_);
        ocean_ = nullptr;
        mw_params p;
        mw_params_default(&p, MW_SEM_OCEANRENDERER);
        p.resolution = resolution; p.unit_width = unitWidth; p.length = length; p.wind_x = wind.x; p.wind_y = wind.y;
        p.amplitude = amplitude; p.choppiness = choppiness; p.gravity = gravity; p.mult = mult; p.seed = seed;
        p.device = device;
        check(mw_ocean_create(&p, &ocean_));
        M_ = mw_ocean_grid_size(ocean_);
        oldLength_ = length; oldWind_ = wind; oldAmplitude_ = amplitude;
    }
    mw_ocean* ocean_ = nullptr;
    int M_ = 0;
    float oldLength_ = 0.f, oldAmplitude_ = 0.f;
    Vector2 oldWind_;
};

// The pond material's displacement properties (W/MistralWaterLib.cginc:53-66) and its vertex-stage Displacement()
// (:154-180) in the three modes of the shader library.
class PondMaterial {
public:
    int mode = MW_POND_GERSTNER;   // _DISPLACEMENTMODE_*: MW_POND_WAVE / MW_POND_GERSTNER / MW_POND_GERSTNER_LEVEL_ONE
    float _Amplitude = 10.f, _Frequency = 2.58f, _Speed = 0.f, _Steepness = 0.99f, _Smoothing = 1.f;  // M/Pond Water Mat.mat
    float _WSpeed[4] = {1.2f, 0.71f, 1.1f, 0.73f};
    float _WDirectionAB[4] = {0.3f, 0.73f, 0.85f, 0.25f};
    float _WDirectionCD[4] = {-0.25f, 1.11f, 0.5f, 0.5f};
    int device = 0;

    // time = _Time.y; normals may be null
    void Displacement(const std::vector<Vector3>& in, float time, std::vector<Vector3>& out, std::vector<Vector3>* normals) const {
        mw_pond_params p;
        p.mode = mode; p.amplitude = _Amplitude; p.frequency = _Frequency; p.speed = _Speed; p.steepness = _Steepness;
        p.smoothing = _Smoothing;
        for (int i = 0; i < 4; i++) { p.wspeed[i] = _WSpeed[i]; p.dir_ab[i] = _WDirectionAB[i]; p.dir_cd[i] = _WDirectionCD[i]; }
        out.resize(in.size());
        if (normals) normals->resize(in.size());
        if (in.empty()) return;
        check(mw_pond_displace(&p, &in[0].x, (int64_t)in.size(), time, &out[0].x, normals ? &(*normals)[0].x : nullptr, device));
    }
};

}  // namespace mistral_water
