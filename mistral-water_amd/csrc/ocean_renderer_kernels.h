// ocean_renderer_kernels.h -- MW_SEM_OCEANRENDERER (S/OceanRenderer.cs + F/*.shader).  PLACEHOLDER: filled in below.
#pragma once
#include <string>
#include "fftmesh_kernels.h"
#include "../../include/mistral_water.h"

namespace mw {
struct OrState {
    float choppiness = 0.f;
    float *out_height = nullptr, *out_disp = nullptr, *out_normal = nullptr, *out_white = nullptr;
};
static inline const char* or_last_error() { return "OceanRenderer semantics not implemented yet"; }
static inline mw_status or_create(OrState&, const mw_params&, int, hipStream_t) { return MW_EINVAL; }
static inline mw_status or_generate(OrState&, float, float*, float*, float*, float*, hipStream_t) { return MW_EINVAL; }
static inline void or_free(OrState&) {}
}  // namespace mw
