// mw_switches.h -- the library's run-time plan switches (host side).
// Every alternative plan below produces the same bits as the default one (tests/test_zz_frame_plan.py, tests/test_state_and_tiles.py hold
// them to that; MW_OR_PACKED = 0 the same textures to float32 rounding, tests/test_ocean_renderer.py) and exists for A/B measurements and for those tests.  A product build never reads the environment: the switches carry
// their defaults and change only through the test hook mw_debug_set_switch (include/mistral_water_hooks.h).  A lab build (-DMW_LAB: what
// tools/build_variant.sh and build_native(extra=...) make; mw_build_id() then says "lab") also takes them from environment variables of
// the same names, read once when the library is loaded.
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>

namespace mw {

enum Switch {
    SW_LATENCY_PLAN,      // 1: single-step enqueues at 256^2 .. 1024^2 through the frame plan's launches; 0: the batched plan's kernels
    SW_FRAME_KERNEL,      // 1: pass 2 of a single step by k_pass2_frame; 0: round 3's sequential-halo kernel
    SW_P1_FRAME_XCD,      // 1: pass 1 of a single step on the XCD-aware grid; 0: the plain (column jobs, 3) grid
    SW_P1_TGROUP,         // time-steps of one pass-1 column job kept on one XCD; -1: the built-in rule
    SW_CZT_ONE,           // 1: chirp-z grids with N <= 20 in one launch; 0: two
    SW_CZT_FUSED,         // 1: chirp-z grids with N <= 128 in two launches; 0: three
    SW_DIRECT_CZT,        // 1: non-FFT grids by chirp-z (N <= 2048); 0: the MFMA GEMM form (read when a handle is CREATED)
    SW_TILES_FORCE_RCCL,  // 1: mw_tiles_gather sends every tile through ncclSend / ncclRecv, the root's own included
    SW_POND_STEPS_PER_WG, // time values per workgroup of k_gerstner_steps; 0: the built-in 8
    SW_POND_XCD,          // 1: the step groups of a vertex chunk on one XCD (measured no faster)
    SW_OR_PACKED,         // 1: OceanRenderer planar-texture calls with a symmetric phase run two transforms per frame; 0: always three
    SW_COUNT
};
struct SwitchDef { const char* name; int def; };
static const SwitchDef g_switch_defs[SW_COUNT] = {
    {"MW_LATENCY_PLAN", 1}, {"MW_FRAME_KERNEL", 1}, {"MW_P1_FRAME_XCD", 1}, {"MW_P1_TGROUP", -1}, {"MW_CZT_ONE", 1}, {"MW_CZT_FUSED", 1},
    {"MW_DIRECT_CZT", 1}, {"MW_TILES_FORCE_RCCL", 0}, {"MW_POND_STEPS_PER_WG", 0}, {"MW_POND_XCD", 0}, {"MW_OR_PACKED", 1},
};
struct SwitchTable {
    std::atomic<int> v[SW_COUNT];
    SwitchTable() {
        for (int k = 0; k < SW_COUNT; k++) {
            int x = g_switch_defs[k].def;
#ifdef MW_LAB
            if (const char* e = std::getenv(g_switch_defs[k].name)) x = std::atoi(e);
#endif
            v[k].store(x);
        }
    }
};
inline SwitchTable& switch_table() { static SwitchTable t; return t; }
inline int sw(Switch k) { return switch_table().v[k].load(std::memory_order_relaxed); }
// name -> index, -1 when unknown
inline int switch_index(const char* name) {
    for (int k = 0; k < SW_COUNT; k++)
        if (name && std::strcmp(name, g_switch_defs[k].name) == 0) return k;
    return -1;
}

}  // namespace mw
