"""ctypes binding of libmistral_water.so (the C ABI declared in include/mistral_water.h).

The library is HIP-only.  Loading fails loudly when the shared object has not been built, and
``mw_ocean_create`` fails with MW_EDEVICE when no MI355X is visible -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
AMD_DIR = os.path.dirname(PKG_DIR)                      # .../mistral-water_amd
REPO_DIR = os.path.dirname(AMD_DIR)
CSRC_DIR = os.path.join(AMD_DIR, "csrc")
LIB_PATH = os.environ.get("MW_LIB") or os.path.join(AMD_DIR, "libmistral_water.so")  # MW_LIB: A/B kernel variants
HEADER_PATH = os.path.join(REPO_DIR, "include", "mistral_water.h")
HOOKS_HEADER_PATH = os.path.join(REPO_DIR, "include", "mistral_water_hooks.h")

MW_OK, MW_EINVAL, MW_ENOTPOW2, MW_ENOTCOMMENSURATE, MW_ENOMEM, MW_EDEVICE, MW_ESTATE = range(7)
MW_SEM_FFTMESH, MW_SEM_OCEANRENDERER = 0, 1
MW_OUT_WHITE_SCALAR, MW_OUT_COLOR_RGBA = 0, 1
STATUS_NAMES = {0: "MW_OK", 1: "MW_EINVAL", 2: "MW_ENOTPOW2", 3: "MW_ENOTCOMMENSURATE", 4: "MW_ENOMEM",
                5: "MW_EDEVICE", 6: "MW_ESTATE"}


class MwParams(C.Structure):
    _fields_ = [("resolution", C.c_int32), ("unit_width", C.c_float), ("length", C.c_float), ("wind_x", C.c_float),
                ("wind_y", C.c_float), ("amplitude", C.c_float), ("choppiness", C.c_float), ("gravity", C.c_float),
                ("t_division", C.c_float), ("mult", C.c_float), ("seed", C.c_uint64), ("semantics", C.c_int32),
                ("device", C.c_int32)]


MW_POND_WAVE, MW_POND_GERSTNER, MW_POND_GERSTNER_LEVEL_ONE = 0, 1, 2
MW_COMM_ID_BYTES = 128


class MwPondParams(C.Structure):
    """mw_pond_params (include/mistral_water.h): the pond material's displacement properties."""
    _fields_ = [("mode", C.c_int32), ("amplitude", C.c_float), ("frequency", C.c_float), ("speed", C.c_float),
                ("steepness", C.c_float), ("smoothing", C.c_float), ("wspeed", C.c_float * 4), ("dir_ab", C.c_float * 4),
                ("dir_cd", C.c_float * 4)]


class MistralWaterError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


BUILD_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-Wno-unused-value",
               # the SLP vectoriser's v_pk_* packing costs ~300 v_mov per kernel and 50 % more VGPRs
               "-fno-slp-vectorize",
               # the 2-virtual-thread kernels are ~20k IR instructions once their field loop is unrolled: above the default cap
               # (16384) the pragma is ignored, the field index stays dynamic and the state arrays land in scratch
               "-mllvm", "-pragma-unroll-threshold=1000000",
               # no pairing of DS operations into ds_read2/ds_write2: the exchange layouts (XLay, mw_math.h) are bank-exact for
               # single 8-byte reads (256 B/clk); a paired read is served in 16-lane groups at 128 B/clk, 2-way conflicted
               "-Xclang", "-target-feature", "-Xclang", "-load-store-opt"]


def csrc_files():
    """The kernel sources (csrc/*.hip, *.h, *.inc), sorted: what the library is built from and what source_hash() covers."""
    return sorted(os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith((".hip", ".h", ".inc")))


def source_hash(extra=()) -> str:
    """What mw_build_id() reports (-DMW_BUILD_HASH): SHA-256 over the kernel sources, the boundary headers and the compile flags."""
    import hashlib
    h = hashlib.sha256()
    for path in csrc_files() + [HEADER_PATH, HOOKS_HEADER_PATH]:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update("\0".join(BUILD_FLAGS + list(extra)).encode())
    return h.hexdigest()[:16]


def built_hash(path: str) -> str | None:
    """The source hash embedded in a built library (the first word of mw_build_id()), read from the file without loading it:
    the id is a string constant "<16 hex digits> <tag>" in .rodata."""
    import re
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    m = re.search(rb"([0-9a-f]{16}) (?:lab:)?[A-Za-z0-9_.\-]+\x00", blob)
    return m.group(1).decode() if m else None


def build_native(force: bool = False, verbose: bool = False, out: str | None = None, extra=(), tag: str | None = None,
                 resource_report: str | None = None) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU).
    `out` / `extra` / `tag`: an A/B variant (tools/build_variant.sh): other output path, extra -D flags, the tag mw_build_id() carries."""
    target = out or LIB_PATH
    extra = list(extra)
    if extra and "-DMW_LAB" not in extra:
        extra.append("-DMW_LAB")      # any flag beyond BUILD_FLAGS makes a lab build: mw_build_id() says "lab:<tag>"
    if not force and not extra and os.path.exists(target) and built_hash(target) == source_hash():
        return target                 # the binary on disk IS these sources and flags (its embedded id says so), whatever the mtimes are
    import shutil
    if shutil.which("hipcc") is None:
        raise RuntimeError(f"{target} is missing or was built from other sources (embedded hash {built_hash(target)}, tree {source_hash()}) "
                           "and there is no hipcc here to rebuild it")
    cmd = ["hipcc"] + BUILD_FLAGS + extra + ['-DMW_BUILD_HASH="%s"' % source_hash(extra)]
    if tag:
        cmd.append('-DMW_BUILD_TAG="%s"' % tag)
    if resource_report:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    cmd += ["-o", target, os.path.join(CSRC_DIR, "mistral_water.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if resource_report:
        open(resource_report, "w").write(r.stderr)
    if verbose or r.returncode != 0:
        print(r.stdout, r.stderr if not resource_report else r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("hipcc failed building libmistral_water.so:\n" + r.stderr[-4000:])
    return target


_lib = None


def lib():
    """The loaded C-ABI library.  Raises (never falls back) when the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, f32p, i32p = C.c_void_p, C.c_void_p, C.c_void_p
    sig = {
        "mw_abi_version": (C.c_int32, []),
        "mw_build_id": (C.c_char_p, []),
        "mw_last_error": (C.c_char_p, []),
        "mw_device_count": (C.c_int32, []),
        "mw_params_default": (None, [C.POINTER(MwParams), C.c_int32]),
        "mw_ocean_create": (C.c_int, [C.POINTER(MwParams), C.POINTER(vp)]),
        "mw_ocean_destroy": (None, [vp]),
        "mw_ocean_create_batch": (C.c_int, [C.POINTER(MwParams), C.c_int32, C.POINTER(vp)]),
        "mw_ocean_batch_size": (C.c_int32, [vp]),
        "mw_ocean_set_stream": (C.c_int, [vp, vp]),
        "mw_ocean_use_own_stream": (C.c_int, [vp]),
        "mw_ocean_reinit_spectrum": (C.c_int, [vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint64]),
        "mw_ocean_get_phase": (C.c_int, [vp, f32p]),
        "mw_ocean_set_phase": (C.c_int, [vp, f32p]),
        "mw_ocean_set_timer": (C.c_int, [vp, C.c_float]),
        "mw_ocean_normal_length": (C.c_float, [vp]),
        "mw_ocean_set_normal_length": (C.c_int, [vp, C.c_float]),
        "mw_comm_unique_id": (C.c_int, [vp]),
        "mw_tiles_create": (C.c_int, [C.POINTER(MwParams), C.c_int32, i32p, C.c_int32, C.POINTER(vp)]),
        "mw_tiles_create_rank": (C.c_int, [C.POINTER(MwParams), C.c_int32, C.c_int32, vp, C.c_int32, C.c_int32, C.POINTER(vp)]),
        "mw_tiles_destroy": (None, [vp]),
        "mw_tiles_count": (C.c_int32, [vp]),
        "mw_tiles_local_count": (C.c_int32, [vp]),
        "mw_tiles_ocean": (vp, [vp, C.c_int32]),
        "mw_tiles_evaluate": (C.c_int, [vp, f32p, C.c_int32, C.c_uint32]),
        "mw_tiles_outputs": (C.c_int, [vp, C.c_int32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "mw_tiles_generate_texture": (C.c_int, [vp, C.c_float]),
        "mw_tiles_textures": (C.c_int, [vp, C.c_int32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "mw_tiles_generate_texture_steps": (C.c_int, [vp, f32p, C.c_int32]),
        "mw_tiles_frames": (C.c_int, [vp, C.c_int32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "mw_tiles_gather": (C.c_int, [vp, C.c_int32, C.c_int32]),
        "mw_tiles_gathered": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_int64)]),
        "mw_tiles_synchronize": (C.c_int, [vp]),
        "mw_ocean_get_stream": (vp, [vp]),
        "mw_ocean_synchronize": (C.c_int, [vp]),
        "mw_ocean_set_choppiness": (C.c_int, [vp, C.c_float]),
        "mw_ocean_set_spectrum": (C.c_int, [vp, f32p, f32p]),
        "mw_ocean_get_spectrum": (C.c_int, [vp, f32p, f32p]),
        "mw_ocean_rest_mesh": (C.c_int, [vp, f32p, f32p, f32p, i32p]),
        "mw_ocean_index_count": (C.c_int64, [vp]),
        "mw_ocean_grid_size": (C.c_int32, [vp]),
        "mw_ocean_evaluate": (C.c_int, [vp, C.c_float, f32p, f32p, f32p]),
        "mw_ocean_update": (C.c_int, [vp, C.c_float, f32p, f32p, f32p]),
        "mw_ocean_timer": (C.c_float, [vp]),
        "mw_ocean_reset_timer": (C.c_int, [vp]),
        "mw_ocean_evaluate_device": (C.c_int, [vp, f32p, C.c_int32, vp, vp, vp, C.c_uint32]),
        "mw_ocean_max_batch": (C.c_int32, [vp]),
        "mw_ocean_generate_texture": (C.c_int, [vp, C.c_float, f32p, f32p, f32p, f32p]),
        "mw_ocean_generate_texture_device": (C.c_int, [vp, C.c_float, vp, vp, vp, vp]),
        "mw_ocean_generate_texture_steps_device": (C.c_int, [vp, f32p, C.c_int32, vp, vp, vp, vp]),
        "mw_ocean_generate_texture_steps_rgba_device": (C.c_int, [vp, f32p, C.c_int32, vp, vp, vp, vp]),
        "mw_ocean_generate_texture_steps": (C.c_int, [vp, f32p, C.c_int32, f32p, f32p, f32p, f32p]),
        "mw_ocean_generate_texture_steps_rgba": (C.c_int, [vp, f32p, C.c_int32, f32p, f32p, f32p, f32p]),
        "mw_ocean_max_frames": (C.c_int32, [vp]),
        "mw_ocean_advance_phase": (C.c_int, [vp, f32p, C.c_int32]),
        "mw_ocean_frame_textures": (C.c_int, [vp, C.c_int32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "mw_host_register": (C.c_int, [vp, C.c_size_t]),
        "mw_host_unregister": (C.c_int, [vp]),
        "mw_ocean_generate_texture_rgba": (C.c_int, [vp, C.c_float, f32p, f32p, f32p, f32p]),
        "mw_ocean_generate_texture_rgba_device": (C.c_int, [vp, C.c_float, vp, vp, vp, vp]),
        "mw_ocean_displace_mesh": (C.c_int, [vp, f32p, f32p, f32p]),
        "mw_ocean_displace_mesh_device": (C.c_int, [vp, vp, vp, vp]),
        "mw_ocean_profile_kernels": (C.c_int, [vp, C.c_int32, C.c_int32, f32p, C.POINTER(C.c_char_p), C.POINTER(C.c_int32)]),
        "mw_ocean_profile_kernels_stats": (C.c_int, [vp, C.c_int32, C.c_int32, f32p, C.POINTER(C.c_char_p), C.POINTER(C.c_int32)]),
        "mw_gerstner_displace": (C.c_int, [f32p, C.c_int64, f32p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                           f32p, C.c_int32]),
        "mw_gerstner_displace_device": (C.c_int, [vp, C.c_int64, f32p, C.c_int32, C.c_float, C.c_float, C.c_float,
                                                  C.c_float, vp, vp]),
        "mw_gerstner_displace_steps_device": (C.c_int, [vp, C.c_int64, f32p, C.c_int32, C.c_float, C.c_float, C.c_float, f32p,
                                                        C.c_int32, vp, vp]),
        "mw_gerstner_max_steps": (C.c_int32, [C.c_int32]),
        "mw_pond_displace": (C.c_int, [C.POINTER(MwPondParams), f32p, C.c_int64, C.c_float, f32p, f32p, C.c_int32]),
        "mw_pond_displace_device": (C.c_int, [C.POINTER(MwPondParams), vp, C.c_int64, C.c_float, vp, vp, vp]),
        "mw_debug_pass1_time_group": (C.c_int32, [vp, C.c_int32]),
        "mw_debug_omega_t": (C.c_int, [vp, C.c_float, f32p]),
        "mw_debug_evaluate_hds": (C.c_int, [vp, C.c_float, f32p, f32p, f32p, f32p]),
        "mw_debug_get_omega": (C.c_int, [vp, f32p]),
        "mw_debug_sincos": (C.c_int, [f32p, C.c_int32, f32p, f32p]),
        "mw_debug_sincos_fast": (C.c_int, [f32p, C.c_int32, f32p, f32p]),
        "mw_debug_stream_read": (C.c_int, [C.c_int64, C.c_int32, C.c_int32]),
        "mw_debug_wave_transpose4": (C.c_int, [f32p]),
        "mw_debug_set_switch": (C.c_int, [C.c_char_p, C.c_int32]),
        "mw_debug_get_switch": (C.c_int32, [C.c_char_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError here = ABI symbol missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


#: every symbol include/mistral_water.h declares (checked by tests/test_abi.py against the header text)
ABI_SYMBOLS = [
    "mw_abi_version", "mw_build_id", "mw_last_error", "mw_device_count", "mw_params_default", "mw_ocean_create", "mw_ocean_destroy",
    "mw_ocean_create_batch", "mw_ocean_batch_size",
    "mw_ocean_set_stream", "mw_ocean_use_own_stream", "mw_ocean_get_stream", "mw_ocean_synchronize", "mw_ocean_set_choppiness",
    "mw_ocean_set_spectrum", "mw_ocean_get_spectrum", "mw_ocean_reinit_spectrum", "mw_ocean_get_phase", "mw_ocean_set_phase",
    "mw_ocean_set_timer", "mw_ocean_normal_length", "mw_ocean_set_normal_length", "mw_comm_unique_id", "mw_tiles_create",
    "mw_tiles_create_rank", "mw_tiles_destroy", "mw_tiles_count",
    "mw_tiles_local_count", "mw_tiles_ocean", "mw_tiles_evaluate", "mw_tiles_outputs", "mw_tiles_generate_texture", "mw_tiles_textures",
    "mw_tiles_generate_texture_steps", "mw_tiles_frames",
    "mw_tiles_gather", "mw_tiles_gathered",
    "mw_tiles_synchronize", "mw_ocean_rest_mesh", "mw_ocean_index_count",
    "mw_ocean_grid_size", "mw_ocean_evaluate", "mw_ocean_update", "mw_ocean_timer", "mw_ocean_reset_timer",
    "mw_ocean_evaluate_device", "mw_ocean_max_batch", "mw_ocean_generate_texture",
    "mw_ocean_generate_texture_device", "mw_ocean_generate_texture_steps_device", "mw_ocean_generate_texture_steps_rgba_device",
    "mw_ocean_generate_texture_steps", "mw_ocean_generate_texture_steps_rgba", "mw_ocean_max_frames", "mw_ocean_advance_phase", "mw_ocean_frame_textures", "mw_host_register", "mw_host_unregister", "mw_ocean_generate_texture_rgba", "mw_ocean_generate_texture_rgba_device",
    "mw_ocean_displace_mesh", "mw_ocean_displace_mesh_device", "mw_gerstner_displace",
    "mw_gerstner_displace_device", "mw_gerstner_displace_steps_device", "mw_gerstner_max_steps", "mw_pond_displace", "mw_pond_displace_device",
]
#: measurement and test hooks (include/mistral_water_hooks.h): exported, but not part of the drop-in boundary
HOOK_SYMBOLS = ["mw_ocean_profile_kernels", "mw_ocean_profile_kernels_stats", "mw_debug_pass1_time_group", "mw_debug_omega_t", "mw_debug_evaluate_hds", "mw_debug_get_omega", "mw_debug_sincos",
                "mw_debug_sincos_fast", "mw_debug_stream_read", "mw_debug_wave_transpose4", "mw_debug_set_switch", "mw_debug_get_switch"]


def build_id() -> str:
    return lib().mw_build_id().decode()


def check(status: int):
    if status != MW_OK:
        raise MistralWaterError(status, lib().mw_last_error().decode("utf-8", "replace"))


def set_switch(name: str, value: int):
    """Test hook: one of the library's run-time plan switches (csrc/mw_switches.h), process-wide."""
    check(lib().mw_debug_set_switch(name.encode(), int(value)))


def get_switch(name: str) -> int:
    return int(lib().mw_debug_get_switch(name.encode()))


def is_lab_build() -> bool:
    """True when the loaded library was built with -DMW_LAB (a measurement build: knob overrides, cycle stamps, switches from the environment)."""
    return " lab:" in build_id()


def require_product_build(who: str):
    """bench.py and the tests measure and check the PRODUCT: a lab build is refused unless MW_ALLOW_LAB=1 says the caller knows."""
    if is_lab_build() and os.environ.get("MW_ALLOW_LAB") != "1":
        raise RuntimeError(f"{who}: the loaded library is a lab build ({build_id()}); set MW_ALLOW_LAB=1 for an A/B run")
