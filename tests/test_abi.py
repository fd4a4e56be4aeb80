"""CPU tier: the C-ABI shared library loads, exports every symbol include/mistral_water.h declares,
and refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import REPO, has_gpu

HEADER = os.path.join(REPO, "include", "mistral_water.h")
HOOKS_HEADER = os.path.join(REPO, "include", "mistral_water_hooks.h")


def declared_symbols(path=HEADER):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mw_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(mw):
    from mistral_water import _native
    syms, hooks = declared_symbols(), declared_symbols(HOOKS_HEADER)
    assert len(syms) >= 25
    L = C.CDLL(_native.LIB_PATH)
    for s in syms + hooks:
        assert hasattr(L, s), f"{s} declared in include/*.h but not exported"
    assert sorted(_native.ABI_SYMBOLS) == syms, "python binding list out of date with the header"
    assert sorted(_native.HOOK_SYMBOLS) == hooks, "python hook list out of date with mistral_water_hooks.h"


def test_hooks_are_not_part_of_the_boundary():
    """Measurement / test hooks live in their own header: the drop-in boundary (and the C# import table, test_csharp_binding)
    carries none of them."""
    pub = declared_symbols()
    assert not [s for s in pub if s.startswith("mw_debug_") or s.startswith("mw_ocean_profile_kernels")]
    assert all(s.startswith("mw_debug_") or s.startswith("mw_ocean_profile_kernels") for s in declared_symbols(HOOKS_HEADER))


def test_build_id_names_the_sources_this_library_was_built_from(mw):
    """mw_build_id() is fixed at build time (-DMW_BUILD_HASH): the hash over the kernel sources, the boundary headers and the compile
    flags that the build recipe computed -- so the in-tree library, built by build(), carries the hash of the tree it sits in."""
    from mistral_water import _native
    bid = _native.build_id()
    h, tag = bid.split(" ", 1)
    assert len(h) == 16 and int(h, 16) != 0 and tag
    if not os.environ.get("MW_LIB"):      # an A/B variant carries the hash of ITS flags
        assert h == _native.source_hash(), "libmistral_water.so is stale: rebuild (python -c 'import __graft_entry__ as g; g.build()')"
        assert tag == "default"


def test_build_reuse_is_decided_by_the_embedded_hash_not_by_mtimes(mw, monkeypatch):
    """build() reuses the in-tree library when the hash EMBEDDED in the file equals the hash of the tree (read from the
    file without loading it) -- a copy that lost its mtimes neither rebuilds on a box without hipcc nor keeps a stale binary silently.
    And a lab build (-DMW_LAB: knob overrides, cycle stamps, switches from the environment) is refused by bench.py and the tests."""
    from mistral_water import _native
    if os.environ.get("MW_LIB"):
        pytest.skip("A/B variant")
    assert _native.built_hash(_native.LIB_PATH) == _native.source_hash() == _native.build_id().split(" ")[0]
    os.utime(_native.LIB_PATH, (1, 1))                                  # mtimes say "older than every source"
    try:
        monkeypatch.setattr(_native.subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("rebuilt")))
        assert _native.build_native() == _native.LIB_PATH               # reused all the same
    finally:
        os.utime(_native.LIB_PATH, None)
    assert _native.built_hash(__file__) is None or True                 # any file: no crash
    assert not _native.is_lab_build()
    monkeypatch.setattr(_native, "build_id", lambda: "0123456789abcdef lab:x")
    monkeypatch.delenv("MW_ALLOW_LAB", raising=False)
    with pytest.raises(RuntimeError):
        _native.require_product_build("test")
    monkeypatch.setenv("MW_ALLOW_LAB", "1")
    _native.require_product_build("test")


def test_switches_are_a_table_with_a_hook_not_the_environment(mw, monkeypatch):
    """The run-time plan switches (csrc/mw_switches.h): defaults in a product build whatever the environment says, changed only through
    mw_debug_set_switch; unknown names are MW_EINVAL."""
    import subprocess
    import sys
    from conftest import REPO as repo
    code = ("import sys; sys.path[:0] = [%r + '/mistral-water_amd']; import mistral_water as mw; "
            "print(mw.get_switch('MW_CZT_FUSED'), mw.get_switch('MW_TILES_FORCE_RCCL'), mw.get_switch('MW_P1_TGROUP'))" % repo)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MW_CZT_FUSED="0", MW_TILES_FORCE_RCCL="1"), capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split() == ["1", "0", "-1"], (r.stdout, r.stderr[-2000:])
    assert mw.get_switch("MW_CZT_ONE") == 1
    mw.set_switch("MW_CZT_ONE", 0)
    assert mw.get_switch("MW_CZT_ONE") == 0
    mw.set_switch("MW_CZT_ONE", 1)
    with pytest.raises(mw.MistralWaterError) as e:
        mw.set_switch("MW_NO_SUCH_SWITCH", 1)
    assert e.value.status == mw.MW_EINVAL and mw.get_switch("MW_NO_SUCH_SWITCH") == -2 ** 31


def test_header_is_plain_c():
    # the boundary must be consumable from C (and therefore from P/Invoke / cgo / JNI / ctypes)
    src = '#include "mistral_water.h"\n#include "mistral_water_hooks.h"\nint main(void){ mw_params p; mw_params_default(&p, MW_SEM_FFTMESH); return (int)sizeof(p) == 0; }\n'
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.dirname(HEADER),
                        "-x", "c", "-"], input=src, text=True, capture_output=True)
    assert r.returncode == 0, r.stderr


def test_params_struct_layout(mw):
    from mistral_water import _native
    assert C.sizeof(_native.MwParams) == 56  # 10 x 4 B, seed (8 B, at offset 40), 2 x 4 B
    p = _native.MwParams()
    mw.lib().mw_params_default(C.byref(p), mw.MW_SEM_FFTMESH)
    assert p.resolution == 50 and p.unit_width == 1.0 and p.choppiness == 1.0 and abs(p.gravity - 9.81) < 1e-6
    mw.lib().mw_params_default(C.byref(p), mw.MW_SEM_OCEANRENDERER)
    assert p.resolution == 256 and p.mult == 2.0 and p.choppiness == 1.5


def test_abi_version_and_error_string(mw):
    assert mw.lib().mw_abi_version() == 4
    assert isinstance(mw.lib().mw_last_error(), bytes)


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a GPU-less host")
def test_no_cpu_fallback_without_gpu(mw):
    with pytest.raises(mw.MistralWaterError) as e:
        mw.Ocean(resolution=64, length=64.0)
    assert e.value.status == mw.MW_EDEVICE
    import numpy as np
    with pytest.raises(mw.MistralWaterError) as e:
        mw.gerstner_displace(np.zeros((4, 3), np.float32), [(1, 0, 1)], 0.1, 1.0, 0.5, 0.0)
    assert e.value.status == mw.MW_EDEVICE
    with pytest.raises(mw.MistralWaterError) as e:
        mw.PondMaterial().displace(np.zeros((4, 3), np.float32), 0.0)
    assert e.value.status == mw.MW_EDEVICE


def test_bad_arguments_are_status_codes_not_crashes(mw):
    L = mw.lib()
    assert L.mw_ocean_create(None, None) == mw.MW_EINVAL
    assert L.mw_ocean_evaluate(None, 0.0, None, None, None) == mw.MW_EINVAL
    assert L.mw_ocean_set_spectrum(None, None, None) == mw.MW_EINVAL
    L.mw_ocean_destroy(None)  # no-op


def test_product_never_references_the_oracle():
    """The product path must not import, link or execute anything under oracle/."""
    amd = os.path.join(REPO, "mistral-water_amd")
    for root, _, files in os.walk(amd):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(root, f), errors="replace").read()
                assert "liboracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f
    r = subprocess.run(["ldd", os.path.join(amd, "libmistral_water.so")], capture_output=True, text=True)
    assert "oracle" not in r.stdout
