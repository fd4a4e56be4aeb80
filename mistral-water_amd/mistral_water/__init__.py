"""mistral_water -- host-side mirror of the reference's MonoBehaviour API over the C ABI.

The reference's host language is C# (Unity); no C# toolchain exists in the build image, so the host
side above the C ABI is mirrored here in Python (for the tests / bench harness) and in C++
(``mistral-water_amd/host/``), keeping the public field names and the Awake()/Update() lifecycle of
``S/FFTMesh.cs`` and ``S/OceanRenderer.cs``.  The C# P/Invoke binding a Unity maintainer would add is
shown in INTEGRATION.md.
"""
from ._native import (MW_EDEVICE, MW_EINVAL, MW_ENOTPOW2, MW_ESTATE, MW_OK, MW_OUT_COLOR_RGBA,  # noqa: F401
                      MW_OUT_WHITE_SCALAR, MW_SEM_FFTMESH, MW_SEM_OCEANRENDERER, MistralWaterError, MwParams,
                      build_native, check, lib)
from .ocean import (FFTMesh, Ocean, OceanRenderer, PondMaterial, Tiles, Vector2, gerstner_displace,  # noqa: F401
                    gerstner_displace_steps_device, host_register, host_unregister)
from ._native import MW_POND_GERSTNER, MW_POND_GERSTNER_LEVEL_ONE, MW_POND_WAVE  # noqa: F401
from ._native import get_switch, set_switch  # noqa: F401,E402  (test hooks: run-time plan switches)
