"""CPU tier: the checkers under AddressSanitizer + UndefinedBehaviorSanitizer.

The oracle (oracle/*.c) and the host emulation of the kernels' phase functions (tests/emul/emul_fftmesh.cpp, i.e. the
product's own mw_math.h / *_kernels.h compiled for the host) are rebuilt with -fsanitize=address,undefined and their whole
CPU test files run against those builds in a subprocess with libasan preloaded.  An out-of-bounds LDS index map, a ragged
tail read past a vertex array or a signed overflow in an index computation aborts the subprocess."""
import os
import subprocess
import sys

import pytest

from conftest import REPO


def _rt(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.parametrize("files", [["tests/test_oracle.py"], ["tests/test_emul.py", "tests/test_ocean_renderer.py"]])
def test_checkers_under_asan_ubsan(files):
    asan, ubsan = _rt("libasan.so"), _rt("libubsan.so")
    if not asan:
        pytest.skip("gcc has no libasan here")
    env = dict(os.environ)
    env.update(MW_SANITIZE="1", LD_PRELOAD=":".join(x for x in (asan, ubsan) if x),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + files,
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in r.stdout
