"""ASan + UBSan build of the C-ABI's host side (SURVEY.md section 5 "sanitizer build of the shim"): `python off-policy_amd/build.py
--sanitize` -> libope_asan.so, driven by tests/sanitizer_host_driver.py in a subprocess with the sanitizer runtime preloaded. No GPU."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_mod():
    spec = importlib.util.spec_from_file_location("ope_build", os.path.join(ROOT, "off-policy_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_host_side_of_the_cabi_is_clean_under_asan_and_ubsan():
    b = _build_mod()
    rt = b.sanitizer_runtime()
    if rt is None:
        pytest.skip("no AddressSanitizer runtime in this toolchain")
    lib = b.build(sanitize=True, verbose=False)          # ~1 min the first time, nothing when up to date
    env = dict(os.environ, OPE_LIB_PATH=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitizer_host_driver.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SANITIZER_DRIVER_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
