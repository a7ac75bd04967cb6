"""CPU-side sanitizer run: `make -C oracle asan` compiles the product's
header-only NUTS driver and threefry (bpl-next_amd/csrc/nuts.hpp, threefry.hpp, through
oracle/nuts_harness.cpp), the C oracle and the CPU port with -fsanitize=address,undefined and runs
them on seeded inputs (oracle/sanitize_driver.cpp).  GPU AddressSanitizer is not available on the
pool, so this is where the product's host C++ is sanitised."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_address_and_undefined_behaviour_sanitizers_are_clean():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True,
                       timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "sanitize_driver: ok" in out
    assert "AddressSanitizer" not in out and "runtime error" not in out and "LeakSanitizer" not in out
