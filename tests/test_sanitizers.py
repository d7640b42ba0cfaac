"""AddressSanitizer + UndefinedBehaviorSanitizer over the product's host C code and the oracle (CPU build only; GPU
sanitizers are not available on this pool): tests/c/sanitize_host.c feeds the scenario parser, the upscaler, the
marker seeding and both frame formatters awkward inputs and cross-checks every result against the oracle's."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_host_code_and_oracle_are_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_host")
    cmd = ["gcc", "-std=gnu99", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "euler_amd", "csrc"),
           "-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "c", "sanitize_host.c"),
           os.path.join(ROOT, "euler_amd", "csrc", "euler_host.c"), os.path.join(ROOT, "oracle", "euler_oracle.c"), "-lm", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "sanitize" in b.stderr and "cannot find" in b.stderr:
        pytest.skip("libasan / libubsan not installed")
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "clean" in r.stdout, (r.stdout[-1000:], r.stderr[-4000:])
