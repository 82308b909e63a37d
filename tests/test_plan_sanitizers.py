"""CPU: the host-side plan builder (csrc/mlp_plan.hip is plain C++: operand maps, stream layout,
the packer's gather tables) under AddressSanitizer + UBSan for every descriptor the C ABI accepts
or rejects.  GPU sanitizers are not available on the pool; this is the host half."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_plan_builder_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "plan_asan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-x", "c++",
           "-I" + os.path.join(ROOT, "zest-nerf_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "plan_asan_main.cpp"), os.path.join(ROOT, "zest-nerf_amd", "csrc", "mlp_plan.hip"),
           "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("g++ without sanitizer runtimes: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-2000:]
    assert "plans built:" in run.stdout and int(run.stdout.split(":")[1].split()[0]) > 100
    assert int(run.stdout.split("shape plans built:")[1]) > 100
