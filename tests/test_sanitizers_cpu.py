"""The host-only part of librtmi (camera derivation, scene generator, SAH BVH builder) and the C oracle under
AddressSanitizer + UndefinedBehaviorSanitizer -- the CPU build is where sanitizers can run (the GPU pool refuses them)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_code_and_oracle_are_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    obj = str(tmp_path / "rt_oracle.o")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    subprocess.run(["gcc", "-std=c11", "-c", *san, "-pthread", os.path.join(ROOT, "oracle", "rt_oracle.c"), "-o", obj],
                   check=True)
    subprocess.run(["g++", "-std=c++17", *san, "-pthread", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "raytracing.cpp_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"),
                    os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"),
                    os.path.join(ROOT, "raytracing.cpp_amd", "csrc", "rtmi_host.cpp"), obj, "-lm", "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sanitizers: ok" in r.stdout
