"""Host-side product code under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5:
the reference has no sanitizer configuration; this repository runs one in its CPU suite)."""
import os
import subprocess

import dgtest as T


def test_host_code_is_asan_ubsan_clean(tmp_path):
    exe = str(tmp_path / "sanitize_main")
    csrc = os.path.join(T.ROOT, "discregrid_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", os.path.join(T.ROOT, "tests", "cpp", "sanitize_main.cpp"),
                           os.path.join(T.ROOT, "tests", "emu", "wave_emu.cpp"), os.path.join(csrc, "dg_build.cpp"),
                           "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-3000:]
    assert "0 problems" in text
