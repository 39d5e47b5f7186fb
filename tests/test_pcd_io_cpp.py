"""include/ope/pcd_io.hpp (the facade's pcl::io::loadPCDFile / savePCDFile, BuildModel main.cpp:113-153,221): header values
are untrusted input.  A compiled C++ program drives the loader over well-formed and malformed files; no GPU call is made."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "object-pose-estimation_amd")


@pytest.mark.skipif(not os.path.exists(os.path.join(PKG, "libope_hip.so")), reason="libope_hip.so not built")
def test_pcd_loader_rejects_malformed_headers(tmp_path):
    exe = str(tmp_path / "pcd_io_check")
    # AddressSanitizer + UBSan on the host-side C++ (the loader parses untrusted headers); the program makes no GPU call, so the
    # sanitizer never sees the HIP runtime's own allocations (leak check off: the runtime's static state is not ours to free)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "pcd_io_check.cpp"),
                           "-o", exe, "-L", PKG, "-lope_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "pcd_io_check: ok" in r.stdout
