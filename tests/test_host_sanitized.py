"""The library's host code (OBJ / MTL / PNG loaders, BVH build + validation + four-wide collapse, texture pyramid,
resolve, PPM, checkpoints) compiled with g++ -fsanitize=address,undefined and run on the reference's model files,
on damaged copies of them and on random triangle soups (tests/host_sanitize.cpp).  Sanitizers are host-only here:
GPU AddressSanitizer is not available on this pool, and the device code is covered by the parity tests instead."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pathtracer_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    exe = tmp_path / "host_sanitize"
    srcs = [os.path.join(ROOT, "tests", "host_sanitize.cpp")] + [os.path.join(CSRC, f) for f in ("pt_obj.cpp", "pt_bvh.cpp", "pt_image.cpp")]
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=undefined", "-pthread", "-o", str(exe)] + srcs + ["-lz"],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe), os.path.join(ROOT, "models"), str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-6000:])
    assert "host code clean" in run.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_threaded_bvh_build_is_clean_under_tsan(tmp_path):
    exe = tmp_path / "host_tsan"
    srcs = [os.path.join(ROOT, "tests", "host_sanitize.cpp")] + [os.path.join(CSRC, f) for f in ("pt_obj.cpp", "pt_bvh.cpp", "pt_image.cpp")]
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-DPT_THREADS_ONLY", "-pthread", "-o", str(exe)] + srcs + ["-lz"],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe), os.path.join(ROOT, "models"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    if run.returncode != 0 and "FATAL: ThreadSanitizer: unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-6000:])
    assert "threaded build clean" in run.stdout
