"""examples/frame_loop.cpp: the reference's frame loop (main.rs:381-454) on the C ABI from compiled host code - the
binding a Rust host would use, in C++ because rustc is absent.  Built with g++ against the in-tree
libpt_hip.so, run as its own process, and its PPM compared byte for byte with the image the Python mirror renders."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import pathtracer_amd as pt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "pathtracer_amd", "lib")


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_compiled_frame_loop_writes_the_same_ppm(gpu, tmp_path):
    exe, ppm = tmp_path / "frame_loop", tmp_path / "out.ppm"
    build = subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "examples", "frame_loop.cpp"), "-L" + LIBDIR, "-lpt_hip",
                            "-Wl,-rpath," + LIBDIR, "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    dim, iterations = 160, 12
    run = subprocess.run([str(exe), os.path.join(ROOT, "models", "Test.obj"), str(ppm), str(dim), str(iterations), "2", "7"],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-3000:])
    assert "gfx950" in run.stdout and f"{dim * dim * iterations} samples" in run.stdout
    scene = pt.Scene(pt.scene_data("Test"), gpu, options={})
    img = scene.render(pt.Config(dim=dim, samples_pr_pixel=iterations, bounces=2, seed=7))
    mine = tmp_path / "mine.ppm"
    pt.save_ppm(str(mine), pt.resolve_rgb8(img, iterations))
    assert ppm.read_bytes() == mine.read_bytes()
    assert np.any(pt.resolve_rgb8(img, iterations))
