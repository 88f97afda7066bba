"""A C program, compiled here with gcc against libgpuhip.so and librthost.so, runs the reference
program's device call sequence (reference: src/main.c:64-68, 171-178, 188): that the declared C ABI
links from C is checked on the CPU, that the program renders the frame the ctypes path renders is
checked on the GPU -- on one device and with the frame split over three device entries."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "raytracer_amd", "lib")


@pytest.fixture(scope="module")
def program(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("c") / "main_sequence")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", os.path.join(ROOT, "tests", "c", "main_sequence.c"),
           "-D__HIP_PLATFORM_AMD__", "-isystem", "/opt/rocm/include",
           "-I", os.path.join(ROOT, "include"), "-L", LIB, "-lgpuhip", "-lrthost", "-lhiptrace", "-L", "/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{LIB}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, proc.stdout
    return exe


def test_c_program_links_against_the_declared_abi(program):
    proc = subprocess.run([program], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert proc.returncode == 2 and "usage" in proc.stderr       # it starts: every library resolved


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [(), (0, 0, 0)], ids=["one-device", "three-entries"])
def test_c_program_renders_the_frame_of_the_ctypes_path(program, host, tmp_path, devices):
    from raytracer_amd import hiptrace
    w, h = 320, 180
    out = str(tmp_path / "frame.bmp")
    proc = subprocess.run([program, out, str(w), str(h), *map(str, devices)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr
    scene = host.lib.rt_scene_init(w, h)
    cam = host.camera_copy(scene.contents.camera)
    tr = hiptrace.Tracer(w, h, 5)                 # the shim's default depth: the reference kernel's 5
    tr.upload_scene(scene, None)
    frame = tr.frame(cam, want_aux=False)
    image = host.lib.rt_image_create(w, h)
    np.ctypeslib.as_array(image.contents.buffer, shape=(h * w,))[:] = frame["rgba8"].reshape(-1)
    expect = str(tmp_path / "expect.bmp")
    assert host.lib.rt_bitmap_save_image(expect.encode(), image)
    assert open(out, "rb").read() == open(expect, "rb").read()
    # the second frame of the program: camera moved one step sideways
    host.lib.rt_move_camera(scene.contents.camera, 0, 1, 0)
    moved = tr.frame(host.camera_copy(scene.contents.camera), want_aux=False)
    counts = dict(zip(proc.stdout.split()[0::2], map(int, proc.stdout.split()[1::2])))
    assert counts == moved["rays"]
    # ... which the program took from device memory where gpu_renderScene(context, scene, NULL) left it
    np.ctypeslib.as_array(image.contents.buffer, shape=(h * w,))[:] = moved["rgba8"].reshape(-1)
    assert host.lib.rt_bitmap_save_image(expect.encode(), image)
    assert open(out + ".next.bmp", "rb").read() == open(expect, "rb").read()
    tr.close()
    host.lib.rt_image_destroy(image)
    host.lib.rt_scene_destroy(scene)
