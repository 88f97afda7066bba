"""CPU: the C-ABI library loads, exports every symbol include/hip_trace.h declares, and fails
cleanly (no abort, an error code and a message) when there is no GPU or the arguments are bad."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:hip_trace|rt)_[A-Za-z0-9_]+)\s*\(", text)))


def test_hip_trace_exports_every_declared_symbol():
    from raytracer_amd import hiptrace
    names = declared_functions("hip_trace.h")
    assert "hip_trace_init" in names and "hip_trace_frame" in names and "hip_trace_upload_scene" in names
    for name in names:
        assert hasattr(hiptrace.lib, name), name
    assert set(names) == set(hiptrace.EXPORTS)


def test_rt_host_exports_every_declared_symbol(host):
    for name in declared_functions("rt_host.h"):
        assert hasattr(host.lib, name), name


def test_product_library_does_not_link_the_oracle():
    import subprocess
    for lib in ("libhiptrace.so", "librthost.so"):
        out = subprocess.run(["ldd", os.path.join(ROOT, "raytracer_amd", "lib", lib)], capture_output=True, text=True).stdout
        assert "oracle" not in out and "libref" not in out


def test_init_rejects_bad_arguments():
    from raytracer_amd import hiptrace
    h = C.c_void_p()
    assert hiptrace.lib.hip_trace_init(None, None) == -1
    for kwargs, code in (({"width": 0}, -1), ({"maxDepth": 0}, -1), ({"maxDepth": 9}, -1), ({"raysPerPixel": 0}, -1), ({"raysPerPixel": 65}, -1),
                         ({"shadingModel": 2}, -7), ({"jitterMode": 3}, -7), ({"rank": 2, "worldSize": 2}, -1)):
        base = dict(width=64, height=36, maxDepth=2, raysPerPixel=1, shadingModel=0, jitterMode=0, device=-1, rank=0, worldSize=1)
        base.update(kwargs)
        opt = hiptrace.Options(**base)
        assert hiptrace.lib.hip_trace_init(C.byref(h), C.byref(opt)) == code, kwargs
        assert not h.value
        assert hiptrace.lib.hip_trace_last_error(None)
    hiptrace.lib.hip_trace_destroy(None)   # NULL is accepted


def test_ctypes_records_match_the_header(tmp_path):
    """The Python mirror of the ABI's records has the sizes and field offsets a C compiler gives the header's."""
    import subprocess
    from raytracer_amd import hiptrace
    probes = [("HipTraceOptions", hiptrace.Options, ["width", "maxDepth", "accel", "lightGrid", "reserved"]),
              ("HipTraceSceneInfo", hiptrace.SceneInfo, ["treeLevels", "deviceBytes", "lightGridBytes", "lightGridReferences", "lightGridLights"])]
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "hip_trace.h"', 'int main(void) {']
    for name, _mirror, fields in probes:
        lines.append(f'    printf("{name} %zu", sizeof({name}));')
        lines += [f'    printf(" %zu", offsetof({name}, {field}));' for field in fields]
        lines.append('    printf("\\n");')
    lines += ['    return 0;', '}']
    source = tmp_path / "sizes.c"
    source.write_text("\n".join(lines))
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(source), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    for (name, mirror, fields), line in zip(probes, out):
        got = [int(v) for v in line.split()[1:]]
        assert line.split()[0] == name
        assert got == [C.sizeof(mirror)] + [getattr(mirror, field).offset for field in fields], name


def test_init_rejects_an_unknown_light_grid_setting():
    from raytracer_amd import hiptrace
    h = C.c_void_p()
    opt = hiptrace.Options(width=64, height=36, maxDepth=2, raysPerPixel=1, device=-1, worldSize=1, lightGrid=2)
    assert hiptrace.lib.hip_trace_init(C.byref(h), C.byref(opt)) == -7 and not h.value
    assert b"lightGrid" in hiptrace.lib.hip_trace_last_error(None)


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from raytracer_amd import hiptrace
    with pytest.raises(hiptrace.HipTraceError) as info:
        hiptrace.Tracer(64, 36, 2)
    assert info.value.status == -2


def test_tile_owner_partition():
    from raytracer_amd import hiptrace
    for tiles_x, tiles_y, world in ((240, 135, 8), (4, 3, 2), (5, 3, 2), (9, 7, 4), (240, 135, 1)):
        seen = {}
        for ty in range(tiles_y):
            for tx in range(tiles_x):
                k = ty * tiles_x + tx
                owner = hiptrace.tile_owner(tx, ty, tiles_x, world)
                assert 0 <= owner < world
                assert (owner, k // world) not in seen, "two tiles share a local slot"
                seen[(owner, k // world)] = k
        per_rank = [sum(1 for (o, _l) in seen if o == r) for r in range(world)]
        assert max(per_rank) - min(per_rank) <= 1
        if world > 1 and tiles_x % world == 0:   # ownership must not be constant down a column
            assert len({hiptrace.tile_owner(0, ty, tiles_x, world) for ty in range(min(tiles_y, world))}) > 1


def test_headers_compile_after_the_references_own(tmp_path):
    """A translation unit that includes the reference's scene.h, octree.h and utils/image.h BEFORE hip_trace.h and
    gpu_hip.h: the library's headers step aside for the reference's definitions of the same records (include guards)
    and the result compiles without a warning.  Needs the reference tree; skipped where it is absent."""
    import subprocess
    ref = "/root/reference/src"
    if not os.path.exists(os.path.join(ref, "scene.h")):
        pytest.skip("the reference tree is not present")
    source = tmp_path / "both.c"
    source.write_text("\n".join([
        '#include "scene.h"', '#include "octree.h"', '#include "utils/image.h"', '#include "hip_trace.h"', '#include "gpu_hip.h"',
        'int use(Scene* scene, Octree* octree, Image* image) {',
        '    GPUContext* context = gpu_initContext(scene, octree, 1);',
        '    if (!context) return 1;',
        '    gpu_renderScene(context, scene, image);',
        '    void* frame = 0;',
        '    int rc = gpu_hip_lastFrame(context, &frame, 0, 0);',
        '    gpu_destroyContext(context);',
        '    return rc + (int)sizeof(HipTraceOptions) + (int)sizeof(Camera) + (int)sizeof(OctreeNode);',
        '}', '']))
    proc = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-I", ref, "-I", os.path.join(ROOT, "include"), "-c", str(source),
                           "-o", str(tmp_path / "both.o")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, proc.stdout
    # and the records are the same size either way round
    sizes = []
    for first in (True, False):
        probe = tmp_path / f"sizes{int(first)}.c"
        heads = ['#include "scene.h"', '#include "octree.h"', '#include "utils/image.h"'] if first else []
        probe.write_text("\n".join(heads + ['#include "gpu_hip.h"', '#include <stdio.h>',
                                            'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(Scene), sizeof(Camera), sizeof(Octree), sizeof(OctreeNode),'
                                            ' sizeof(Triangle), sizeof(Material), sizeof(Image)); return 0; }', '']))
        exe = tmp_path / f"sizes{int(first)}"
        subprocess.run(["gcc", "-std=gnu11", "-I", ref, "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)], check=True)
        sizes.append(subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout)
    assert sizes[0] == sizes[1], sizes
