"""GPU: the randomised parity sweeps of tests/fuzz, a few dozen cases each with fixed seeds.

Each sweep draws scenes, cameras and options at random and holds the frame against the CPU oracle (hit mask,
primitive id and hit distance bit-exact, RGBA8 within 1/255 on >= 99.9 % of pixels) or, for the light grids, against
the frame rendered without them (identical bytes).  The same modules run by hand with hundreds of cases and other
seeds: `python tests/fuzz/fuzz_parity.py 400 7`.
"""
import importlib.util
import os

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
HERE = os.path.dirname(os.path.abspath(__file__))


def sweep(name, count, seed):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, "fuzz", name + ".py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    lines = []
    try:
        module.run(count, seed, lines.append)
    except AssertionError:
        pytest.fail("\n".join(lines[-3:]))
    assert "passed" in lines[-1], lines[-1]
    return lines


def test_random_scenes_cameras_and_options_match_the_oracle():
    sweep("fuzz_parity", 30, 2024)


def test_light_grids_change_nothing_in_random_scenes():
    lines = sweep("fuzz_grid", 60, 77)
    # the sweep is about grids: most of its scenes must have got some
    assert int(lines[-1].split("passed,")[1].split()[0]) >= 40, lines[-1]


def test_planes_and_lights_next_to_them_match_the_oracle():
    sweep("fuzz_planes", 60, 31)


def test_random_cameras_on_the_million_triangle_scene_match_the_oracle():
    sweep("fuzz_s1", 3, 5)
