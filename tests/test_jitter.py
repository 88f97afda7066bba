"""Hashed light jitter (soft shadows): the oracle against frames produced by the reference's own
raytracer_raycast with the hashed stream fed in at its random_bilateral() seam (CPU, bit-exact),
the octree walk against brute force under jitter, and the HIP path against the oracle (GPU)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = [("default_96x54_d5", "default", (96, 54), 5), ("default_64x36_d2", "default", (64, 36), 2),
         ("mesh20x12_80x45_d4", "mesh", (80, 45), 4)]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def scene_of(host, kind, w, h):
    return host.lib.rt_scene_init(w, h) if kind == "default" else host.lib.rt_scene_synth_mesh(w, h, 20, 12, 0, 3, 8)


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "frames_jitter.npz"))


def test_hash_values(oracle, golden):
    from raytracer_amd import hiptrace
    for key, value in zip(golden["hash_keys"], golden["hash_values"]):
        assert oracle.lib.oracle_jitter_hash(*map(int, key)) == int(value)
        assert hiptrace.lib.hip_trace_jitter_hash(*map(int, key)) == int(value)


@pytest.mark.parametrize("name,kind,size,depth", CASES)
def test_oracle_matches_reference_frames_under_jitter(host, oracle, golden, name, kind, size, depth):
    scene = scene_of(host, kind, *size)
    cam = host.camera_copy(scene.contents.camera)
    out = oracle.render(scene, None, cam, depth, jitter=oracle.JITTER_HASHED, want=("rgb",))
    assert np.array_equal(bits(out["rgb"]), bits(golden[name + "/rgb"])), "float RGB differs from the reference"
    hard = oracle.render(scene, None, cam, depth, want=("rgb",))
    assert not np.array_equal(out["rgb"], hard["rgb"]), "jitter changed nothing"
    if oracle.ref is not None:       # the compiled reference, live, where it exists
        live, disagreements = oracle.ref_render_jittered(scene, cam, depth)
        assert disagreements == 0
        assert np.array_equal(bits(live), bits(out["rgb"]))
    host.lib.rt_scene_destroy(scene)


def test_shading_order_is_postorder_of_the_ray_tree(host, oracle):
    scene = host.lib.rt_scene_init(64, 36)
    cam = host.camera_copy(scene.contents.camera)
    offsets, positions = oracle.shading_order(scene, None, cam, 4)
    assert offsets[0] == 0 and offsets[-1] == positions.size and np.all(np.diff(offsets.astype(np.int64)) >= 0)
    saw_glass = False
    for i in range(64 * 36):
        ps = [int(p) for p in positions[offsets[i]:offsets[i + 1]]]
        if not ps:
            continue
        assert ps[-1] == 1                                  # the primary hit is shaded last
        assert len(set(ps)) == len(ps) and all(1 <= p < 16 for p in ps)
        for k, p in enumerate(ps):                          # children come before their parent,
            if p > 1:
                assert p // 2 in ps[k + 1:]
            if p % 2 == 1 and p > 1 and p - 1 in ps:        # ... the refracted subtree before the reflected one
                assert k < ps.index(p - 1)
                saw_glass = True
    assert saw_glass
    host.lib.rt_scene_destroy(scene)


def test_octree_walk_equals_brute_force_under_jitter(host, oracle):
    scene = host.lib.rt_scene_synth_mesh(96, 54, 30, 20, 0, 3, 6)
    tree = host.lib.rt_octree_buildFromScene(scene)
    cam = host.camera_copy(scene.contents.camera)
    a = oracle.render(scene, None, cam, 3, jitter=1, want=("rgb",))
    b = oracle.render(scene, tree, cam, 3, jitter=1, want=("rgb",))
    assert np.array_equal(bits(a["rgb"]), bits(b["rgb"]))
    host.lib.rt_octree_destroy(tree)
    host.lib.rt_scene_destroy(scene)


@pytest.mark.gpu
@pytest.mark.parametrize("accel", [pytest.param(0, id="bvh"), pytest.param(1, id="octree")])
@pytest.mark.parametrize("kind,size,depth", [("default", (200, 112), 5), ("mesh", (192, 108), 4)])
def test_gpu_matches_oracle_under_jitter(host, oracle, kind, size, depth, accel):
    from raytracer_amd import hiptrace
    w, h = size
    scene = scene_of(host, kind, w, h) if kind == "default" else host.lib.rt_scene_synth_mesh(w, h, 60, 40, 0, 5, 10)
    tree = host.lib.rt_octree_buildFromScene(scene)
    cam = host.camera_copy(scene.contents.camera)
    cpu = oracle.render(scene, None, cam, depth, jitter=oracle.JITTER_HASHED)
    tr = hiptrace.Tracer(w, h, depth, accel=accel, jitter=hiptrace.JITTER_HASHED)
    tr.upload_scene(scene, tree)
    gpu = tr.frame(cam)
    assert np.array_equal(gpu["hitMask"], cpu["hitMask"]) and np.array_equal(gpu["primitiveId"], cpu["primitiveId"])
    err = np.abs(gpu["rgba8"].view(np.uint8).astype(np.int16) - cpu["rgba8"].view(np.uint8).astype(np.int16))
    frac = (err.reshape(-1, 4).max(axis=1) <= 1).mean()
    assert frac >= 0.999, f"only {frac:.5f} of pixels within 1/255, max error {err.max()}/255"
    assert 0 < gpu["rays"]["shadow"] <= cpu["counters"]["shadow"]     # perfect mirrors skip theirs on the device
    hard = hiptrace.Tracer(w, h, depth, accel=accel)
    hard.upload_scene(scene, tree)
    assert not np.array_equal(hard.frame(cam)["rgba8"], gpu["rgba8"])
    tr.close()
    hard.close()
    host.lib.rt_octree_destroy(tree)
    host.lib.rt_scene_destroy(scene)
