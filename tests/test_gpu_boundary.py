"""The drop-in boundary on a GPU: the Change<T> state machine, validation and error
behaviour of GpuWrapper::update / update_uniforms (gpu_wrapper.rs:116-300,469-576,
render_config.rs:163-268) and the iterator contract (engine-pathtracer lib.rs:127-234),
exercised through the C ABI exactly as the Rust shim would."""
import threading

import numpy as np
import pytest

from renderbaby_amd import Change, Engine, RenderConfig, RenderError, abi, scenes
from tests import _oracle

pytestmark = pytest.mark.gpu


def _engine(scene, **kw):
    rc = RenderConfig.from_scene(scene)
    e = Engine.new(rc, **kw)
    return e, rc


def test_first_update_requires_create():
    s = scenes.cornell(16, 8, 1, 2)
    e, rc = _engine(s)
    bad = RenderConfig.from_scene(s)
    bad.lights = Change.update(s.lights)
    with pytest.raises(RenderError) as ei:
        e.render(bad)
    assert ei.value.code == 8 and "Invalid Lights" in ei.value.message  # validate_init
    e.render(rc)  # the engine is still usable
    e.close()


def test_validate_rejects_bad_values_after_init():
    s = scenes.cornell(16, 8, 1, 2)
    e, rc = _engine(s)
    e.render(rc)
    up = RenderConfig.from_scene(s, create=False)

    def with_uniforms(**kw):
        u = s.uniforms.copy()
        for k, v in kw.items():
            u["camera"][k] = v
        r = RenderConfig.from_scene(s, create=False)
        r.uniforms = Change.update(u)
        return r

    for rcx, code in ((with_uniforms(pane_distance=150.0), 1), (with_uniforms(pane_width=-1.0), 2),
                      (with_uniforms(dir=(0, 0, 0)), 3)):
        with pytest.raises(RenderError) as ei:
            e.render(rcx)
        assert ei.value.code == code
    sp = s.spheres.copy()
    sp[0]["radius"] = 0.0
    up.spheres = Change.update(sp)
    with pytest.raises(RenderError) as ei:
        e.render(up)
    assert ei.value.code == 5
    odd = RenderConfig.from_scene(s, create=False)
    odd.uvs = Change.update(np.zeros(3, np.float32))
    with pytest.raises(RenderError) as ei:
        e.render(odd)
    assert ei.value.code == 6
    dele = RenderConfig.from_scene(s, create=False)
    dele.uniforms = Change.delete()
    with pytest.raises(RenderError) as ei:
        e.render(dele)
    assert ei.value.code == 10  # CannotDeleteNonexistent
    for f in ("uvs", "meshes", "lights", "textures"):  # todo!() arms
        d = RenderConfig.from_scene(s, create=False)
        setattr(d, f, Change.delete())
        with pytest.raises(RenderError) as ei:
            e.render(d)
        assert ei.value.code == 14, f
    e.close()


def test_keep_update_delete_semantics_match_fresh_renders():
    a = scenes.cornell(40, 24, 3, 4)
    e, rc = _engine(a)
    f0 = e.render(rc).pixels.copy()
    assert np.array_equal(f0, _oracle.render(a)[2])

    # Update uniforms only (more samples, new resolution), Keep everything else: Keep leaves the
    # uniforms' own counts in force (gpu_wrapper.rs:475-495), so the adapter always sends them
    b = a.with_params(width=32, height=20, spp=2)
    keep = RenderConfig(uniforms=Change.update(b.uniforms))
    f1 = e.render(keep).pixels.copy()
    assert f1.shape == (20, 32, 4)
    assert np.array_equal(f1, _oracle.render(b)[2])

    # Create after init is ignored for spheres ("Create not allowed after initialization") ...
    other = scenes.cornell(32, 20, 2, 4, seed=99)
    ign = RenderConfig(uniforms=Change.update(b.uniforms), spheres=Change.create(other.spheres))
    assert np.array_equal(e.render(ign).pixels, f1)
    # ... but acts as Update for the BVH fields (gpu_wrapper.rs:242-280)
    upd = RenderConfig(uniforms=Change.update(b.uniforms), spheres=Change.update(other.spheres))
    f2 = e.render(upd).pixels.copy()
    assert np.array_equal(f2, _oracle.render(other)[2])

    # Delete spheres: spheres_count = 0
    nos = scenes.Scene(b.uniforms, np.zeros(0, abi.SPHERE), b.lights, b.meshes, b.bvh_nodes, b.bvh_indices,
                       b.bvh_triangles, b.uvs)
    f3 = e.render(RenderConfig(uniforms=Change.update(b.uniforms), spheres=Change.delete())).pixels.copy()
    assert np.array_equal(f3, _oracle.render(nos)[2])

    # Delete the BVH: node count 0 => no triangles are hit
    bare = scenes.Scene(b.uniforms, np.zeros(0, abi.SPHERE), b.lights, np.zeros(0, abi.MESH), np.zeros(0, abi.BVH_NODE),
                        np.zeros(0, np.uint32), np.zeros(0, abi.GPU_TRIANGLE), np.zeros(0, np.float32))
    f4 = e.render(RenderConfig(uniforms=Change.update(b.uniforms), bvh_nodes=Change.delete(),
                               bvh_indices=Change.delete(), bvh_triangles=Change.delete())).pixels.copy()
    assert np.array_equal(f4, _oracle.render(bare)[2])
    e.close()


def test_update_without_uniforms_fails_like_the_reference_panics():
    s = scenes.cornell(16, 8, 1, 2)
    e, rc = _engine(s)
    e.render(rc)
    with pytest.raises(RenderError) as ei:
        e.render(RenderConfig())  # all Keep: gpu_wrapper.rs:303-329 "Uniforms must be initialized"
    assert ei.value.code == 11 and "Uniforms must be initialized" in ei.value.message
    e.render(RenderConfig.from_scene(s, create=False))
    e.close()


def test_malformed_bvh_is_refused_not_hung():
    s = scenes.mesh_scene(12, 12, 16, 8, 1, 2)
    assert len(s.bvh_nodes) >= 3
    e, rc = _engine(s)
    nodes = s.bvh_nodes.copy()
    nodes[1]["primitive_count"] = 0
    nodes[1]["left"] = 0  # cycle back to the root
    nodes[1]["right"] = 0
    bad = RenderConfig.from_scene(s)
    bad.bvh_nodes = Change.create(nodes)
    with pytest.raises(RenderError) as ei:
        e.render(bad)
    assert ei.value.code == 13
    nodes = s.bvh_nodes.copy()
    leaf = int(np.flatnonzero(nodes["primitive_count"] > 0)[0])
    nodes[leaf]["primitive_count"] = 10_000_000
    bad.bvh_nodes = Change.create(nodes)
    with pytest.raises(RenderError) as ei:
        e.render(bad)
    assert ei.value.code == 13
    tris = s.bvh_triangles.copy()
    tris[0]["mesh_index"] = 77
    bad = RenderConfig.from_scene(s)
    bad.bvh_triangles = Change.create(tris)
    with pytest.raises(RenderError) as ei:
        e.render(bad)
    assert ei.value.code == 7
    e.close()


def test_iterator_driven_from_a_worker_thread():
    # the reference moves the iterator to the FrameBuffer worker thread (frame_buffer.rs:141-148)
    s = scenes.cornell(32, 16, 3, 3)
    e, rc = _engine(s)
    it = e.frame_iterator(rc)
    frames, errs = [], []

    def pump():
        try:
            while it.has_next():
                frames.append(it.next().pixels.copy())
        except Exception as ex:  # pragma: no cover
            errs.append(ex)

    t = threading.Thread(target=pump)
    t.start()
    t.join()
    assert not errs and len(frames) == 3
    assert np.array_equal(frames[-1], _oracle.render(s)[2])
    it.destroy()
    e.close()


def test_frame_contract():
    s = scenes.feature_scene(20, 12, 2, 3)
    e, rc = _engine(s)
    f = e.render(rc)
    f.validate()
    assert f.pixels.dtype == np.uint8 and f.pixels.shape == (12, 20, 4) and np.all(f.pixels[..., 3] == 255)
    # x mirrored relative to the shader's pixel index (gpu_wrapper.rs:446-458)
    acc = e.read_accumulation()
    L = _oracle.lib()
    fin = acc[..., :3] / acc[..., 3:4]
    mapped = (fin / (fin + np.float32(1.0))).astype(np.float32)
    px = np.array([[_oracle.color_map(mapped[y, x]) for x in range(20)] for y in range(12)], dtype=np.uint32)
    assert np.array_equal(f.pixels[..., 0], (px & 255).astype(np.uint8)[:, ::-1])
    e.close()


def test_iterator_delivering_every_n_passes():
    # extension (rb_iter_set_passes_per_frame): one frame per n samples; the frames are the
    # reference's frames n-1, 2n-1, ... and the last one
    s = scenes.cornell(40, 24, 7, 4)
    rc = RenderConfig.from_scene(s)
    eng = Engine.new(rc)
    it = eng.frame_iterator(rc, passes_per_frame=3)
    frames = [f.pixels.copy() for f in it]
    assert len(frames) == 3                                   # 3 + 3 + 1 samples
    for frame, upto in zip(frames, (3, 6, 7)):
        assert np.array_equal(frame, _oracle.render(s, 0, upto)[2]), upto
    with pytest.raises(Exception) as ei:
        it.next()
    assert "No more frames available" in str(ei.value)
    # back to one frame per pass
    it = eng.frame_iterator(RenderConfig.from_scene(s, create=False))
    assert len([1 for _ in it]) == 7
    eng.close()


@pytest.mark.parametrize("kw", [dict(fast_bvh=True), dict(device_bvh=True), dict(reference_walk=True), dict()],
                         ids=["host-sah", "device-ploc", "reference-walk", "default"])
def test_scene_updates_rebuild_the_library_trees(kw):
    # Change::Update of the BVH / triangle / sphere fields must rebuild everything derived from them:
    # prepared triangles, the opt-in walk's tree (host or device built), the sphere tree
    a = scenes.mesh_scene(24, 24, 40, 30, 2, 4, seed=3)
    b = scenes.mesh_scene(30, 20, 40, 30, 2, 4, seed=9)
    c = scenes.spheres_scene(n=3000, width=40, height=30, spp=2, max_depth=4, extent=10.0)
    d = scenes.spheres_scene(n=40, width=40, height=30, spp=2, max_depth=4, extent=3.0)      # below the tree threshold
    eng = Engine.new(RenderConfig.from_scene(a), **kw)
    for i, s in enumerate((a, b, c, a, d, c)):
        rc = RenderConfig.from_scene(s, create=(i == 0))
        frame = eng.render(rc)
        assert np.array_equal(frame.pixels, _oracle.render(s)[2]), i
        assert np.array_equal(eng.read_accumulation().view(np.uint32), _oracle.render(s)[0].view(np.uint32)), i
    eng.close()
