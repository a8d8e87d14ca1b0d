"""Scene exchange in the reference engine's binary entity-stream layout (directx-renderer-kurth_amd/scene_binary.py;
serialization_binary.cpp:105-274, 420-465): documented offsets, round trip, and a running world resumed from the streams."""
import struct

import numpy as np
import pytest


def test_entity_stream_layout_offsets():
    """The byte layout the module documents (derived from the reference's struct definitions under the MSVC x64 ABI)."""
    from directx_renderer_kurth_amd import scenes, scene_binary as sb
    s = scenes.Scene("one")
    b = s.add_body((1.0, 2.0, 3.0), (0.0, 0.0, 0.6, 0.8), gravity_factor=0.5, linear_damping=0.25, angular_damping=0.125)
    s.add_collider(b, scenes.OBB, (0.0, 0.0, 0.0, 1.0, 0.5, 0.25, 0.125, 1.0, 2.0, 3.0), (0.1, 0.7, 2.0))
    s.add_collider(b, scenes.SPHERE, (0.0, 1.0, 0.0, 0.75), (0.2, 0.3, 4.0))
    mp = np.arange(13, dtype=np.float32)[None] + 1.0
    vel = np.array([[1, 2, 3, 4, 5, 6]], np.float32)
    (eid, stream), = sb.dump_entities(s, velocities=vel, mass_properties=mp)
    assert eid == 0
    # component group: tag (1 + 16), transform (1 + 48), position / position_rotation / position_scale absent (3), dynamic (1), mesh + 2 lights absent (3),
    # rigid body (1 + 112), force field / cloth / cloth render absent (3), physics reference (1 + 4 + 2 * 80 + 4), 5 absent
    assert len(stream) == 17 + 49 + 3 + 1 + 3 + 113 + 3 + (1 + 4 + 160 + 4) + 5
    assert stream[0] == 1 and stream[1:8] == b"body_0\0"
    tr = stream[18:66]
    assert struct.unpack_from("<4f", tr, 0) == pytest.approx((0.0, 0.0, 0.6, 0.8)) and struct.unpack_from("<3f", tr, 16) == (1.0, 2.0, 3.0) and struct.unpack_from("<3f", tr, 28) == (1.0, 1.0, 1.0)
    assert stream[66:69] == b"\0\0\0" and stream[69] == 1 and stream[70:73] == b"\0\0\0" and stream[73] == 1
    rb = stream[74:186]
    assert struct.unpack_from("<3f", rb, 0) == (1.0, 2.0, 3.0) and struct.unpack_from("<f", rb, 12) == (4.0,)          # localCOG, invMass
    assert struct.unpack_from("<9f", rb, 16) == tuple(float(x) for x in range(5, 14))                                  # invInertia, memory order
    assert struct.unpack_from("<3f", rb, 52) == (0.5, 0.25, 0.125) and struct.unpack_from("<6f", rb, 64) == (1, 2, 3, 4, 5, 6) and rb[88:] == b"\0" * 24
    ref = stream[190:]
    assert stream[186:189] == b"\0\0\0" and stream[189] == 1 and struct.unpack_from("<I", ref, 0) == (2,)
    obb = ref[4:84]
    assert struct.unpack_from("<10f", obb, 0) == (0.0, 0.0, 0.0, 1.0, 0.5, 0.25, 0.125, 1.0, 2.0, 3.0)                  # quat @0, center @16, radius @28
    assert struct.unpack_from("<i3f", obb, 48) == pytest.approx((-1, 0.1, 0.7, 2.0)) and struct.unpack_from("<BBH", obb, 64) == (scenes.OBB, 0, 0)
    assert struct.unpack_from("<4f", ref, 84) == (0.0, 1.0, 0.0, 0.75) and ref[84 + 64] == scenes.SPHERE
    assert struct.unpack_from("<I", ref, 164) == (0,) and stream[-5:] == b"\0" * 5
    hull = sb._collider_union(scenes.HULL, (0, 0, 0, 1, 1, 2, 3, 7.0), (0.1, 0.5, 1.0))
    assert struct.unpack_from("<I", hull, 32) == (7,) and struct.unpack_from("<3f", hull, 16) == (1.0, 2.0, 3.0)          # geometryIndex behind the 8-byte alignment gap
    with pytest.raises(ValueError):
        sb.read_entity(stream[:-1])
    with pytest.raises(ValueError):
        sb.read_entity(stream + b"\0")
    with pytest.raises(ValueError):
        sb.read_entity(b"\0" * 6 + b"\1" + b"\0" * 40)    # a mesh component: outside the physics subset


def test_binary_round_trip_and_oracle_trajectory(oracle):
    from directx_renderer_kurth_amd import scenes, scene_binary as sb
    scene = scenes.by_name("shapes")
    a = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_SCALAR))
    for _ in range(20):
        a.step_internal(scene.dt)
    ents = sb.unpack(sb.pack(sb.dump_entities(scene, transforms=a.transforms(1), velocities=a.velocities(), mass_properties=a.mass_properties())))
    assert len(ents) == len(scene.bodies) + 1
    loaded, constraints, vel = sb.load_entities(ents, dt=scene.dt)
    assert not constraints and len(loaded.bodies) == len(scene.bodies) and len(loaded.colliders) == len(scene.colliders)
    assert np.array_equal(vel, a.velocities())
    b = loaded.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_SCALAR))
    assert np.array_equal(a.mass_properties(), b.mass_properties())
    b.write_state(a.transforms(1), vel)
    for _ in range(30):
        a.step_internal(scene.dt); b.step_internal(scene.dt)
    assert np.array_equal(a.transforms(1), b.transforms(1)) and np.array_equal(a.velocities(), b.velocities())


def test_constraints_are_listed_by_both_entities_and_added_once():
    from directx_renderer_kurth_amd import scenes, scene_binary as sb
    s = scenes.Scene("pair")
    a = s.add_body((0, 1, 0)); b = s.add_body((1, 1, 0))
    s.add_collider(a, scenes.SPHERE, (0, 0, 0, 0.3), (0.1, 0.5, 1.0)); s.add_collider(b, scenes.SPHERE, (0, 0, 0, 0.3), (0.1, 0.5, 1.0))
    s.add_joint("ball", a, b, (0.5, 1.0, 0.0))
    pod = bytes(range(24))
    ents = sb.dump_entities(s, constraint_pods={"ball": [pod]})
    for eid, stream in ents:                            # each of the two entities' streams holds the constraint (serialization_binary.cpp:213-231)
        ref = sb.read_entity(stream)["physics_reference"]
        assert struct.unpack_from("<I", ref, 4 + 80) == (1,) and struct.unpack_from("<iII", ref, 4 + 80 + 4) == (1, 0, 1) and ref[-24:] == pod
    _, constraints, _ = sb.load_entities(ents)
    assert constraints == [(1, 0, 1, pod)]
    bad = [(5, ents[0][1])]                             # an entity that is neither end of the constraint it lists (the reference asserts, :254)
    with pytest.raises(ValueError):
        sb.load_entities(bad + [ents[1]])


@pytest.mark.parametrize("name", ["zones", "shapes_hull", "cloths", "terrain", "vehicle"])
def test_binary_round_trip_of_the_other_component_kinds(name):
    """Force-field entities (force + transform + colliders), hull colliders (geometry index behind the alignment gap), cloth
    components, the heightmap collider component (written, skipped on read) and a jointed vehicle: what is written reads back to
    the same bodies, colliders, fields and cloths; triggers are not part of the format."""
    from directx_renderer_kurth_amd import scenes, scene_binary as sb
    scene = scenes.by_name(name)
    ents = sb.unpack(sb.pack(sb.dump_entities(scene)))
    loaded, constraints, vel = sb.load_entities(ents, dt=scene.dt)
    assert not constraints and not vel.any()
    def f32(t):
        return tuple(tuple(np.float32(x) for x in v) if isinstance(v, tuple) else (v if isinstance(v, bool) else np.float32(v)) for v in t)
    assert [f32(b) for b in loaded.bodies] == [f32(b) for b in scene.bodies]      # (the stream holds fp32, the scene description Python floats)

    def cols(sc):
        return sorted((b, t, tuple(np.float32(x) for x in sh), tuple(np.float32(x) for x in m), tuple(np.float32(x) for x in p), tuple(np.float32(x) for x in r)) for b, t, sh, m, p, r in sc.colliders)
    assert cols(loaded) == cols(scene)
    assert len(loaded.fields) == len(scene.fields) and len(loaded.cloths) == len(scene.cloths)
    for (f0, p0, r0, c0), (f1, p1, r1, c1) in zip(scene.fields, loaded.fields):
        assert tuple(np.float32(x) for x in f0) == tuple(np.float32(x) for x in f1) and len(c0) == len(c1) and (p0 is None) == (p1 is None)
        assert [(t, tuple(np.float32(x) for x in sh)) for t, sh in c0] == [(t, tuple(np.float32(x) for x in sh)) for t, sh in c1]
    for a, b in zip(scene.cloths, loaded.cloths):
        assert tuple(np.float32(x) for x in a[:8]) == tuple(np.float32(x) for x in b[:8])
    if scene.heightmap is not None:
        assert any("heightmap_collider" in sb.read_entity(stream) for _, stream in ents)


@pytest.mark.gpu
def test_binary_streams_resume_a_running_world_on_the_device(mi):
    """A running world with every joint type written as entity streams (poses, velocities, constraints as their PODs) and read back
    into a fresh world through mi_add_body / mi_add_collider / mi_add_constraint: the copy continues bit-identically.  (The stream
    keeps no global constraint order — a reader gets them entity by entity — so the running world is itself built from a stream.)"""
    from directx_renderer_kurth_amd import scenes, scene_binary as sb
    scene = scenes.by_name("joints_mix")
    sizes = {"distance": (0, 28), "ball": (1, 24), "fixed": (2, 40), "hinge": (3, 104), "cone_twist": (4, 120), "slider": (5, 72)}
    w0 = scene.instantiate(mi.World())
    counts = {}
    for j in scene.joints:
        k = j[0][:-6] if j[0].endswith("_local") else j[0]
        counts[k] = counts.get(k, 0) + 1
    pods = {k: [bytes(w0.constraint_get(sizes[k][0], i, sizes[k][1])) for i in range(n)] for k, n in counts.items()}
    scene2, c0, vel0 = sb.load_entities(sb.dump_entities(scene, mass_properties=w0.mass_properties(), constraint_pods=pods), dt=scene.dt)
    assert len(c0) == len(scene.joints) and not vel0.any()

    def build(sc, cons):
        w = sc.instantiate(mi.World())
        for t, a, b, pod in cons:
            w.add_constraint(t, a, b, pod)
        return w
    g = build(scene2, c0)
    for _ in range(40):
        g.step_internal(scene.dt)
    index, live = {}, []
    for t, a, b, _ in c0:                               # the running world's PODs (accumulated motor / limit state included), in its add order
        k = index.get(t, 0); index[t] = k + 1
        live.append((t, a, b, bytes(g.constraint_get(t, k, sb.CONSTRAINT_BYTES[t]))))
    blob = sb.pack(sb.dump_entities(scene2, transforms=g.transforms(1), velocities=g.velocities(), mass_properties=g.mass_properties(), constraints=live))
    scene3, c1, vel = sb.load_entities(sb.unpack(blob), dt=scene.dt)
    assert [c[:3] for c in c1] == [c[:3] for c in c0] and np.array_equal(vel, g.velocities())
    h = build(scene3, c1)
    h.write_state(g.transforms(1), vel)
    g.snapshot()                                        # both worlds re-order their bodies at the next step (see mi_snapshot_save)
    for _ in range(40):
        g.step_internal(scene.dt); h.step_internal(scene.dt)
    assert np.array_equal(g.transforms(1), h.transforms(1)) and np.array_equal(g.velocities(), h.velocities())
