"""Force fields, triggers and collision begin / end events (row N2 of SURVEY §8f; reference physics.cpp:759-787, 952-1178) on the GPU,
through the C-ABI, against the CPU oracle in follow mode: same trajectories (the fields change the dynamics) and the same event records
in the same order, floats included."""
import os

import numpy as np
import pytest

from parity_util import follow_step

pytestmark = pytest.mark.gpu


def _events_equal(ge, oe):
    return len(ge) == len(oe) and ge.tobytes() == oe.tobytes()


def test_zones_follow_trajectory_and_events(mi, oracle):
    """`zones`: every collider type under sphere / capsule / cylinder / AABB / OBB / hull shaped fields and triggers (all 21 boolean
    overlap tests), two global fields, collision events on.  150 free-running steps: bit-equal events every step, trajectories within the
    stated tolerances of test_gpu_step_parity (in practice bit-equal)."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("zones")
    g = scene.instantiate(mi.World())
    o = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_CUSTOM))
    counts = np.zeros(4, np.int64)
    worst = 0.0
    for i in range(150):
        r = follow_step(g, o, scene.dt, 30, None, resync=False)
        assert r["pairs_equal"] and r["counts_equal"], "step %d" % i
        assert r["vel_err"] <= 1e-4 * max(1.0, r["vel_scale"]), "step %d: velocity error %g" % (i, r["vel_err"])
        worst = max(worst, r["pos_err"], r["vel_err"])
        ge, oe = g.drain_events(), o.drain_events()
        assert _events_equal(ge, oe), "step %d: %d device events vs %d oracle events" % (i, len(ge), len(oe))
        counts += np.bincount(ge["kind"], minlength=4)
    print("zones: events by kind", counts.tolist(), "worst pos/vel error", worst)
    assert (counts > 0).all()
    assert worst <= 1e-3
    tested, hit = o.zone_pair_stats()
    assert (tested[np.triu_indices(6)] > 0).all()


def test_events_accumulate_over_a_multi_step_frame(mi, oracle):
    """Events of several internal steps drained at once keep their step stamps and the per-step order (trigger events, then collision
    events, each by pair)."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("zones")
    g = scene.instantiate(mi.World())
    h = scene.instantiate(mi.World())
    per_step = []
    for _ in range(40):
        h.step_internal(scene.dt, 30)
        per_step.append(h.drain_events())
    for _ in range(40):
        g.step_internal(scene.dt, 30)
    ev = g.drain_events()
    assert _events_equal(ev, np.concatenate(per_step))
    assert len(ev) and (np.diff(ev["step"].astype(np.int64)) >= 0).all()
    assert len(g.drain_events()) == 0


def test_trigger_and_field_kat(mi):
    """Known answers on the device alone: a global field accelerates a free body by F/m*dt per step; a localized field only acts inside
    its collider; a sphere dropped through a trigger raises enter then leave, then a collision begin on the ground."""
    dt = 1.0 / 120.0
    inv_mass = 1.0 / (4.0 / 3.0 * np.pi * 0.125)
    w = mi.World()
    a = w.add_body((0, 10, 0), gravity_factor=0.0, linear_damping=0.0, angular_damping=0.0)
    w.add_collider(a, mi.SPHERE, (0, 0, 0, 0.5), (0.1, 0.5, 1.0))
    b = w.add_body((20, 10, 0), gravity_factor=0.0, linear_damping=0.0, angular_damping=0.0)
    w.add_collider(b, mi.SPHERE, (0, 0, 0, 0.5), (0.1, 0.5, 1.0))
    w.add_force_field((2.0, 0.0, 0.0))
    f = w.add_force_field((0.0, 5.0, 0.0), pos=(0.0, 10.0, 0.0))
    w.add_force_field_collider(f, mi.AABB, (-1, -1, -1, 1, 1, 1))
    w.step_internal(dt, 1)
    v = w.velocities()
    np.testing.assert_allclose(v[0, :3], [2.0 * inv_mass * dt, 5.0 * inv_mass * dt, 0.0], rtol=1e-5)
    np.testing.assert_allclose(v[1, :3], [2.0 * inv_mass * dt, 0.0, 0.0], rtol=1e-5, atol=1e-9)

    w = mi.World()
    s = w.add_body((0, 4, 0), linear_damping=0.0, angular_damping=0.0)
    w.add_collider(s, mi.SPHERE, (0, 0, 0, 0.5), (0.1, 0.5, 1.0))
    ground = w.add_static_collider(mi.AABB, (-10, -1, -10, 10, 0, 10), (0.1, 0.5, 1.0))
    t = w.add_trigger(pos=(0, 2, 0))
    w.add_trigger_collider(t, mi.AABB, (-1, -0.25, -1, 1, 0.25, 1))
    w.add_trigger_collider(t, mi.SPHERE, (0, 0, 0, 0.3))
    w.enable_collision_events()
    for _ in range(150):
        w.step_internal(dt, 30)
    ev = w.drain_events()
    assert ev["kind"][:3].tolist() == [mi.TRIGGER_ENTER, mi.TRIGGER_LEAVE, mi.COLLISION_BEGIN]
    assert ev["a"][0] == t and ev["b"][0] == s and ev["bodyA"][2] == s and ev["bodyB"][2] == mi.STATIC and ev["b"][2] == ground
    assert ev["relativeVelocity"][2][1] > 3.0 and abs(abs(ev["normal"][2][1]) - 1.0) < 1e-6


def test_snapshot_keeps_the_previous_overlap_sets(mi):
    """A world restored from a snapshot raises the same events as the original from there on (the previous step's overlap and collision
    sets travel with the snapshot)."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("zones")
    g = scene.instantiate(mi.World())
    for _ in range(60):
        g.step_internal(scene.dt, 30)
    g.drain_events()
    r = mi.World.restore(g.snapshot())
    for i in range(30):
        g.step_internal(scene.dt, 30); r.step_internal(scene.dt, 30)
        ge, re_ = g.drain_events(), r.drain_events()
        re_["step"] += 60  # the restored world counts its own steps
        assert _events_equal(ge, re_), "step %d" % i
    assert np.array_equal(g.transforms(1), r.transforms(1))


def test_event_ring_overflow_is_reported(mi):
    """A ring too small for a step's events loses events: the loss is reported as MI_ERR_CAPACITY, never silently."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("zones")
    os.environ["MI_EVENT_CAPACITY"] = "16"
    try:
        g = scene.instantiate(mi.World())
        for _ in range(60):
            g.step_internal(scene.dt, 30)
        g.drain_events()
        with pytest.raises(mi.PhysicsError):
            g.step_internal(scene.dt, 30)
    finally:
        del os.environ["MI_EVENT_CAPACITY"]


def test_moving_zones(mi, oracle):
    """Moving a trigger / force-field entity between steps: the colliders follow and the field's force turns with the rotation; events and
    trajectories keep matching the oracle bit for bit."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("zones")
    g = scene.instantiate(mi.World())
    o = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_CUSTOM))
    moved_events = 0
    for i in range(120):
        if i >= 40 and i % 4 == 0:
            a = 0.05 * (i - 40)
            q = (0.0, float(np.sin(a / 2)), 0.0, float(np.cos(a / 2)))
            for w in (g, o):
                w.set_trigger_transform(1, (2.0 * np.sin(a), 0.2, 2.0 * np.cos(a) - 2.0), q)
                w.set_force_field_transform(1, (-2.0 + a, 1.0, 0.0), q)
                w.set_force_field_transform(2, (0.0, 0.0, 0.0), q)       # a global field gains a rotating transform
        r = follow_step(g, o, scene.dt, 30, None, resync=False)
        assert r["pairs_equal"] and r["counts_equal"] and r["vel_err"] <= 1e-4 * max(1.0, r["vel_scale"]), "step %d" % i
        ge, oe = g.drain_events(), o.drain_events()
        assert _events_equal(ge, oe), "step %d: %d device events vs %d oracle events" % (i, len(ge), len(oe))
        if i >= 40:
            moved_events += int((ge["kind"] < 2).sum())
    assert moved_events > 10
