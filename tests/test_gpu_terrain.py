"""Heightmap terrain collision (row N4 of SURVEY §8f; reference heightmap_collider.h / heightmap_collision.cpp, physics.cpp:1236-1249) on the
GPU against the oracle in follow mode."""
import os

import numpy as np
import pytest

from parity_util import follow_step

pytestmark = pytest.mark.gpu


def test_terrain_follow_trajectory(mi, oracle):
    """300 bodies of every collider type on a rolling 2 x 2-chunk heightmap with a hole: per step the device's terrain contacts (one
    manifold slot each, emitted cell by cell) are exactly the contacts the oracle's mip-pyramid walk finds (same count per collider, same
    point / normal / depth), and the free-running trajectories stay within the tolerances of test_gpu_step_parity."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("terrain")
    g = scene.instantiate(mi.World())
    o = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_CUSTOM))
    worst = {}
    terrain_contacts = 0
    for i in range(240):
        r = follow_step(g, o, scene.dt, 30, None, resync=False)
        assert r["pairs_equal"], "step %d: broadphase pair set differs" % i
        assert r["counts_equal"], "step %d: contact counts differ" % i
        assert o.terrain_slot_mismatch() == 0, "step %d: terrain contact sets differ" % i
        assert r.get("contact_fr_equal", True)
        assert r["vel_err"] <= 1e-4 * max(1.0, r["vel_scale"]), "step %d: velocity error %g" % (i, r["vel_err"])
        for k in ("contact_point_err", "contact_depth_err", "contact_normal_err", "pos_err", "rot_err", "vel_err"):
            if k in r:
                worst[k] = max(worst.get(k, 0.0), r[k])
        slots = g.manifolds()[0]
        terrain_contacts = max(terrain_contacts, int((slots[:, 1] >= 0x80000000).sum()))
    print("terrain: worst", worst, "terrain contacts (max per step)", terrain_contacts, "contacts", r.get("num_contacts"))
    assert terrain_contacts > 100
    assert worst["pos_err"] <= 1e-3 and worst["rot_err"] <= 1e-3
    for k in ("contact_point_err", "contact_depth_err", "contact_normal_err"):
        assert worst.get(k, 0.0) <= 1e-5 * (1.0 + 100.0), k
    # supported shapes rest on the surface; cylinders and hulls fall through it, like in the reference
    t = g.transforms(1)
    kinds = np.arange(len(t)) % 10
    resting = t[kinds < 8]
    h = np.array([g.heightmap_height_at(float(p[0]), float(p[2])) for p in resting])
    on_terrain = h > -1e30
    assert (resting[on_terrain, 1] > h[on_terrain] - 0.3).all()
    assert (t[kinds >= 8, 1] < -3.0).mean() > 0.8          # (a few ride on supported bodies)


def test_single_body_on_terrain_without_any_pair(mi, oracle):
    """No broadphase pair at all: the step still collides the body with the terrain (the manifold slots start at 0)."""
    from directx_renderer_kurth_amd import scenes
    s = scenes.Scene("one_on_terrain", dt=1.0 / 120.0)
    s.heightmap = (1, 32.0, (0.1, 0.8, 1.0), (-16.0, 0.0, -16.0), 4.0, scenes.terrain_heights(1))
    b = s.add_body((1.0, 5.0, -2.0))
    s.add_collider(b, scenes.OBB, (0, 0, 0, 1, 0, 0, 0, 0.5, 0.3, 0.7), scenes.DEFAULT_MATERIAL)
    g = s.instantiate(mi.World()); o = s.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_CUSTOM))
    for i in range(240):
        r = follow_step(g, o, s.dt, 30, None, resync=False)
        assert r["counts_equal"] and o.terrain_slot_mismatch() == 0 and r["vel_err"] <= 1e-4 * max(1.0, r["vel_scale"]), "step %d" % i
    t = g.transforms(1)[0]
    h = g.heightmap_height_at(float(t[0]), float(t[2]))
    assert h - 0.05 < t[1] < h + 1.0 and np.abs(g.velocities()).max() < 0.5
    assert g.heightmap_height_at(1.0, -2.0) == o.heightmap_height_at(1.0, -2.0)


def test_terrain_travels_with_the_snapshot(mi):
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("terrain")
    g = scene.instantiate(mi.World())
    for _ in range(90):
        g.step_internal(scene.dt, 30)
    r = mi.World.restore(g.snapshot())
    for _ in range(30):
        g.step_internal(scene.dt, 30); r.step_internal(scene.dt, 30)
    assert np.array_equal(g.transforms(1), r.transforms(1)) and np.array_equal(g.velocities(), r.velocities())
    assert g.heightmap_height_at(-3.0, -5.0) == r.heightmap_height_at(-3.0, -5.0) > -1e30


def test_terrain_slot_budget_overflow_is_reported(mi):
    """More terrain contacts than the slot budget allows: contacts would be dropped, so the world fails with MI_ERR_CAPACITY at its next
    synchronisation instead of simulating on without them."""
    from directx_renderer_kurth_amd import scenes
    os.environ["MI_TERRAIN_SLOTS_PER_COLLIDER"] = "1"; os.environ["MI_TERRAIN_MIN_SLOTS"] = "2"
    try:
        s = scenes.Scene("terrain_overflow", dt=1.0 / 120.0)
        s.heightmap = (1, 32.0, (0.1, 0.8, 1.0), (-16.0, 0.0, -16.0), 4.0, scenes.terrain_heights(1))
        for k in range(4):
            b = s.add_body((2.0 * k - 3.0, 3.2, 0.5 * k))
            s.add_collider(b, scenes.OBB, (0, 0, 0, 1, 0, 0, 0, 0.9, 0.3, 0.9), scenes.DEFAULT_MATERIAL)
        g = s.instantiate(mi.World())
        with pytest.raises(mi.PhysicsError):
            for _ in range(240):
                g.step_internal(s.dt, 30)
    finally:
        del os.environ["MI_TERRAIN_SLOTS_PER_COLLIDER"]; del os.environ["MI_TERRAIN_MIN_SLOTS"]
