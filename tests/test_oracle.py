"""CPU tests of the oracle (the CPU restatement of the reference's src/physics path): committed golden fixtures, analytic
known-answer tests, scalar-vs-8-lane consistency, and the batch scheduler's invariants.  No GPU needed."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="module")
def golden_tools():
    import sys
    sys.path.insert(0, GOLDEN)
    import make_golden
    return make_golden


@pytest.mark.parametrize("fixture,num_type_pairs", [("narrow_pairs.npz", 10), ("narrow_pairs_cylinder.npz", 5), ("narrow_pairs_hull.npz", 6)])
def test_narrow_pairs_golden(oracle, golden_tools, fixture, num_type_pairs):
    """Every intersection() pair on 48-64 seeded poses (incl. coincident centres, parallel capsules / cylinders, SAT `parallel`):
    the 10 sphere/capsule/box pairs, the 5 with a cylinder, the 6 with a convex hull."""
    g = load(fixture)
    s = golden_tools.scene_from_arrays(g["bodies"], g["colliders"], hull_vertices=g["hull_vertices"], hull_triangles=g["hull_triangles"], hull_ranges=g["hull_ranges"])
    w = s.instantiate(oracle.OracleWorld())
    w.step_internal(1e-9, 1)
    cols, aabbs = w.world_colliders()
    assert np.array_equal(cols.view(np.uint8).reshape(len(cols), 64), g["world_colliders"])
    assert np.array_equal(aabbs, g["aabbs"])
    assert np.array_equal(w.pairs(), g["pairs"])
    cpairs, counts = w.collisions()
    assert np.array_equal(cpairs, g["colliding_pairs"]) and np.array_equal(counts, g["counts"])
    contacts = w.contacts()[0]
    assert np.array_equal(contacts.view(np.uint8).reshape(len(contacts), 32), g["contacts"])
    np.testing.assert_array_equal(w.mass_properties(), g["mass"])
    # all type pairs of the fixture are exercised and produce contacts
    types = cols["type"]
    seen = {(int(types[a]), int(types[b])) for a, b in cpairs}
    assert len(seen) == num_type_pairs, seen


def test_narrow_contact_invariants(oracle):
    g = load("narrow_pairs.npz")
    c = g["contacts"].view(oracle.CONTACT_DTYPE).reshape(-1)
    n = np.linalg.norm(c["normal"], axis=1)
    assert np.all(np.abs(n - 1.0) < 1e-4)          # unit normals
    cols = g["world_colliders"].view(oracle.COLLIDER_DTYPE).reshape(-1)
    pair_of_contact = g["colliding_pairs"][g["contact_collision"]]
    aabb_aabb = (cols["type"][pair_of_contact[:, 0]] == 3) & (cols["type"][pair_of_contact[:, 1]] == 3)
    # penetration depth is positive (collision_narrow.cpp:395) — except for aabb-vs-aabb, where the reference multiplies the depth by
    # the sign of the separation (collision_narrow.cpp:1092-1093, a latent reference quirk we reproduce: negative when B is on the -axis side)
    assert np.all(c["depth"][~aabb_aabb] >= -1e-6)
    assert np.any(c["depth"][aabb_aabb] < 0)
    assert np.all(g["counts"] >= 1) and np.all(g["counts"] <= 4)


def test_sphere_sphere_closed_form(oracle):
    """KAT: depth = r1 + r2 - d, normal = (c2 - c1)/d, point = midpoint of the surface points (collision_narrow.cpp:374-400)."""
    cols = np.zeros(3, oracle.COLLIDER_DTYPE)
    cols["type"] = 0
    cols["shape"][0, :4] = (0, 0, 0, 0.5); cols["shape"][1, :4] = (0.8, 0, 0, 0.4); cols["shape"][2, :4] = (0, 0, 0, 0.25)
    cols["friction"] = 0.5; cols["restitution"] = 0.1
    contacts, counts = oracle.narrowphase_ordered(cols, [[0, 1], [0, 2]])
    assert list(counts) == [1, 1]
    np.testing.assert_allclose(contacts["depth"], [0.1, 0.75], atol=1e-6)
    np.testing.assert_allclose(contacts["normal"][0], [1, 0, 0], atol=1e-7)
    np.testing.assert_allclose(contacts["normal"][1], [0, 1, 0], atol=0)      # coincident centres -> +Y
    np.testing.assert_allclose(contacts["point"][0], [0.45, 0, 0], atol=1e-6)
    fr = int(contacts["friction_restitution"][0])
    assert fr >> 16 == int(np.float32(0.5) * 0xFFFF) and fr & 0xFFFF == int(np.float32(0.1) * 0xFFFF)


def test_scheduler_golden_and_invariants(oracle):
    g = load("scheduler.npz")
    for name in ("small", "chain", "dense", "ground"):
        bp, dummy, slots = g[name + "_pairs"], int(g[name + "_dummy"]), g[name + "_slots"]
        out = oracle.schedule(bp, dummy)
        assert np.array_equal(out, slots), name
        # every constraint scheduled exactly once (padding lanes repeat lane 0)
        seen = set()
        for row in out:
            lanes = [int(row[0])] + [int(x) for x in row[1:] if int(x) != int(row[0])]
            bodies = []
            for ci in lanes:
                assert ci not in seen
                seen.add(ci)
                bodies += [int(b) for b in bp[ci] if int(b) != dummy]
            assert len(bodies) == len(set(bodies)), "lanes of a batch must not share a dynamic body"
        assert seen == set(range(len(bp)))


def test_c1_trajectory_golden(oracle, golden_tools):
    g = load("c1_trajectory.npz")
    s = golden_tools.scene_from_arrays(g["bodies"], g["colliders"])
    for mode_name, mode in (("scalar", oracle.SOLVER_SCALAR), ("wide8", oracle.SOLVER_WIDE8)):
        w = s.instantiate(oracle.OracleWorld(solver=mode))
        step = 0
        for cp in (1, 60, 120, 240):
            while step < cp:
                w.step_internal(s.dt); step += 1
            assert np.array_equal(w.transforms(1), g["%s_t%d" % (mode_name, cp)]), (mode_name, cp)
            assert np.array_equal(w.velocities(), g["%s_v%d" % (mode_name, cp)])


def test_scalar_vs_wide_first_step(oracle, golden_tools):
    """The reference's own A/B check (editor toggle simdConstraintSolver): same maths, different Gauss-Seidel order.  With no
    contacts yet (step 1 of the ragdoll) both orders must agree to rounding; with contacts they agree at solver-convergence level."""
    g = load("ragdoll_trajectory.npz")
    np.testing.assert_allclose(g["scalar_t1"], g["wide8_t1"], atol=2e-5)
    np.testing.assert_allclose(g["scalar_v1"], g["wide8_v1"], atol=2e-3)
    c1 = load("c1_trajectory.npz")
    # after 240 steps both settle on the ground: same resting heights within 5 cm, nothing tunnels through the ground
    assert c1["scalar_t240"][:, 1].min() > 0.2 and c1["wide8_t240"][:, 1].min() > 0.2


def test_ragdoll_trajectory_golden(oracle, golden_tools):
    g = load("ragdoll_trajectory.npz")
    s = golden_tools.scene_from_arrays(g["bodies"], g["colliders"], dt=1.0 / 60.0)
    from directx_renderer_kurth_amd import scenes
    ref = scenes.c4_ragdolls(1)
    s.joints = ref.joints
    w = s.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_SCALAR))
    step = 0
    for cp in (1, 30, 120):
        while step < cp:
            w.step_internal(s.dt); step += 1
        np.testing.assert_allclose(w.transforms(1), g["scalar_t%d" % cp], atol=1e-6)
    # joints hold: anchor separation of the 13 joints stays small while the ragdoll collapses
    t = w.transforms(1)
    assert np.isfinite(t).all() and t[:, 1].min() > -0.05 and t[:, 1].max() < 1.5


def test_free_fall_and_mass_kat(oracle):
    g = load("kat.npz")
    dt, v = 1.0 / 120.0, 0.0
    for expected in g["free_fall_vy"]:
        v = (v - 9.81 * dt) / (1.0 + dt * 0.4)     # rigid_body.cpp:110-118
        assert abs(v - float(expected)) < 2e-6
    m = g["box_mass"][0]
    mass = 1.0 * 2.0 * 3.0 * 2.0                      # volume * density
    assert abs(m[3] - 1.0 / mass) < 1e-7
    inertia = np.array([mass / 12 * (2 ** 2 + 3 ** 2), mass / 12 * (1 ** 2 + 3 ** 2), mass / 12 * (1 ** 2 + 2 ** 2)])  # physics.cpp:1498-1502
    np.testing.assert_allclose([m[4], m[8], m[12]], 1.0 / inertia, rtol=1e-6)


def test_sphere_rests_on_ground_within_slop(oracle):
    """A sphere dropped from 1 cm comes to rest with penetration inside the solver's slop (0.001, constraints.cpp:3360)."""
    from directx_renderer_kurth_amd import scenes
    s = scenes.Scene("rest")
    scenes._ground(s, 10.0)
    b = s.add_body((0, 0.51, 0))
    s.add_collider(b, scenes.SPHERE, (0, 0, 0, 0.5), scenes.DEFAULT_MATERIAL)
    w = s.instantiate(oracle.OracleWorld())
    for _ in range(240):
        w.step_internal(s.dt)
    y = w.transforms(1)[0, 1]
    assert 0.5 - 0.002 <= y <= 0.5 + 1e-4, y
    assert np.abs(w.velocities()).max() < 1e-2


def test_physics_step_accumulator(oracle):
    """physicsStep (physics.cpp:1364-1413): fixed 1/120 s sub-steps, at most 4 per frame, excess time dropped, lerp in between."""
    from directx_renderer_kurth_amd import scenes
    s = scenes.c1_boxes(4)
    w = s.instantiate(oracle.OracleWorld())
    st = oracle.Settings()
    w.step(0.004, st)
    assert abs(w.timer.value - 0.004) < 1e-9 and np.array_equal(w.transforms(1), w.transforms(2))  # no sub-step yet
    w.step(0.005, st)                                                       # 0.009 >= 1/120 -> one sub-step
    assert abs(w.timer.value - (0.009 - 1 / 120.0)) < 1e-7
    t0, t1, ti = w.transforms(2), w.transforms(1), w.transforms(0)
    a = w.timer.value * 120.0
    np.testing.assert_allclose(ti[:, :3], t0[:, :3] + a * (t1[:, :3] - t0[:, :3]), atol=1e-6)
    w.step(1.0, st)                                                         # 4 sub-steps max, remainder dropped by fmod
    assert w.timer.value < 1 / 120.0


def test_hull_mass_properties_and_contact(oracle):
    """Convex hulls: the tetrahedron-covariance mass properties of a box-shaped hull equal the closed-form box
    (physics.cpp:1520-1580 vs :1496-1502), and sphere-vs-hull GJK + EPA gives the face normal and the penetration depth (within EPA's
    own 0.01 stop criterion, collision_epa.h:139)."""
    from directx_renderer_kurth_amd import scenes
    w = oracle.OracleWorld()
    g = w.add_hull_geometry(*scenes.hull_box(0.5, 0.35, 0.45))
    hb = w.add_body((0, 0, 0))
    w.add_collider(hb, 5, (0, 0, 0, 1, 0, 0, 0, float(g)), (0.1, 0.5, 2.0))
    bb = w.add_body((3, 0, 0))
    w.add_collider(bb, 4, (0, 0, 0, 1, 0, 0, 0, 0.5, 0.35, 0.45), (0.1, 0.5, 2.0))
    mp = w.mass_properties()
    np.testing.assert_allclose(mp[hb], mp[bb], rtol=2e-5, atol=1e-6)          # {localCOG, invMass, invInertia}
    assert abs(1.0 / mp[hb][3] - 2.0 * 1.0 * 0.7 * 0.9) < 1e-5
    sb = w.add_body((0, 0.35 + 0.3 - 0.05, 0))                                # sphere r = 0.3 resting 5 cm inside the top face
    w.add_collider(sb, 0, (0, 0, 0, 0.3), (0.1, 0.5, 1.0))
    w.step_internal(1e-9, 1)
    contacts, bp, ci = w.contacts()
    assert len(contacts) == 1
    n = contacts["normal"][0]; d = contacts["depth"][0]
    assert abs(abs(n[1]) - 1.0) < 1e-3 and abs(d - 0.05) < 0.011


def test_overlap_checks_golden_and_consistency(oracle, golden_tools):
    """The boolean overlap tests behind force fields and triggers (overlapCheck, collision_narrow.cpp:1593-1689) on the poses of the three
    narrowphase fixtures: pinned flags, and agreement with the contact-generating tests of the same pairs.  The two families are separate
    code in the reference; they must agree except where the reference itself differs: sphereVsCylinder compares a squared distance with
    the plain radius (bounding_volumes.cpp:723), which only ever adds overlaps for r < 1, and cylinder-cylinder has no parallel closed form."""
    g = load("overlap_pairs.npz")
    for name in ("narrow_pairs", "narrow_pairs_cylinder", "narrow_pairs_hull"):
        pairs, flags, cols = golden_tools.overlap_flags(name + ".npz")
        assert np.array_equal(pairs, g[name + "_pairs"]) and np.array_equal(flags, g[name + "_overlaps"])
        colliding = {(int(a), int(b)) for a, b in load(name + ".npz")["colliding_pairs"]}
        collide = np.array([(int(a), int(b)) in colliding for a, b in pairs])
        cyl = (cols["type"][pairs[:, 0]] == oracle.CYLINDER) | (cols["type"][pairs[:, 1]] == oracle.CYLINDER)
        assert np.array_equal(collide[~cyl], flags[~cyl])
        assert not (collide[cyl] & ~flags[cyl]).any() and (not cyl.any() or (collide[cyl] == flags[cyl]).mean() > 0.9)
        assert flags.any() and not flags.all()


def _one_sphere_world(oracle, pos, gravity_factor=0.0):
    w = oracle.OracleWorld()
    b = w.add_body(pos, gravity_factor=gravity_factor, linear_damping=0.0, angular_damping=0.0)
    w.add_collider(b, oracle.SPHERE, (0, 0, 0, 0.5), (0.1, 0.5, 1.0))
    return w, b


def test_force_field_kat(oracle):
    """Global field: every body gets F each step (physics.cpp:1273); localized field: only while the body's collider overlaps the field's
    (:963-967); a field entity's rotation turns the force (:767-771).  v = F / m * dt with m = 4/3 pi r^3 (density 1), no damping."""
    dt = np.float32(1.0 / 120.0)
    inv_mass = np.float32(1.0) / np.float32(4.0 / 3.0 * np.pi * 0.125)
    w, b = _one_sphere_world(oracle, (0, 10, 0))
    w.add_force_field((2.0, 0.0, 0.0))
    w.add_force_field((0.0, 0.0, 1.0), rot=(0.0, 0.70710678, 0.0, 0.70710678))   # +z turned by 90 deg about y -> +x
    w.step_internal(float(dt), 1)
    v = w.velocities()[0, :3]
    np.testing.assert_allclose(v, [3.0 * inv_mass * dt, 0.0, 0.0], rtol=1e-6, atol=1e-7)

    w, b = _one_sphere_world(oracle, (0, 10, 0))
    f = w.add_force_field((0.0, 5.0, 0.0), pos=(0.0, 10.0, 0.0))
    w.add_force_field_collider(f, oracle.AABB, (-1, -1, -1, 1, 1, 1))
    far = w.add_body((20, 10, 0), gravity_factor=0.0, linear_damping=0.0, angular_damping=0.0)
    w.add_collider(far, oracle.SPHERE, (0, 0, 0, 0.5), (0.1, 0.5, 1.0))
    w.step_internal(float(dt), 1)
    v = w.velocities()
    np.testing.assert_allclose(v[0, :3], [0.0, 5.0 * inv_mass * dt, 0.0], rtol=1e-6)
    assert not v[1].any()
    w.set_force_field(f, (0.0, 0.0, 0.0))
    w.step_internal(float(dt), 1)
    np.testing.assert_allclose(w.velocities()[0, :3], [0.0, 5.0 * inv_mass * dt, 0.0], rtol=1e-6)


def test_trigger_and_collision_events_kat(oracle):
    """A sphere dropped through a trigger box onto the ground: enter, leave, then a collision begin whose record carries the contact point
    under the sphere, the normal along y and the approach velocity; lifted away afterwards: collision end.  Events come out per step in the
    reference's callback order (physics.cpp:1000-1032, 1128-1174)."""
    w, b = _one_sphere_world(oracle, (0, 4, 0), gravity_factor=1.0)
    ground = w.add_static_collider(oracle.AABB, (-10, -1, -10, 10, 0, 10), (0.1, 0.5, 1.0))
    t = w.add_trigger(pos=(0, 2, 0))
    w.add_trigger_collider(t, oracle.AABB, (-1, -0.25, -1, 1, 0.25, 1))
    w.add_trigger_collider(t, oracle.SPHERE, (0, 0, 0, 0.3))          # overlaps the box: still one event per body
    w.enable_collision_events()
    events = []
    for _ in range(150):
        w.step_internal(1.0 / 120.0, 30)
        events.extend(w.drain_events())
    kinds = [int(e["kind"]) for e in events]
    assert kinds[:3] == [oracle.TRIGGER_ENTER, oracle.TRIGGER_LEAVE, oracle.COLLISION_BEGIN], kinds
    enter, leave, begin = events[:3]
    assert enter["a"] == t and enter["b"] == b and leave["a"] == t and leave["b"] == b and enter["step"] < leave["step"] < begin["step"]
    assert begin["bodyA"] == b and begin["bodyB"] == oracle.STATIC and begin["b"] == ground
    assert abs(begin["position"][1]) < 0.05 and abs(abs(begin["normal"][1]) - 1.0) < 1e-6
    assert begin["relativeVelocity"][1] > 3.0      # B (static) minus A (falling sphere)
    w.set_velocity(b, (0, 6.0, 0), (0, 0, 0))
    tail = []
    for _ in range(220):
        w.step_internal(1.0 / 120.0, 30)
        tail.extend(w.drain_events())
    assert int(tail[0]["kind"]) == oracle.COLLISION_END
    assert [int(e["kind"]) for e in tail if e["kind"] < 2][:2] == [oracle.TRIGGER_ENTER, oracle.TRIGGER_LEAVE]


def test_zones_scene_reaches_every_overlap_pair(oracle):
    """The `zones` parity scene exercises all 21 boolean type pairs, hits and misses, and raises all four event kinds."""
    from directx_renderer_kurth_amd import scenes
    s = scenes.by_name("zones")
    w = s.instantiate(oracle.OracleWorld())
    kinds = set()
    for _ in range(150):
        w.step_internal(s.dt, 8)
        kinds.update(int(k) for k in w.drain_events()["kind"])
    tested, hit = w.zone_pair_stats()
    iu = np.triu_indices(6)
    assert (tested[iu] > 0).all(), tested
    miss = tested - hit
    miss[3, 3] = 1  # two world-space AABBs that pass the broadphase overlap by definition
    assert (hit[iu] > 0).all() and (miss[iu] > 0).all(), (tested, hit)
    assert kinds == {0, 1, 2, 3}


def test_cloth_oracle_orders_and_invariants(oracle):
    """The cloth restatement (cloth.cpp): the reference's storage order and the device's colour order are both Gauss-Seidel sweeps over the
    same constraints — each colour is conflict-free, both keep the locked row and hold the stretch constraints' rest lengths within the
    solver's slack while the cloth swings down from its horizontal start, and the two trajectories stay close."""
    res = {}
    for colour in (False, True):
        w = oracle.OracleWorld()
        b = w.add_body((0, 100, 0)); w.add_collider(b, oracle.SPHERE, (0, 0, 0, 0.5), (0.1, 0.5, 1.0))   # the step needs one rigid body (physics.cpp:1184)
        c = w.add_cloth(10.0, 10.0, 20, 20, 8.0)
        w.cloth_set_fixed_vertices(c, (0, 12, 0), (0, 0, 0, 1), True)
        w.set_cloth_iterations(1, 4, 1)
        w.set_cloth_colour_order(colour)
        top = w.cloth_state(c)[0][:20].copy()
        for _ in range(240):
            w.step_internal(1.0 / 120.0, 1)
        p, v = w.cloth_state(c)
        cons, col = w.cloth_constraints(c)
        assert np.array_equal(p[:20], top)
        stretch = cons[col < 4]
        length = np.linalg.norm(p[stretch["a"]] - p[stretch["b"]], axis=1)
        assert np.abs(length / stretch["restDistance"] - 1.0).max() < 0.15
        assert np.isfinite(v).all() and 1.0 < np.abs(v).max() < 30.0
        res[colour] = p
        for k in range(12):
            ids = np.concatenate([cons["a"][col == k], cons["b"][col == k]])
            assert len(ids) == len(np.unique(ids))
    assert np.abs(res[True] - res[False]).max() < 0.25
    assert res[True][:, 1].min() < 6.0                     # the free edge has swung down from the bar at y = 12


def test_terrain_oracle_pyramid_equals_cell_sweep_and_kats(oracle):
    """The heightmap restatement: getHeightAt reproduces the vertices and interpolates between them; outside / missing chunks give -FLT_MAX;
    a sphere resting on a flat terrain touches it with a +-y normal at the right depth; bodies come to rest on a rolling terrain."""
    from directx_renderer_kurth_amd import scenes
    w = oracle.OracleWorld()
    flat = np.full((129, 129), 32768, np.uint16)
    w.set_heightmap(1, 64.0, (0.1, 0.8, 1.0), (-32.0, -1.0, -32.0), 2.0)
    w.heightmap_set_chunk(0, 0, flat)
    level = -1.0 + 32768 / 65535 * 2.0
    assert abs(w.heightmap_height_at(3.3, -7.1) - level) < 1e-6
    assert w.heightmap_height_at(40.0, 0.0) < -1e30 and w.heightmap_height_at(0.0, -32.5) < -1e30
    b = w.add_body((0.2, level + 0.45, 0.3), gravity_factor=0.0)
    w.add_collider(b, oracle.SPHERE, (0, 0, 0, 0.5), (0.1, 0.5, 1.0))
    w.step_internal(1e-6, 1)
    c = w.contacts()[0]
    flat_hits = np.isclose(np.abs(c["normal"][:, 1]), 1.0)     # triangles under the centre; neighbours are touched at their edges (not de-duplicated, like the reference)
    assert flat_hits.any() and np.allclose(c["depth"][flat_hits], 0.05, atol=1e-5) and (c["depth"] <= 0.05 + 1e-5).all()
    assert np.allclose(c["point"][:-1, 1], level, atol=1e-5)
    assert np.allclose(c["point"][-1], (0.2, level - 0.05, 0.3), atol=1e-5) and np.allclose(c["normal"][-1], (0, -1, 0))   # the "lowest point under the terrain" contact comes last

    s = scenes.by_name("terrain")
    w = s.instantiate(oracle.OracleWorld())
    h = s.heightmap[5][(0, 0)]
    x0, z0 = -24.0, -24.0
    for (i, j) in ((0, 0), (5, 17), (100, 64), (127, 127)):
        expect = -2.0 + float(h[i, j]) / 65535 * 6.0
        assert abs(w.heightmap_height_at(x0 + j * 24.0 / 128 + 1e-4, z0 + i * 24.0 / 128 + 1e-4) - expect) < 2e-3
    for _ in range(360):
        w.step_internal(s.dt, 30)
    t = w.transforms(1); v = w.velocities()
    kinds = np.arange(len(t)) % 10
    assert np.isfinite(t).all() and (t[kinds >= 8, 1] < -5.0).mean() > 0.8            # cylinders / hulls: no terrain case (heightmap_collision.cpp:545-570)
    rest = t[kinds < 8]
    hh = np.array([w.heightmap_height_at(float(p[0]), float(p[2])) for p in rest])
    ok = hh > -1e30
    assert ok.sum() > 150 and (rest[ok, 1] > hh[ok] - 0.3).all() and np.abs(v[kinds < 8][ok]).max() < 8.0


def _first_contact_deltas(a, b, scene, steps, window=10):
    """Step two oracle worlds side by side; relative velocity difference over the first `window` steps that have contacts."""
    deltas = []
    for i in range(steps):
        a.step_internal(scene.dt); b.step_internal(scene.dt)
        if len(a.contacts()[0]) > 0 and len(deltas) < window:
            va, vb = a.velocities(), b.velocities()
            deltas.append(float(np.abs(va - vb).max() / max(1.0, np.abs(va).max())))
    return deltas


@pytest.mark.parametrize("name,steps", [("c1", 120), ("c2_small", 60)])
def test_row_form_against_reference_formula(oracle, name, steps):
    """The device evaluates a contact row in Jacobian form with fused multiply-adds (csrc/solver_rows.h; restated in
    oconstraints.h: solveCollisionConstraintRowForm); the reference forms anchor velocities with cross products
    (constraints.cpp:3381-3449).  Same mathematics, different rounding.  Both in the reference's emission order, from the same
    start: velocities agree to 1e-5 relative over the first ten steps with contacts (rounding-level differences, before a tumbling
    pile amplifies them: config 1 is chaotic, two runs of it that differ in the last bit are metres apart after a second), and on
    the sphere lattice, which is not chaotic, positions agree to 1e-3 m after the whole run."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name(name)
    a = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_SCALAR))
    b = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_SCALAR)); b.set_scalar_row_form(True)
    deltas = _first_contact_deltas(a, b, scene, steps)
    dpos = float(np.abs(a.transforms(1)[:, :3] - b.transforms(1)[:, :3]).max())
    print(name, "row form vs reference formula: relative velocity difference over the first contact steps", ["%.1e" % d for d in deltas], "| positions after %d steps: %.2e m" % (steps, dpos))
    # The first two steps with contacts start from (almost) identical states: rounding-level agreement.  Afterwards the runs are two
    # different trajectories of a system with thresholds (a contact whose depth sits at the 1 mm slop gets its bias or not,
    # constraints.cpp:3360; a tangent shorter than 1e-4 is dropped, math.h:595): the sphere lattice with its 1 mm jitter has
    # hundreds of contacts on that edge, so its differences grow to 1e-2 relative within ten steps — and still end 2.5 mm apart.
    assert len(deltas) >= 5 and max(deltas[:2]) <= 1e-6
    if name == "c2_small":
        assert dpos <= 1e-2


@pytest.mark.parametrize("name,steps", [("c1", 120), ("c2_small", 60)])
def test_replay_order_against_the_wide_solver(oracle, name, steps):
    """SOLVER_REPLAY (what the device's replay mode is compared with bit for bit: the reference's batch order, the device's row
    arithmetic) against SOLVER_WIDE8 (the same batch order with the reference's 8-wide arithmetic, constraints.cpp:3618-3709): same
    order, different rounding — rounding-level agreement over the first steps with contacts."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name(name)
    a = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_WIDE8))
    b = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_REPLAY))
    deltas = _first_contact_deltas(a, b, scene, steps)
    print(name, "replay order (row arithmetic) vs the 8-wide solver: relative velocity difference over the first contact steps", ["%.1e" % d for d in deltas])
    assert len(deltas) >= 5 and max(deltas[:2]) <= 1e-6


def test_avx2_semantics_delta(oracle):
    """The reference's 8-wide path normalises the friction direction with _mm256_rsqrt_ps (12-bit estimate, no Newton step:
    math_simd.h:283-289); the oracle's 8-wide path uses exact 1/sqrt unless asked otherwise.  The distance between the two is the
    'AVX2 semantics' delta SURVEY section 8(c) promised to report: relative velocity difference over the first ten contact steps
    (expected <= 4e-4: the estimate's 1.5 * 2^-12 error scales the tangent direction only)."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("c1")
    exact = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_WIDE8))
    est = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_WIDE8))
    deltas = []
    try:
        for i in range(240):
            exact.set_wide_rsqrt(False); exact.step_internal(scene.dt)
            est.set_wide_rsqrt(True); est.step_internal(scene.dt)
            ve, vr = exact.velocities(), est.velocities()
            d = float(np.abs(ve - vr).max() / max(1.0, np.abs(ve).max()))
            if (d > 0.0 or deltas) and len(deltas) < 5:   # from the first step in which a friction direction is normalised at all
                deltas.append(d)
    finally:
        exact.set_wide_rsqrt(False)
    print("AVX2-semantics delta (rsqrt estimate vs exact in noz), relative velocity difference from the first step it shows:", ["%.1e" % d for d in deltas])
    assert deltas and 0.0 < deltas[0] <= 4e-4


def test_polynomial_trig_of_the_wide_paths(oracle):
    """The reference's wide paths evaluate cos / sin / atan2 / acos with short polynomials (core/simd.h:28-49, 122-164).  Their
    restatement against libm: the known shapes of the approximations (exact at the quadrant points, errors of the published size),
    so that a slip in a coefficient or in the quadrant logic shows."""
    w = oracle.OracleWorld()
    xs = np.linspace(-7.0, 7.0, 2801, dtype=np.float32)
    cos_err = max(abs(w.poly_trig(0, float(x)) - np.cos(np.float64(x))) for x in xs)
    sin_err = max(abs(w.poly_trig(1, float(x)) - np.sin(np.float64(x))) for x in xs)
    assert 2e-4 < cos_err < 1.2e-3 and 2e-4 < sin_err < 1.2e-3, (cos_err, sin_err)      # the parabola pair with the 0.225 correction: ~1e-3
    assert abs(w.poly_trig(0, 0.0) - 1.0) < 1e-6 and abs(w.poly_trig(0, float(np.float32(np.pi))) + 1.0) < 1e-5 and abs(w.poly_trig(1, 0.0)) < 1e-6
    angles = np.linspace(-np.pi, np.pi, 1441)[1:-1]
    atan_err = max(abs(w.poly_trig(2, float(np.float32(np.sin(a))), float(np.float32(np.cos(a)))) - a) for a in angles)
    assert 5e-4 < atan_err < 3e-3, atan_err                                                # rational first-quadrant fit, b = 0.596227
    for y, x, want in ((0.0, 1.0, 0.0), (1.0, 0.0, np.pi / 2), (-1.0, 0.0, -np.pi / 2), (1.0, 1.0, np.pi / 4), (1.0, -1.0, 3 * np.pi / 4), (-1.0, -1.0, -3 * np.pi / 4)):
        assert abs(w.poly_trig(2, y, x) - want) < 2e-6, (y, x, w.poly_trig(2, y, x), want)   # exact on the axes and diagonals
    cs = np.linspace(-1.0, 1.0, 2001, dtype=np.float32)
    acos_err = max(abs(w.poly_trig(3, float(c)) - np.arccos(np.float64(c))) for c in cs)
    assert 1e-5 < acos_err < 1e-4, acos_err                                                # the cubic * sqrt(1 - x) form: ~7e-5
    print("polynomial trig of the wide paths against libm: cos %.1e, sin %.1e, atan2 %.1e rad, acos %.1e rad" % (cos_err, sin_err, atan_err, acos_err))


def test_avx2_semantics_delta_of_the_joints(oracle):
    """Row a33 for the joints: the reference's 8-wide hinge and cone-twist initialisation (constraints.cpp:1309-1777, 2072-2634)
    computes the hinge angle, the twist angle, the swing rotation and the motor targets with the polynomial functions above and
    rsqrt-based normalisation; its scalar path calls libm.  BASELINE config 4 (256 ragdolls) in the oracle's 8-lane mode (joints in
    the scheduler's batch order, contacts 8 wide), once with each, per step from identical inputs: the distance between the two is
    the joints' "AVX2 semantics" delta.  The angles only matter where a limit is violated or a motor runs: the delta is exactly zero
    while the ragdolls fall freely, ~7.5e-3 rad/s once limits engage (an angle error of up to ~3e-3 rad in a violated limit changes
    its bias velocity by 3e-3 * beta / dt ~ 2e-2), and a handful of bodies per step differ by whole rad/s: their limit is just
    violated with one angle and just not with the other, so one side solves the limit row and the other does not."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("c4")
    exact = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_WIDE8))
    wide = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_WIDE8))
    first, typical, flips, zero_steps = None, [], 0, 0
    try:
        for i in range(40):
            exact.set_wide_joint_math(False); exact.step_internal(scene.dt)
            wide.set_wide_joint_math(True); wide.step_internal(scene.dt)
            ve, vw = exact.velocities(), wide.velocities()
            d = np.abs(ve - vw).max(axis=1)
            if d.max() == 0.0:
                zero_steps += 1
            else:
                if first is None:
                    first = (i, float(d.max()))
                typical.append(float(np.median(d[d > 0])))
                flips = max(flips, int((d > 0.1).sum()))
            wide.write_state(exact.transforms(1), ve)          # per-step delta: both continue from the exact world's state
    finally:
        exact.set_wide_joint_math(False)
    print("AVX2-semantics delta of the joints on config 4 (3 584 bodies), per step from identical inputs: zero in %d of 40 steps; first non-zero at step %d: %.1e rad/s; "
          "median body delta in the other steps %.1e ... %.1e; at most %d bodies per step beyond 0.1 (a limit engaged on one side only)"
          % (zero_steps, first[0], first[1], min(typical), max(typical), flips))
    assert first is not None and 1e-4 < first[1] < 5e-2 and max(typical) < 2e-2 and flips < 0.03 * exact.velocities().shape[0] and zero_steps > 5


def test_oracle_under_sanitizers():
    """The CPU restatement under AddressSanitizer + UndefinedBehaviourSanitizer (oracle/san_driver.cpp: mixed shapes, three joint
    kinds, the three solver modes, 90 steps each).  GPU sanitizers are not available on the MI355X pool; the oracle shares the
    algorithms' index arithmetic with the kernels, so this is where out-of-bounds logic would show."""
    import os, shutil, subprocess
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-C", root, "sanitize"], capture_output=True, text=True, timeout=900)
    if r.returncode != 0 and ("cannot find -lasan" in r.stderr or "libasan" in r.stderr and "No such file" in r.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("mode ") == 3


@pytest.mark.parametrize("name,steps", [("c1", 60), ("c3_small", 40), ("shapes", 30)])
def test_wide_sweep_reports_the_scalar_sweeps_pairs(oracle, name, steps):
    """determineOverlapsSIMD (collision_broad.cpp:168-295: the active boxes in SoA blocks of eight, one start endpoint against eight of
    them per compare) against determineOverlapsScalar (:87-166): the same pairs in the same order, step after step (the reference
    switches between them with physics_settings::simdBroadPhase; the cpu_baseline of bench.py times the wide one)."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name(name)
    a = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_SCALAR))
    b = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_SCALAR)); b.set_wide_broadphase(True)
    most = 0
    for i in range(steps):
        a.step_internal(scene.dt); b.step_internal(scene.dt)
        pa, pb = a.pairs(), b.pairs()
        assert np.array_equal(pa, pb), "step %d: the wide sweep's pair list differs" % i
        most = max(most, len(pa))
    assert most > 0 and np.array_equal(a.transforms(1), b.transforms(1))

