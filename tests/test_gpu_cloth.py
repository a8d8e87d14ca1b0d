"""Cloth (row N4 of SURVEY §8f; reference cloth.cpp, physics.cpp:1354-1358) on the GPU against the oracle's colour-ordered restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(mi, oracle, scene):
    g = scene.instantiate(mi.World())
    o = scene.instantiate(oracle.OracleWorld())
    return g, o


def test_cloths_match_the_oracle_bit_for_bit(mi, oracle):
    """Three cloths (two in LDS, one on the global planes) under wind with velocity, position and drift iterations on, 240 free-running steps:
    particle positions and velocities equal the colour-ordered oracle bit for bit (the cloth does not interact with the rigid bodies, and the
    kernels use the oracle's operation order with contraction off)."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("cloths")
    g, o = _build(mi, oracle, scene)
    for i in range(240):
        g.step_internal(scene.dt, 4); o.step_internal(scene.dt, 4)
        if i % 40 == 39 or i < 3:
            for c in range(len(scene.cloths)):
                gp, gv = g.cloth_state(c); op, ov = o.cloth_state(c)
                assert np.array_equal(gp, op) and np.array_equal(gv, ov), "step %d cloth %d: max |dp| %g" % (i, c, np.abs(gp - op).max())
    gp, gv = g.cloth_state(0)
    assert np.isfinite(gp).all() and np.abs(gv).max() > 0.1 and np.abs(gp[:, 1] - 14.0).max() > 1.0      # the wind carries it away from its flat start
    assert np.array_equal(gp[:20], o.cloth_state(0)[0][:20]) and np.abs(gv[:20]).max() == 0.0  # the locked row stays


def test_cloth_through_physics_step_and_property_changes(mi, oracle):
    """mi_step takes the iteration counts from its settings (physics.h:387-389); changing mass / stiffness re-derives the particle and
    constraint masses (cloth.cpp:198-204); moving the locked row drags the cloth."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.cloths(2)
    g, o = _build(mi, oracle, scene)
    gs, os_ = mi.Settings(numClothVelocityIterations=1, numClothPositionIterations=2, numClothDriftIterations=1), oracle.Settings(numClothVelocityIterations=1, numClothPositionIterations=2, numClothDriftIterations=1)
    for i in range(60):
        g.step(1.0 / 60.0, gs); o.step(1.0 / 60.0, os_)
        if i == 20:
            for w in (g, o):
                w.cloth_set_properties(0, 12.0, 0.9, 0.2, 1.0)
        if i == 40:
            for w in (g, o):
                w.cloth_set_fixed_vertices(1, (8.5, 9.5, -3.0), (0.0, 0.29552021, 0.0, 0.95533649), False)
    for c in range(2):
        gp, gv = g.cloth_state(c); op, ov = o.cloth_state(c)
        assert np.array_equal(gp, op) and np.array_equal(gv, ov), "cloth %d: max |dp| %g" % (c, np.abs(gp - op).max())


def test_cloth_travels_with_the_snapshot(mi):
    """A restored world continues its cloths bit-identically (particle state and derived masses are part of the image)."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("cloths")
    g = scene.instantiate(mi.World())
    for _ in range(50):
        g.step_internal(scene.dt, 4)
    r = mi.World.restore(g.snapshot())
    for _ in range(50):
        g.step_internal(scene.dt, 4); r.step_internal(scene.dt, 4)
    for c in range(len(scene.cloths)):
        gp, gv = g.cloth_state(c); rp, rv = r.cloth_state(c)
        assert np.array_equal(gp, rp) and np.array_equal(gv, rv)
