"""Row N3 of SURVEY §8f: checkpoint + resume.  A world restored from a snapshot continues bit-identically."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["shapes_hull", "c4_small"])
def test_snapshot_restore_continues_bit_identically(mi, name):
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name(name)
    w = scene.instantiate(mi.World())
    for _ in range(40):
        w.step_internal(scene.dt)
    if name == "shapes_hull":
        w.delete_body(7)                                  # deleted bodies and pending forces are part of the image
        w.apply_force_torque(11, (30.0, 0.0, -20.0), (0.0, 5.0, 0.0))
    blob = w.snapshot()
    for _ in range(40):
        w.step_internal(scene.dt)
    r = mi.World.restore(blob)
    assert r.num_bodies == w.num_bodies and r.num_colliders == w.num_colliders
    for _ in range(40):
        r.step_internal(scene.dt)
    assert np.array_equal(w.transforms(1), r.transforms(1))
    assert np.array_equal(w.velocities(), r.velocities())
    assert r.stats()["numJoints"] == w.stats()["numJoints"]
    with pytest.raises(mi.PhysicsError):
        mi.World.restore(blob[:100])
