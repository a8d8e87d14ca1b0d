"""The dataflow contact sweep (one persistent kernel for all iterations) against the launch-per-colour sweep: same schedule, same
per-body order, so trajectories must be bit-identical — on a world large enough for many workgroups, and with joints (one dataflow
launch per iteration)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _world(mi, scene, flow):
    old = os.environ.pop("MI_PHYSICS_NO_FLOW", None)
    try:
        if not flow:
            os.environ["MI_PHYSICS_NO_FLOW"] = "1"
        return scene.instantiate(mi.World())    # the switch is read when the world is created
    finally:
        os.environ.pop("MI_PHYSICS_NO_FLOW", None)
        if old is not None:
            os.environ["MI_PHYSICS_NO_FLOW"] = old


@pytest.mark.parametrize("name,steps", [("c3_mid", 150), ("c4_small", 60), ("shapes_hull", 100)])
def test_flow_equals_launch_sweep(mi, name, steps):
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name(name)
    wf, wl = _world(mi, scene, True), _world(mi, scene, False)
    used_flow = False
    for i in range(steps):
        wf.step_internal(scene.dt); wf.synchronize()   # the two worlds never overlap on the GPU (a persistent kernel wants the whole chip)
        wl.step_internal(scene.dt); wl.synchronize()
        used_flow = used_flow or wf.stats()["flowProbes"] > 0
        assert wl.stats()["flowProbes"] == 0
        if i % 10 == 9 or i == steps - 1:
            assert np.array_equal(wf.transforms(1), wl.transforms(1)), "step %d" % i
            assert np.array_equal(wf.velocities(), wl.velocities()), "step %d" % i
    assert used_flow
    assert wf.stats()["numCollisions"] > 0


@pytest.mark.parametrize("abort_step,steps", [(50, 60), (11, 12)])
def test_flow_abort_is_recovered(mi, abort_step, steps):
    """Safety net of the persistent kernel: if the dataflow sweep of a step gives up (injected here), the device skips that step's
    integration and the host redoes solve + integration with the launch sweep from the saved pre-solve velocities — at the next
    step's first synchronisation (abort in the middle of the run) or when results are read (abort in the last step).  The trajectory
    must equal the launch-sweep one bit for bit."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("c4_small" if abort_step == 50 else "c3_small")   # ragdolls (joints + ground contacts by step 50) / a pile
    os.environ["MI_FLOW_TEST_ABORT"] = str(abort_step)
    try:
        wf = _world(mi, scene, True)
    finally:
        os.environ.pop("MI_FLOW_TEST_ABORT", None)
    wl = _world(mi, scene, False)
    for i in range(steps):
        wf.step_internal(scene.dt); wl.step_internal(scene.dt); wl.synchronize()
    assert np.array_equal(wf.transforms(1), wl.transforms(1))
    assert np.array_equal(wf.velocities(), wl.velocities())
    assert wf.stats()["numFlowRecoveries"] == 1
