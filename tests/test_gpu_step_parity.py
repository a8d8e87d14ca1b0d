"""GPU parity (through the C-ABI) against the CPU oracle in follow mode, on the BASELINE configs at oracle-sized scales."""
import numpy as np
import pytest

from parity_util import follow_step

pytestmark = pytest.mark.gpu

KINDS = {"distance": 0, "ball": 1, "fixed": 2, "hinge": 3, "cone_twist": 4, "slider": 5}


def _worlds(mi, oracle, scene):
    g = scene.instantiate(mi.World())
    o = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_CUSTOM))
    return g, o


def _joint_counts(scene):
    out = {}
    for j in scene.joints:
        k = KINDS[j[0][:-6] if j[0].endswith("_local") else j[0]]
        out[k] = out.get(k, 0) + 1
    return out


def _run(mi, oracle, scene, steps, resync, vel_tol, pos_tol):
    g, o = _worlds(mi, oracle, scene)
    jc = _joint_counts(scene)
    worst = {}
    ties = 0
    for i in range(steps):
        r = follow_step(g, o, scene.dt, 30, jc, resync=resync)
        assert r["pairs_equal"], "step %d: broadphase pair set differs" % i
        assert r["counts_equal"], "step %d: contact counts differ" % i
        assert r.get("contact_fr_equal", True)
        ties += r["num_tie_pairs"]
        for k in ("contact_point_err", "contact_depth_err", "contact_normal_err", "pos_err", "rot_err", "vel_err"):
            if k in r:
                worst[k] = max(worst.get(k, 0.0), r[k])
        assert r["vel_err"] <= vel_tol * max(1.0, r["vel_scale"]), "step %d: velocity error %g (scale %g)" % (i, r["vel_err"], r["vel_scale"])
    print(scene.name, "worst errors over", steps, "steps:", worst, "colors", r["num_colors"], "contacts", r.get("num_contacts"), "tie pairs", ties)
    assert worst["pos_err"] <= pos_tol and worst["rot_err"] <= pos_tol
    for k in ("contact_point_err", "contact_depth_err", "contact_normal_err"):
        assert worst.get(k, 0.0) <= 1e-5 * (1.0 + 100.0), k  # 1e-5 abs + 1e-5 rel on coordinates up to ~100 m
    return worst


@pytest.mark.parametrize("name,steps", [("c1", 120), ("c2_small", 60), ("c3_small", 60), ("shapes", 90), ("shapes_hull", 150)])
def test_follow_trajectory_contacts(mi, oracle, name, steps):
    """Free-running trajectories (the device world is never re-synchronised; errors accumulate).  Stated tolerances:
    pair SET exact (modulo the reference's endpoint-tie artefact, see parity_util); contact counts exact; contacts 1e-5;
    velocities 1e-4 relative per step; positions 1e-3 m after 60+ steps.  In practice these runs are bit-exact: the kernels are
    built with -ffp-contract=off and use correctly rounded sqrt/div, like the oracle."""
    from directx_renderer_kurth_amd import scenes
    _run(mi, oracle, scenes.by_name(name), steps, resync=False, vel_tol=1e-4, pos_tol=1e-3)


def test_follow_ragdolls_per_step(mi, oracle):
    """Config 4 (hinge + cone-twist chains).  Joint init calls atan2f/acosf/sinf/cosf, whose device (ocml) and glibc results differ
    in the last ulp; ragdoll dynamics amplify that chaotically, so each step is compared from identical inputs (resync)."""
    from directx_renderer_kurth_amd import scenes
    _run(mi, oracle, scenes.by_name("c4_small"), 120, resync=True, vel_tol=1e-4, pos_tol=1e-4)


def _run_multi_task_joints(mi, oracle, scene, steps, min_tasks, need_later_phase):
    """Per-step follow of a jointed world large enough for SEVERAL cluster tasks: the island -> task assignment (k_cl_joint_weights /
    _assign / _scatter on the refresh steps, k_cl_joint_assign_cached in between), the joint offsets of the tasks behind the first,
    contacts between islands of different tasks (cut into the later phases), jointed bodies handed between phases."""
    g, o = _worlds(mi, oracle, scene)
    jc = _joint_counts(scene)
    most_tasks, later_steps, worst_v, worst_p = 0, 0, 0.0, 0.0
    for i in range(steps):
        r = follow_step(g, o, scene.dt, 30, jc, resync=True)
        assert r["pairs_equal"], "step %d: broadphase pair set differs" % i
        assert r["counts_equal"], "step %d: contact counts differ" % i
        assert r["vel_err"] <= 1e-4 * max(1.0, r["vel_scale"]), "step %d: velocity error %g (scale %g)" % (i, r["vel_err"], r["vel_scale"])
        assert r["pos_err"] <= 1e-4 and r["rot_err"] <= 1e-4, "step %d: pose error %g / %g" % (i, r["pos_err"], r["rot_err"])
        st = g.stats()
        assert st["numFlowRecoveries"] == 0, "step %d: the cluster sweep gave up" % i
        most_tasks = max(most_tasks, st["clusterTasks"][0]); later_steps += sum(st["clusterTasks"][1:]) > 0
        worst_v = max(worst_v, r["vel_err"]); worst_p = max(worst_p, r["pos_err"])
    print(scene.name, "per-step follow over", steps, "steps: worst velocity error %.2e, pose error %.2e; most first-phase tasks %d, steps with later-phase tasks %d; last step: tasks %s manifolds %s contacts %s"
          % (worst_v, worst_p, most_tasks, later_steps, st["clusterTasks"], st["clusterManifolds"], r.get("num_contacts")))
    assert most_tasks >= min_tasks, "the world ran in %d first-phase tasks only" % most_tasks
    if need_later_phase:
        assert later_steps >= steps // 4, "islands of different tasks hardly ever touched (%d of %d steps)" % (later_steps, steps)


def test_follow_c4_full_size_per_step(mi, oracle):
    """BASELINE config 4 at its full size: 256 ragdolls (3 584 bodies, 3 328 joints) — the cluster sweep spreads the islands over
    ~20 tasks, the benchmark's own path for --workload c4 — per step from identical inputs, 1e-4."""
    from directx_renderer_kurth_amd import scenes
    _run_multi_task_joints(mi, oracle, scenes.by_name("c4"), 60, min_tasks=16, need_later_phase=False)


def test_follow_ragdoll_heap_per_step(mi, oracle):
    """128 ragdolls packed closer than their arm span, in two layers, dropped into a heap: islands of different tasks touch."""
    from directx_renderer_kurth_amd import scenes
    _run_multi_task_joints(mi, oracle, scenes.by_name("c4_heap"), 90, min_tasks=4, need_later_phase=True)


def test_follow_joints_mix_per_step(mi, oracle):
    """The joint kinds and add variants the BASELINE configs leave out, per step from identical inputs: distance joints (device
    kernels k_distance_init / k_distance_solve; constraints.cpp:189-264) added from global and from local points, ball joints from
    local points, cone-twist swing and twist motors of both motor types (constraints.cpp:1880-1960), with contacts in the same solve."""
    from directx_renderer_kurth_amd import scenes
    _run(mi, oracle, scenes.by_name("joints_mix"), 180, resync=True, vel_tol=1e-4, pos_tol=1e-4)


def test_ragdolls_free_running_invariants(mi):
    """Config 4 free-running for 2 s: ragdoll dynamics are chaotic, so instead of trajectory parity the joints' own invariants are
    checked on the device result: finite state, nothing below the ground, and every joint's two anchors stay together (< 6 cm;
    the oracle itself peaks at 4.3-4.4 cm on this scene with either Gauss-Seidel order: Baumgarte beta 0.1, 30 iterations)."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.by_name("c4_small")
    g = scene.instantiate(mi.World())
    for _ in range(120):
        g.step_internal(scene.dt)
    t = g.transforms(1)
    assert np.isfinite(t).all() and np.isfinite(g.velocities()).all()
    assert t[:, 1].min() > -0.05 and t[:, 1].max() < 2.0

    def rot(q, v):
        x, y, z, w = q
        u = np.array([x, y, z]); return v + 2.0 * np.cross(u, np.cross(u, v) + w * v)

    worst = 0.0
    for ctype, n in ((3, sum(1 for j in scene.joints if j[0] == "hinge")), (4, sum(1 for j in scene.joints if j[0] == "cone_twist"))):
        pairs = [(j[1], j[2]) for j in scene.joints if KINDS[j[0]] == ctype]
        for cid, (a, b) in enumerate(pairs):
            pod = g.constraint_get(ctype, cid).view(np.float32)
            pa = t[a, :3] + rot(t[a, 3:], pod[0:3]); pb = t[b, :3] + rot(t[b, 3:], pod[3:6])
            worst = max(worst, float(np.linalg.norm(pa - pb)))
    print("worst joint anchor separation after 120 steps: %.4f m" % worst)
    assert worst < 0.06


def test_physics_step_fixed_timestep(mi, oracle):
    """physicsStep() semantics (physics.cpp:1364-1413): accumulator, <= 4 sub-steps per frame, dropped time, interpolated transform."""
    from directx_renderer_kurth_amd import scenes
    scene = scenes.c1_boxes(16)
    g = scene.instantiate(mi.World())
    o = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_SCALAR))
    # scalar-order oracle differs from the coloured schedule at solver-convergence level only once contacts exist;
    # the first frames are free fall, where the stepping logic is what is being compared.
    for dt in (1 / 60.0, 1 / 90.0, 0.004, 0.05, 1 / 120.0):
        g.step(dt, mi.Settings()); o.step(dt, oracle.Settings())
        assert abs(g.timer.value - o.timer.value) < 1e-7
        for which in (0, 1, 2):
            np.testing.assert_allclose(g.transforms(which), o.transforms(which), atol=1e-6)
