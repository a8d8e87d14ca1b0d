"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle (the reference ships no test vectors and cannot
be built here — SURVEY.md §4/§8c — so these pin OUR restatement against regressions and give the GPU tests fixed expected outputs).

  python tests/golden/make_golden.py

Fixtures (numpy .npz, loadable with allow_pickle=False):
  narrow_pairs.npz   10 in-scope collider type pairs x 48 seeded poses (incl. degenerate ones): scene description, world-space
                     colliders/AABBs, broadphase pair set and per-pair contacts as produced by one oracle step.
  narrow_pairs_cylinder.npz  the same for the 5 type pairs that involve a cylinder.
  narrow_pairs_hull.npz      the same for the 6 type pairs that involve a convex hull (box, octahedron, icosahedron geometries).
  scheduler.npz      body-pair lists -> exact 8-lane slot tables of the greedy batch scheduler (constraints.cpp:51-184).
  c1_trajectory.npz  config 1 (64 OBBs on the ground): transforms + velocities after 1, 60, 120, 240 steps, scalar and 8-lane solver.
  ragdoll_trajectory.npz  one humanoid ragdoll dropped on the ground at 60 Hz (learned_locomotion.cpp:440-446,469-474): 1, 30, 120 steps.
  kat.npz            analytic known-answer inputs/outputs (free fall with damping, box inertia, sphere at rest).
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from directx_renderer_kurth_amd import scenes  # noqa: E402
from oracle import oracle as orc  # noqa: E402

SPHERE, CAPSULE, CYLINDER, AABB, OBB, HULL = 0, 1, 2, 3, 4, 5
MAT = (0.1, 0.5, 1.0)


def hulls_to_arrays(s):
    """Hull geometries of a scene as flat arrays: vertices, triangles, and the (first, count) ranges of both per geometry."""
    v = np.concatenate([h[0] for h in s.hulls]) if s.hulls else np.zeros((0, 3), np.float32)
    t = np.concatenate([h[1] for h in s.hulls]) if s.hulls else np.zeros((0, 3), np.uint32)
    r = np.array([[sum(len(g[0]) for g in s.hulls[:i]), len(h[0]), sum(len(g[1]) for g in s.hulls[:i]), len(h[1])] for i, h in enumerate(s.hulls)], np.int64).reshape(-1, 4)
    return v, t, r


def scene_to_arrays(s):
    bodies = np.array([list(b[0]) + list(b[1]) + [float(b[2]), b[3], b[4], b[5]] for b in s.bodies], np.float32).reshape(-1, 11)
    cols = np.zeros((len(s.colliders), 22), np.float64)
    for i, (body, ctype, shape, mat, pos, rot) in enumerate(s.colliders):
        cols[i, 0] = body; cols[i, 1] = ctype; cols[i, 2:2 + len(shape)] = shape; cols[i, 12:15] = mat; cols[i, 15:18] = pos; cols[i, 18:22] = rot
    return bodies, cols


def scene_from_arrays(bodies, cols, name="fixture", dt=1.0 / 120.0, hull_vertices=None, hull_triangles=None, hull_ranges=None):
    s = scenes.Scene(name, dt)
    if hull_ranges is not None:
        for v0, nv, t0, nt in hull_ranges:
            s.add_hull_geometry(hull_vertices[v0:v0 + nv], hull_triangles[t0:t0 + nt])
    for b in bodies:
        s.add_body(b[0:3], b[3:7], kinematic=bool(b[7]), gravity_factor=float(b[8]), linear_damping=float(b[9]), angular_damping=float(b[10]))
    nshape = {0: 4, 1: 7, 2: 7, 3: 6, 4: 10, 5: 8}
    for c in cols:
        ctype = int(c[1])
        s.add_collider(int(c[0]), ctype, [np.float32(x) for x in c[2:2 + nshape[ctype]]], tuple(np.float32(c[12:15])), tuple(np.float32(c[15:18])), tuple(np.float32(c[18:22])))
    return s


def narrow_scene(seed=7321, per_pair=48, kinds=("sphere", "capsule", "aabb", "obb"), only_with=None):
    rng = scenes.XorShift64(seed)
    s = scenes.Scene("narrow_pairs")
    kinds = list(kinds)
    case = 0
    geoms = []
    if "hull" in kinds:
        geoms = [s.add_hull_geometry(*scenes.hull_box(0.5, 0.35, 0.45)), s.add_hull_geometry(*scenes.hull_octahedron(0.6)), s.add_hull_geometry(*scenes.hull_icosahedron(0.55))]

    def shape(kind):
        if kind == "sphere":
            return SPHERE, (0, 0, 0, rng.between(0.3, 0.6)), 0.6
        if kind == "capsule":
            h = rng.between(0.3, 0.6)
            return CAPSULE, (0, -h, 0, 0, h, 0, rng.between(0.2, 0.4)), 1.0
        if kind == "cylinder":
            h = rng.between(0.3, 0.6)
            return CYLINDER, (0, -h, 0, 0, h, 0, rng.between(0.2, 0.5)), 1.0
        if kind == "hull":
            q = rng.unit_quat()
            return HULL, (float(q[0]), float(q[1]), float(q[2]), float(q[3]), 0, 0, 0, float(geoms[int(rng.between(0, 2.999))])), 0.8
        he = (rng.between(0.3, 0.6), rng.between(0.3, 0.6), rng.between(0.3, 0.6))
        if kind == "aabb":
            return AABB, (-he[0], -he[1], -he[2], he[0], he[1], he[2]), 1.0
        q = rng.unit_quat()
        return OBB, (float(q[0]), float(q[1]), float(q[2]), float(q[3]), 0, 0, 0) + he, 1.0

    for ia, ka in enumerate(kinds):
        for kb in kinds[ia:]:
            if only_with is not None and only_with not in (ka, kb):
                continue
            for k in range(per_pair):
                base = ((case % 32) * 40.0, 5.0, (case // 32) * 40.0)
                ta, sa, ra = shape(ka)
                tb, sb, rb = shape(kb)
                ident = (0.0, 0.0, 0.0, 1.0)
                rot_a = ident if (ka == "aabb" or k % 8 == 1) else tuple(float(x) for x in rng.unit_quat())
                rot_b = ident if (kb == "aabb" or k % 8 == 1) else tuple(float(x) for x in rng.unit_quat())
                if k % 8 == 2:
                    rot_b = rot_a  # parallel axes (capsule |cos| > 0.99 branch, SAT `parallel` flag)
                d = np.array([rng.between(-1, 1), rng.between(-1, 1), rng.between(-1, 1)])
                d = d / max(1e-6, float(np.linalg.norm(d))) * rng.between(0.0, 1.15 * (ra + rb) * 0.75)
                if k % 8 == 0:
                    d = d * 0.0  # coincident centres: degenerate normal branches
                if k % 8 == 3:
                    d = np.array([0.0, float(np.linalg.norm(d)), 0.0])  # stacked along +y (face contacts)
                a = s.add_body(base, rot_a)
                b = s.add_body((base[0] + float(d[0]), base[1] + float(d[1]), base[2] + float(d[2])), rot_b)
                s.add_collider(a, ta, sa, MAT)
                s.add_collider(b, tb, sb, MAT)
                case += 1
    return s


def gen_narrow(fname="narrow_pairs.npz", **kw):
    s = narrow_scene(**kw)
    w = s.instantiate(orc.OracleWorld())
    w.step_internal(1e-9, 1)
    cols, aabbs = w.world_colliders()
    pairs = w.pairs()
    cpairs, counts = w.collisions()
    contacts, bp, ci = w.contacts()
    bodies, carr = scene_to_arrays(s)
    hv, ht, hr = hulls_to_arrays(s)
    np.savez_compressed(os.path.join(HERE, fname), hull_vertices=hv, hull_triangles=ht, hull_ranges=hr, bodies=bodies, colliders=carr, world_colliders=cols.view(np.uint8).reshape(len(cols), 64),
                        aabbs=aabbs, pairs=pairs, colliding_pairs=cpairs, counts=counts, contacts=contacts.view(np.uint8).reshape(len(contacts), 32),
                        contact_collision=ci, mass=w.mass_properties())
    print(fname + ": %d bodies, %d pairs, %d collisions, %d contacts" % (len(bodies), len(pairs), len(cpairs), len(contacts)), orc.stats())


def gen_scheduler():
    rng = np.random.default_rng(1234)
    out = {}
    for name, n, nb, pstatic in (("small", 37, 12, 0.3), ("chain", 200, 201, 0.0), ("dense", 1000, 120, 0.2), ("ground", 500, 500, 1.0)):
        a = rng.integers(0, nb, n).astype(np.uint32)
        b = rng.integers(0, nb, n).astype(np.uint32)
        if name == "chain":
            a = np.arange(n, dtype=np.uint32); b = a + 1
        if name == "ground":
            a = np.arange(n, dtype=np.uint32)
        b = np.where(b == a, (b + 1) % nb, b).astype(np.uint32)
        stat = rng.random(n) < pstatic
        b = np.where(stat, nb, b).astype(np.uint32)  # dummy id = nb
        bp = np.stack([a, b], 1)
        out[name + "_pairs"] = bp
        out[name + "_dummy"] = np.uint32(nb)
        out[name + "_slots"] = orc.schedule(bp, nb)
    np.savez_compressed(os.path.join(HERE, "scheduler.npz"), **out)
    print("scheduler:", {k: v.shape for k, v in out.items() if k.endswith("slots")})


def gen_trajectory(fname, scene, checkpoints, modes):
    out = {}
    bodies, carr = scene_to_arrays(scene)
    out["bodies"] = bodies; out["colliders"] = carr
    for mode_name, mode in modes:
        w = scene.instantiate(orc.OracleWorld(solver=mode))
        step = 0
        for cp in checkpoints:
            while step < cp:
                w.step_internal(scene.dt); step += 1
            out["%s_t%d" % (mode_name, cp)] = w.transforms(1)
            out["%s_v%d" % (mode_name, cp)] = w.velocities()
            out["%s_n%d" % (mode_name, cp)] = np.array([len(w.pairs()), len(w.contacts()[0])], np.uint32)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, {k: v.tolist() for k, v in out.items() if "_n" in k})


def gen_kat():
    # free fall with damping (rigid_body.cpp:95-124): v_{n+1} = (v_n + g dt) / (1 + dt * 0.4); sphere far above everything
    s = scenes.Scene("kat")
    b = s.add_body((0, 1000.0, 0))
    s.add_collider(b, SPHERE, (0, 0, 0, 0.5), MAT)
    w = s.instantiate(orc.OracleWorld())
    vs = []
    for _ in range(10):
        w.step_internal(1.0 / 120.0)
        vs.append(w.velocities()[0, 1])
    # box inertia (physics.cpp:1498-1502): diag = m/12 (dy^2+dz^2) ...
    s2 = scenes.Scene("kat2")
    b2 = s2.add_body((0, 0, 0))
    s2.add_collider(b2, AABB, (-0.5, -1.0, -1.5, 0.5, 1.0, 1.5), (0.1, 0.5, 2.0))
    w2 = s2.instantiate(orc.OracleWorld())
    np.savez_compressed(os.path.join(HERE, "kat.npz"), free_fall_vy=np.array(vs, np.float32), box_mass=w2.mass_properties())
    print("kat: vy", vs[:3], "box", w2.mass_properties()[0, 3:])


def overlap_flags(fixture):
    """Boolean overlapCheck of every broadphase pair of a narrow_pairs fixture (type-ordered like the classify step)."""
    g = np.load(os.path.join(HERE, fixture), allow_pickle=False)
    s = scene_from_arrays(g["bodies"], g["colliders"], hull_vertices=g["hull_vertices"], hull_triangles=g["hull_triangles"], hull_ranges=g["hull_ranges"])
    w = s.instantiate(orc.OracleWorld())
    w.step_internal(1e-9, 1)
    w.use_hull_geometries()
    cols, _ = w.world_colliders()
    pairs = g["pairs"].astype(np.uint32).copy()
    swap = ~(cols["type"][pairs[:, 0]] < cols["type"][pairs[:, 1]])   # collision_narrow.cpp:2374 (swaps on equal types too)
    pairs[swap] = pairs[swap][:, ::-1]
    return pairs, orc.overlap_ordered(cols, pairs), cols


def gen_overlap():
    out = {}
    for name in ("narrow_pairs", "narrow_pairs_cylinder", "narrow_pairs_hull"):
        pairs, flags, _ = overlap_flags(name + ".npz")
        out[name + "_pairs"] = pairs; out[name + "_overlaps"] = flags
    np.savez_compressed(os.path.join(HERE, "overlap_pairs.npz"), **out)


if __name__ == "__main__":
    orc.build()
    gen_narrow()
    gen_narrow("narrow_pairs_cylinder.npz", seed=99173, per_pair=64, kinds=("sphere", "capsule", "cylinder", "aabb", "obb"), only_with="cylinder")
    gen_narrow("narrow_pairs_hull.npz", seed=40411, per_pair=64, kinds=("sphere", "capsule", "cylinder", "aabb", "obb", "hull"), only_with="hull")
    gen_scheduler()
    gen_trajectory("c1_trajectory.npz", scenes.c1_boxes(), (1, 60, 120, 240), (("scalar", orc.SOLVER_SCALAR), ("wide8", orc.SOLVER_WIDE8)))
    ragdoll = scenes.c4_ragdolls(1)
    gen_trajectory("ragdoll_trajectory.npz", ragdoll, (1, 30, 120), (("scalar", orc.SOLVER_SCALAR), ("wide8", orc.SOLVER_WIDE8)))
    gen_kat()
    gen_overlap()
