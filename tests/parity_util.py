"""Shared helpers for the GPU-vs-oracle parity tests.

Whole-step parity works in "follow" mode: the HIP world steps first; the oracle then (1) computes its own world-space
colliders, AABBs and sort-and-sweep pair SET, (2) runs its narrowphase on the device's ordered candidate pairs and (3) solves
contacts manifold by manifold in the device's colour schedule (and joints in the device's joint order).  Within a colour no two
manifolds share a dynamic body, so sequential execution on the CPU is the same Gauss-Seidel sweep the GPU ran in parallel.
"""
import numpy as np


def pair_set(pairs):
    p = np.asarray(pairs, np.uint64).reshape(-1, 2)
    lo = np.minimum(p[:, 0], p[:, 1]); hi = np.maximum(p[:, 0], p[:, 1])
    return np.unique((hi << np.uint64(32)) | lo)


def quat_dist(a, b):
    """max |q_a - (+/-)q_b| per body."""
    d1 = np.abs(a - b).max(axis=1); d2 = np.abs(a + b).max(axis=1)
    return np.minimum(d1, d2)


def own_narrowphase_check(orc_world, o_pairs, slots, g_counts, tie_pairs=None):
    """The oracle's OWN prune + narrowphase on its OWN broadphase pairs (collision_narrow.cpp:2358-2379, restated here on the oracle's
    world-space colliders) against what the device kept: the SET of colliding collider pairs and the contact total must agree.  In
    follow mode the oracle evaluates the device's post-classify pair list, so a pair the device pruned by mistake would be invisible
    there; here it shows up as a missing colliding pair.  tie_pairs: the endpoint-tie pairs the reference's sweep drops and the device
    reports (accepted by follow_step); they are added to the oracle's list so that both sides test the same candidates."""
    from oracle import oracle as orc
    RIGID, STATIC = 0, 1  # physics_object_type (oracle/oshapes.h:198-201)
    cols, aabbs = orc_world.world_colliders()
    p = np.asarray(o_pairs, np.int64).reshape(-1, 2)
    if tie_pairs is not None and len(tie_pairs):
        # (as the sweep would have reported them: the collider that starts later on the sorting axis first; equal starts: higher index first)
        t = np.asarray(tie_pairs, np.uint64)
        lo, hi = (t & np.uint64(0xFFFFFFFF)).astype(np.int64), (t >> np.uint64(32)).astype(np.int64)
        ax = orc_world.sorting_axis()[0]
        lo_later = aabbs[lo, ax] > aabbs[hi, ax]
        p = np.concatenate([p, np.stack([np.where(lo_later, lo, hi), np.where(lo_later, hi, lo)], axis=1)])
    ta, tb = cols["objectType"][p[:, 0]], cols["objectType"][p[:, 1]]
    ia, ib = cols["objectIndex"][p[:, 0]], cols["objectIndex"][p[:, 1]]
    keep = ((ta == RIGID) | (tb == RIGID)) & ~((ta == RIGID) & (tb == RIGID) & (ia == ib))       # :2358-2369
    keep &= ((ta == RIGID) & (tb == RIGID)) | (ta == STATIC) | (tb == STATIC)                      # collisions only (:2378-2379)
    p = p[keep]
    swap = cols["type"][p[:, 0]] >= cols["type"][p[:, 1]]                                         # typeA < typeB or swapped (:2374: equal types are swapped too)
    p[swap] = p[swap][:, ::-1]
    # The oracle's OWN orientation stands (its sweep reports (later start, earlier start), collision_broad.cpp:127; :2374 swaps equal
    # types, so A starts first on the sorting axis).  Only equal-type pairs whose boxes start at exactly the same coordinate, where the
    # reference's order is that of its endpoint array from earlier frames and the device's the collider index, are evaluated in the
    # device's orientation; they are counted.
    sl = np.asarray(slots, np.int64).reshape(-1, 2)
    axis = orc_world.sorting_axis()[0]
    same_type = cols["type"][p[:, 0]] == cols["type"][p[:, 1]]
    start_tie = same_type & (aabbs[p[:, 0], axis] == aabbs[p[:, 1], axis])
    if start_tie.any() and len(sl):
        dev_key = (np.maximum(sl[:, 0], sl[:, 1]) << 32) | np.minimum(sl[:, 0], sl[:, 1])
        order = np.argsort(dev_key)
        own_key = (np.maximum(p[:, 0], p[:, 1]) << 32) | np.minimum(p[:, 0], p[:, 1])
        pos = np.clip(np.searchsorted(dev_key[order], own_key), 0, len(order) - 1)
        hit = start_tie & (dev_key[order][pos] == own_key)
        p[hit] = sl[order][pos][hit]
    orc_world.use_hull_geometries()
    _, counts = orc.narrowphase_ordered(cols, p.astype(np.uint32))
    own = pair_set(p[counts > 0]); dev = pair_set(np.asarray(slots)[np.asarray(g_counts) > 0])
    return {"own_colliding_equal": bool(np.array_equal(own, dev)), "own_missing_on_device": int(len(np.setdiff1d(own, dev))),
            "own_extra_on_device": int(len(np.setdiff1d(dev, own))), "own_contacts": int(counts.sum()), "device_contacts": int(np.asarray(g_counts).sum()),
            "own_start_ties": int(start_tie.sum())}


def orientation_check(gpu, orc_world, slots):
    """The device's A/B order of its candidate pairs against the reference's rule, evaluated on the ORACLE's colliders, boxes and sorting
    axis: lower collider type first (collision_narrow.cpp:2374); equal types: the box that starts first on the sweep's sorting axis
    (collision_broad.cpp:127 emits (later start, earlier start), :2374 swaps it).  Returns the number of slots that break the rule and
    the number of exact start ties (where the reference's order comes from its endpoint array's history; not judged)."""
    cols, aabbs = orc_world.world_colliders()
    sl = np.asarray(slots, np.int64).reshape(-1, 2)
    sl = sl[(sl[:, 0] < len(cols)) & (sl[:, 1] < len(cols))]   # (terrain contacts have no collider pair)
    if not len(sl):
        return {"orient_bad": 0, "orient_ties": 0, "axis_equal": True}
    axis = orc_world.sorting_axis()[0]
    ta, tb = cols["type"][sl[:, 0]], cols["type"][sl[:, 1]]
    sa, sb = aabbs[sl[:, 0], axis], aabbs[sl[:, 1], axis]
    same = ta == tb
    bad = (ta > tb) | (same & (sa > sb))
    return {"orient_bad": int(bad.sum()), "orient_ties": int((same & (sa == sb)).sum()), "axis_equal": bool(gpu.sorting_axis()[0] == axis)}


def follow_step(gpu, orc_world, scene_dt, iterations=30, joint_counts=None, resync=False, own_narrowphase=False):
    """Step both worlds once; returns a dict of comparison metrics.  `orc_world.solver` must be SOLVER_CUSTOM.
    resync=True copies the oracle's state into the device world after the comparison, so every step starts from identical
    inputs (used where libm-vs-device trigonometry makes free-running trajectories drift chaotically: joints)."""
    gpu.step_internal(scene_dt, iterations)
    g_pairs = gpu.pairs()
    slots, g_counts, g_contacts, g_bp = gpu.manifolds()
    order, cs = gpu.schedule()
    if joint_counts:
        for t, n in joint_counts.items():
            if n:
                orc_world.set_joint_order(t, gpu.joint_order(t, n))
    orc_world.set_follow(slots, order)
    prev_var = np.sort(orc_world.sorting_variance().astype(np.float64))  # decided this step's sorting axis
    orc_world.step_internal(scene_dt, iterations)
    o_pairs = orc_world.pairs()
    o_counts = orc_world.slot_counts().astype(np.uint32)
    gs, os_ = pair_set(g_pairs), pair_set(o_pairs)
    pairs_ok = np.array_equal(gs, os_)
    num_ties = 0
    tie_pairs = None
    if not pairs_ok:
        # The reference's sort-and-sweep drops a pair whose endpoints TIE on the sorting axis depending on the previous frame's
        # endpoint order (stable insertion sort with '>', collision_broad.cpp:387-398) although aabbVsAABB is inclusive
        # (bounding_volumes.h:352-358).  The device reports the inclusive set; accept exactly those extra pairs.
        extra = np.setdiff1d(gs, os_); missing = np.setdiff1d(os_, gs)
        _, aabbs = orc_world.world_colliders()
        axis = orc_world.sorting_axis()[0]
        i = (extra & np.uint64(0xFFFFFFFF)).astype(np.int64); j = (extra >> np.uint64(32)).astype(np.int64)
        tie = (aabbs[i, 3 + axis] == aabbs[j, axis]) | (aabbs[j, 3 + axis] == aabbs[i, axis])
        pairs_ok = len(missing) == 0 and bool(tie.all())
        num_ties = int(len(extra))
        tie_pairs = extra if pairs_ok else None
    out = {
        "pairs_equal": pairs_ok, "num_tie_pairs": num_ties,
        "num_pairs": len(g_pairs), "num_slots": len(slots), "num_manifolds": int((g_counts > 0).sum()),
        "counts_equal": np.array_equal(g_counts, o_counts[:len(g_counts)]) and len(o_counts) == len(g_counts),
        "num_colors": int((np.diff(cs[:65].astype(np.int64)) > 0).sum()),
    }
    if out["counts_equal"] and len(slots):
        oc, obp, oci = orc_world.contacts()
        mask = np.arange(4)[None, :] < g_counts[:, None]
        gc = g_contacts[mask]
        out["contact_point_err"] = float(np.abs(gc["point"] - oc["point"]).max()) if len(oc) else 0.0
        out["contact_depth_err"] = float(np.abs(gc["depth"] - oc["depth"]).max()) if len(oc) else 0.0
        out["contact_normal_err"] = float(np.abs(gc["normal"] - oc["normal"]).max()) if len(oc) else 0.0
        out["contact_fr_equal"] = bool(np.array_equal(gc["friction_restitution"], oc["friction_restitution"])) if len(oc) else True
        out["num_contacts"] = int(len(oc))
    out.update(orientation_check(gpu, orc_world, slots))
    # the device sums the axis statistic in double, the reference in float collider after collider: they may pick different axes only
    # where the two largest variances agree to within that float sum's rounding
    out["axis_near_tie"] = bool(prev_var[2] - prev_var[1] <= 1e-3 * max(prev_var[2], 1e-30))
    if own_narrowphase:
        out.update(own_narrowphase_check(orc_world, o_pairs, slots, g_counts, tie_pairs))
    gt, ot = gpu.transforms(1), orc_world.transforms(1)
    gv, ov = gpu.velocities(), orc_world.velocities()
    out["pos_err"] = float(np.abs(gt[:, :3] - ot[:, :3]).max())
    out["rot_err"] = float(quat_dist(gt[:, 3:], ot[:, 3:]).max())
    out["vel_err"] = float(np.abs(gv - ov).max())
    out["vel_scale"] = float(np.abs(ov).max())
    if resync:
        gpu.write_state(ot, ov)
    return out
