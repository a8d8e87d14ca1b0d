"""Developer tool: first step at which device and oracle disagree on terrain contacts; prints both contact lists of the colliders involved."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes
from oracle import oracle
from parity_util import follow_step

scene = scenes.by_name("terrain")
g = scene.instantiate(mi.World()); o = scene.instantiate(oracle.OracleWorld(solver=oracle.SOLVER_CUSTOM))
for i in range(240):
    r = follow_step(g, o, scene.dt, 30, None, resync=False)
    if not r["counts_equal"] or o.terrain_slot_mismatch():
        slots, counts, contacts, bp = g.manifolds()
        oc = o.slot_counts()
        cols, aabbs = o.world_colliders()
        print("step", i, "mismatch colliders", o.terrain_slot_mismatch(), "slots", len(slots), len(oc))
        t = slots[:, 1] >= 0x80000000
        diff = np.nonzero(counts != oc[:len(counts)])[0]
        print("slots whose counts differ:", diff, [(int(slots[d, 0]), hex(int(slots[d, 1])), int(counts[d]), int(oc[d]), int(cols["type"][slots[d, 0]])) for d in diff])
        for s in np.nonzero(t)[0][:0]:
            flag = "" if (s < len(oc) and oc[s] == counts[s]) else "  <-- differs"
            print(" slot", s, "collider", slots[s, 0], "type", cols["type"][slots[s, 0]], "k", slots[s, 1] & 0x7FFFFFFF, "gpu count", counts[s], "oracle", oc[s] if s < len(oc) else None,
                  "point", contacts[s, 0]["point"], "n", contacts[s, 0]["normal"], "depth", contacts[s, 0]["depth"], flag)
        oc_all, obp, oci = o.contacts()
        cp, cc = o.collisions()
        for ci in range(len(cols)):
            exp = o.terrain_contacts(ci)
            got = per_pre.get(ci, 0) if False else int(((slots[:, 0] == ci) & t).sum())
            if len(exp) != got or ci == 77:
                print("collider", ci, "type", cols["type"][ci], "oracle", len(exp), "device", got)
                print("  oracle:", exp)
                for s_ in np.nonzero((slots[:, 0] == ci) & t)[0]:
                    print("  device: k", slots[s_, 1] & 0x7FFFFFFF, contacts[s_, 0]["point"], contacts[s_, 0]["depth"], contacts[s_, 0]["normal"])
                print("  collider rec", cols[ci], "aabb", aabbs[ci])
        sys.exit(0)
        start = np.concatenate([[0], np.cumsum(cc)]).astype(np.int64)
        per = {}
        for s_ in np.nonzero(t)[0]:
            per[int(slots[s_, 0])] = per.get(int(slots[s_, 0]), 0) + 1
        print("device terrain contacts per collider:", per)
        for j, (a, b) in enumerate(cp):
            if b >= 0x80000000:
                c = oc_all[start[j]]
                print("  collider", a, "k", b & 0x7FFFFFFF, "point", c["point"], "n", c["normal"], "depth", c["depth"])
        break
else:
    print("no mismatch")
