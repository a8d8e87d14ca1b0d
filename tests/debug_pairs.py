import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes
from oracle import oracle as orc
from parity_util import follow_step, pair_set
name = sys.argv[1]; steps = int(sys.argv[2])
s = scenes.by_name(name)
g = s.instantiate(mi.World()); o = s.instantiate(orc.OracleWorld(solver=orc.SOLVER_CUSTOM))
def brute(aabbs):
    mn = aabbs[:, :3]; mx = aabbs[:, 3:]
    ov = np.ones((len(mn), len(mn)), bool)
    for k in range(3):
        ov &= (mx[:, None, k] >= mn[None, :, k]) & (mn[:, None, k] <= mx[None, :, k])
    i, j = np.nonzero(np.triu(ov, 1))
    return pair_set(np.stack([i, j], 1))
for it in range(steps):
    r = follow_step(g, o, s.dt, 30)
    if not r["pairs_equal"]:
        gc, ga = g.world_colliders(); oc, oa = o.world_colliders()
        print("step", it, "aabb equal:", np.array_equal(ga, oa), "axis", o.sorting_axis())
        b = brute(oa); gs = pair_set(g.pairs()); os_ = pair_set(o.pairs())
        print("brute", len(b), "gpu", len(gs), "oracle", len(os_), "gpu==brute", np.array_equal(gs, b), "oracle==brute", np.array_equal(os_, b))
        for nm, st in (("gpu", gs), ("oracle", os_)):
            miss = np.setdiff1d(b, st); extra = np.setdiff1d(st, b)
            print(nm, "missing", len(miss), "extra", len(extra))
            for p in miss[:5]:
                i, j = int(p & 0xFFFFFFFF), int(p >> 32)
                print("   ", i, j, oa[i], oa[j])
        break
else:
    print("all equal", r)
