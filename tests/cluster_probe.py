"""Developer probe: follow-mode parity of the cluster sweep on small scenes, printed step by step (not a test)."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes
from oracle import oracle as orc
from parity_util import follow_step
names = sys.argv[1].split(","); steps = int(sys.argv[2])
for name in names:
    sc = scenes.by_name(name)
    g = sc.instantiate(mi.World()); o = sc.instantiate(orc.OracleWorld(solver=orc.SOLVER_CUSTOM))
    worst = 0.0
    for i in range(steps):
        r = follow_step(g, o, sc.dt, 30)
        worst = max(worst, r["vel_err"])
        if i % 10 == 0 or r["vel_err"] > 1e-6 or not r["pairs_equal"] or not r["counts_equal"]:
            print(name, i, {k: r[k] for k in ("pairs_equal", "counts_equal", "num_manifolds", "num_colors", "vel_err", "pos_err")}, flush=True)
        if r["vel_err"] > 1e-2: break
    st = g.stats()
    print(name, "worst vel err", worst, "recoveries", st["numFlowRecoveries"], "colors", st["numColors"], flush=True)
