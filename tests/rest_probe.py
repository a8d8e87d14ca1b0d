"""Developer tool: max penetration / kinetic energy of rest_stacks over time, device (production order) vs the oracle's two reference orders."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes
from oracle import oracle as orc
name = sys.argv[1] if len(sys.argv) > 1 else "rest_stacks"; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 480
s = scenes.by_name(name)
g = s.instantiate(mi.World()); os_ = s.instantiate(orc.OracleWorld(solver=orc.SOLVER_SCALAR)); ow = s.instantiate(orc.OracleWorld(solver=orc.SOLVER_WIDE8))
for i in range(steps):
    g.step_internal(s.dt); os_.step_internal(s.dt); ow.step_internal(s.dt)
    if i >= 200 and i % 20 == 19:
        slots, counts, contacts, bp = g.manifolds()
        mask = np.arange(4)[None, :] < counts[:, None]
        dg = contacts["depth"][mask]; ds = os_.contacts()[0]["depth"]; dw = ow.contacts()[0]["depth"]
        v = g.velocities(); 
        print("step %3d  max penetration mm: device %.3f  scalar %.3f  wide %.3f | 99th pct %.3f %.3f %.3f | mean %.3f %.3f %.3f | contacts %d %d %d | max |v| %.4f" % (
            i + 1, 1e3 * dg.max(), 1e3 * ds.max(), 1e3 * dw.max(), 1e3 * np.percentile(dg, 99), 1e3 * np.percentile(ds, 99), 1e3 * np.percentile(dw, 99), 1e3 * dg.mean(), 1e3 * ds.mean(), 1e3 * dw.mean(), len(dg), len(ds), len(dw), np.abs(v).max()))
