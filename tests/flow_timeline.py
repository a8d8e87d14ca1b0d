"""Developer tool: timeline of the dataflow sweep (per colour: when its manifolds finish each iteration)."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes
name = sys.argv[1]; settle = int(sys.argv[2])
s = scenes.by_name(name)
w = s.instantiate(mi.World())
for i in range(settle):
    w.step_internal(s.dt)
w.synchronize()
w.flow_trace(True)
w.step_internal(s.dt); w.synchronize()
slots, cs = w.schedule()
n = int(w.stats()["numCollisions"]); nc = int(w.stats()["numColors"])
tr = w.flow_trace(True, n).astype(np.int64)[:, :30]
t0 = tr[tr > 0].min()
tr = (tr - t0) * 0.01  # us
print("manifolds", n, "colours", nc, "sweep length %.1f us" % tr.max())
print("iteration end (max over all manifolds), us:", np.round(tr.max(axis=0), 1).tolist())
print("per-iteration period (median over manifolds): %.2f us" % np.median(np.diff(tr, axis=1)))
cs = cs.astype(np.int64)
for c in range(nc):
    blk = tr[cs[c]:cs[c + 1]]
    if len(blk):
        print("colour %2d n=%6d  it0 done: median %.1f max %.1f | it1: median %.1f max %.1f | it29: median %.1f max %.1f" % (c, len(blk), np.median(blk[:, 0]), blk[:, 0].max(), np.median(blk[:, 1]), blk[:, 1].max(), np.median(blk[:, 29]), blk[:, 29].max()))
