"""Developer tool: long free run of a BASELINE configuration on the device; checks that the state stays finite, nothing leaves the scene, and
that the safety nets of the persistent kernels never fired.  usage: python tests/soak.py [scene] [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
scene = scenes.by_name(name)
w = scene.instantiate(mi.World())
t0 = time.time()
for i in range(steps):
    w.step_internal(scene.dt, 30)
    if i % 500 == 499:
        s = w.stats()
        print("step", i + 1, "contacts", s["numContacts"], "manifolds", s["numCollisions"], "colors", s["numColors"], "tasks", s["clusterTasks"], "recoveries", s["numFlowRecoveries"], "elapsed %.1fs" % (time.time() - t0), flush=True)
w.synchronize()
dt = time.time() - t0
t = w.transforms(1); v = w.velocities(); s = w.stats()
print("%s: %d steps in %.2fs (%.1f steps/s incl. host loop)" % (name, steps, dt, steps / dt))
print("finite", bool(np.isfinite(t).all() and np.isfinite(v).all()), "y range", float(t[:, 1].min()), float(t[:, 1].max()), "max |v|", float(np.abs(v).max()), "flow recoveries", s["numFlowRecoveries"])
assert np.isfinite(t).all() and np.isfinite(v).all() and s["numFlowRecoveries"] == 0 and t[:, 1].min() > -1.0
