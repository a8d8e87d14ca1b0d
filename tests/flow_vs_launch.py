"""Developer check: the dataflow sweep and the launch-per-colour sweep must give bit-identical trajectories (same schedule)."""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes
name = sys.argv[1]; steps = int(sys.argv[2])
s = scenes.by_name(name)
os.environ.pop("MI_PHYSICS_NO_FLOW", None)
wf = s.instantiate(mi.World())
os.environ["MI_PHYSICS_NO_FLOW"] = "1"
wl = s.instantiate(mi.World())
for i in range(steps):
    wf.step_internal(s.dt)
    if len(sys.argv) > 3: wf.synchronize()  # no overlap of the two worlds on the GPU
    wl.step_internal(s.dt)
    tf, tl = wf.transforms(1), wl.transforms(1)
    vf, vl = wf.velocities(), wl.velocities()
    if not (np.array_equal(tf, tl) and np.array_equal(vf, vl)) or i % 20 == 0 or i == steps - 1:
        bad = np.nonzero((vf != vl).any(axis=1))[0]
        print("step %d: manifolds %d/%d colours %d/%d  differing bodies %d  max |dv| %.3e" % (i, wf.stats()["numCollisions"], wl.stats()["numCollisions"], wf.stats()["numColors"], wl.stats()["numColors"], len(bad), float(np.abs(vf - vl).max())), flush=True)
        if len(bad):
            print("first differing bodies:", bad[:10], "stats", wf.stats())
            try:
                wf.step_internal(s.dt); wf.synchronize(); wf.step_internal(s.dt)
            except Exception as e:
                print("next step raised:", e)
            break
