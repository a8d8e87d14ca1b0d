"""Developer tool: stage times of the settled C3 world for the current MI_FLOW_* / MI_PHYSICS_NO_FLOW environment."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes
name = sys.argv[1]; settle = int(sys.argv[2]); steps = int(sys.argv[3])
s = scenes.by_name(name)
w = s.instantiate(mi.World())
for i in range(settle):
    w.step_internal(s.dt)
w.synchronize()
w.enable_stage_timing(True)
acc = {}
for i in range(steps):
    w.step_internal(s.dt); st = w.stats()
    for k, v in st.items():
        acc[k] = acc.get(k, 0) + v / steps
print({k: round(acc[k], 3) for k in ("msSolve", "msSolverSetup", "msTotal", "numCollisions", "numColors", "flowProbes")}, flush=True)
