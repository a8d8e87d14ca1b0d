import sys, time
sys.path.insert(0, "/root/repo")
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes
s = scenes.by_name("c5"); w = s.instantiate(mi.World())
for i in range(130):
    w.step_internal(s.dt)
w.synchronize(); print(w.stats())
