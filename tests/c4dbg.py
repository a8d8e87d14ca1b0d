import sys; sys.path.insert(0, "/root/repo")
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes
s = scenes.by_name("c4"); w = s.instantiate(mi.World())
for i in range(60): w.step_internal(s.dt)
w.synchronize(); w.enable_stage_timing(True)
for i in range(20): w.step_internal(s.dt)
st = w.stats()
print({k: st[k] for k in ("numCollisions", "numContacts", "numColors", "numJoints", "clusterTasks", "clusterManifolds", "clusterSharedBodies", "clusterParts", "numFlowRecoveries", "msSolverSetup", "msSolve", "msTotal")})
