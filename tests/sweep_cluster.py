"""Developer tool: solve / setup time of C3 at two points of the run for a list of environment settings (one world per setting).
usage: python tests/sweep_cluster.py "MI_CLUSTER_TASK=800" "MI_CLUSTER_TASK=1000 MI_CLUSTER_TASK_LATER=500" ..."""
import os, sys
sys.path.insert(0, "/root/repo")
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes
s = scenes.by_name("c3")
for setting in sys.argv[1:] or [""]:
    env = dict(kv.split("=") for kv in setting.split()) if setting else {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    w = s.instantiate(mi.World())
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    out = []
    done = 0
    for upto in (245, 360, 440):
        for _ in range(upto - done): w.step_internal(s.dt)
        done = upto
        w.synchronize(); w.stats(); w.enable_stage_timing(True)
        for _ in range(20): w.step_internal(s.dt)
        done += 20
        st = w.stats(); w.enable_stage_timing(False)
        out.append("step %d: contacts %6d solve %.3f setup %.3f total %.3f tasks %s rec %d" % (upto, st["avgContacts"], st["msSolve"], st["msSolverSetup"], st["msTotal"], st["clusterTasks"], st["numFlowRecoveries"]))
    print("[%s]\n   " % setting + "\n   ".join(out), flush=True)
    w.close()
