import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes
name = sys.argv[1]; settle = int(sys.argv[2]); steps = int(sys.argv[3])
t0 = time.time(); s = scenes.by_name(name); t1 = time.time()
w = s.instantiate(mi.World()); t2 = time.time()
print(name, "gen %.1fs instantiate %.1fs bodies %d" % (t1 - t0, t2 - t1, w.num_bodies), flush=True)
for i in range(settle):
    w.step_internal(s.dt)
    if i % 60 == 0:
        w.synchronize(); print("settle", i, w.stats(), flush=True)
w.synchronize()
w.enable_stage_timing(True)
acc = {}
for i in range(10):
    w.step_internal(s.dt); st = w.stats()
    for k, v in st.items():
        if k.startswith("ms"): acc[k] = acc.get(k, 0) + v / 10
print("stage ms:", {k: round(v, 3) for k, v in acc.items()}, st)
w.enable_stage_timing(False)
w.synchronize(); t0 = time.time()
for i in range(steps): w.step_internal(s.dt)
w.synchronize(); t1 = time.time()
print("%s: %.3f ms/step  %.1f steps/s" % (name, (t1 - t0) / steps * 1e3, steps / (t1 - t0)))
tr = w.transforms(); print("nan:", np.isnan(tr).any(), "minY %.2f maxY %.2f" % (tr[:, 1].min(), tr[:, 1].max()))
slots, cs = w.schedule()
import numpy as np
print("colour sizes:", np.diff(cs.astype(np.int64))[:int(w.stats()["numColors"]) + 1].tolist(), "serial:", int(cs[65] - cs[64]))
cog, inv = w.body_state()
pairs, counts, contacts, bp = w.manifolds()
act = counts > 0
deg = np.bincount(np.concatenate([bp[act, 0], bp[act, 1]]), minlength=w.num_bodies + 1)[:w.num_bodies]
print("body degree: max %d mean %.2f hist %s" % (deg.max(), deg.mean(), np.bincount(deg)[:24].tolist()))
