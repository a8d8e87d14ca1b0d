"""Developer tool: per-step overhead of the slab wrapper (state hand-over + ownership bookkeeping + exchange) at world_size 1."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes, parallel
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
name = sys.argv[1]; settle = int(sys.argv[2]); steps = int(sys.argv[3])
s = scenes.by_name(name)
for kind in ("plain", "slab"):
    w = s.instantiate(mi.World()) if kind == "plain" else parallel.SlabWorld(s, device=0, rank=0, world_size=1)
    for i in range(settle): w.step_internal(s.dt)
    w.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): w.step_internal(s.dt)
    w.synchronize(); torch.cuda.synchronize()
    print("%s %s: %.3f ms/step" % (name, kind, (time.perf_counter() - t0) / steps * 1e3), flush=True)
dist.destroy_process_group()
