"""Rank-in-isolation benchmark of the spatial slabs: a PROJECTED strong-scaling figure from ONE GPU.

The unsplit world (config 5's generator at N bodies) runs on the one GPU and its per-step states are recorded.  Then, for 2 / 4 / 8
slabs cut at body-count quantiles on the axis of largest variance, every rank is run ALONE on the same GPU: it owns its slab, its
incoming halo messages are rebuilt from the recorded unsplit trajectory (what the neighbours would have sent: their bodies inside
the band at the cut, with the MIGRATE flag for those that crossed it) and staged in device memory beforehand, and each step runs the
rank's real per-step sequence — mi_slab_pack, mi_slab_unpack, mi_step_internal — timed by the host clock around the whole run.
projected speed-up = unsplit ms/step / slowest rank's ms/step.  It is a projection: no message travels over a link (at ~0.4 MB per
neighbour and step against 150 GB/s per xGMI link the wire time is ~3 us), ranks do not wait for each other, and every rank's
neighbours are the unsplit world, not other slabs.  usage: python tests/slab_projection.py [bodies] [settle] [steps]"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
import directx_renderer_kurth_amd as mi
from directx_renderer_kurth_amd import scenes
from directx_renderer_kurth_amd.parallel import quantile_cuts, max_variance_axis

bodies = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 100
K = int(sys.argv[3]) if len(sys.argv) > 3 else 30
MARGIN = 3.0
t0 = time.time()
scene = scenes.c3_mixed(bodies, area=700.0 * (bodies / 1e6) ** 0.5)
w = scene.instantiate(mi.World())
print("scene + world: %.0f s, %d bodies" % (time.time() - t0, w.num_bodies), flush=True)
for _ in range(settle):
    w.step_internal(scene.dt)
w.synchronize()
inv_mass = w.mass_properties()[:, 3].copy()
rec_t, rec_v = [w.transforms(1)], [w.velocities()]
for _ in range(K):
    w.step_internal(scene.dt)
    rec_t.append(w.transforms(1)); rec_v.append(w.velocities())
# the unsplit world's own speed over the same K steps (re-run from the first recorded state, no read-backs)
def timed_run(world, prepare, per_step):
    prepare()
    for k in range(3):
        per_step(k)            # warm-up on the first recorded steps (list lengths, buffer sizes settle)
    prepare()
    world.synchronize(); torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(K):
        per_step(k)
    world.synchronize(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / K * 1e3
ms_unsplit = timed_run(w, lambda: w.write_state(rec_t[0], rec_v[0]), lambda k: w.step_internal(scene.dt))
st = w.stats()
print("unsplit: %.3f ms/step (%d contacts, tasks %s, recoveries %d)" % (ms_unsplit, st["numContacts"], st["clusterTasks"], st["numFlowRecoveries"]), flush=True)

dev = torch.device("cuda", 0)
axis = max_variance_axis(rec_t[0][:, :3])
x = [t[:, axis] for t in rec_t]
def message(k, lo, hi, side):
    """What the neighbour on `side` (0 = left of lo, 1 = right of hi) would send before step k: its bodies (by the previous recorded
    state) that now lie inside the band of width MARGIN at the cut; flag 1 = crossed into this slab."""
    prev, cur = x[max(k - 1, 0)], x[k]
    if side: sel = (prev >= hi) & (cur < hi + MARGIN); mig = cur < hi
    else: sel = (prev < lo) & (cur >= lo - MARGIN); mig = cur >= lo
    idx = np.nonzero(sel)[0]
    n = len(idx)
    rec = np.zeros((n, 18), np.uint32)
    rec[:, 0] = idx; rec[:, 1] = mig[idx].astype(np.uint32)
    f = rec[:, 2:].view(np.float32)
    f[:, 0:3] = rec_t[k][idx, 0:3]; f[:, 4:8] = rec_t[k][idx, 3:7]
    f[:, 8:11] = rec_v[k][idx, 0:3]; f[:, 11] = inv_mass[idx]; f[:, 12:15] = rec_v[k][idx, 3:6]
    return n, rec
lines = []
for size in (2, 4, 8):
    cuts = quantile_cuts(x[0].astype(np.float64), size)
    worst, per_rank = 0.0, []
    for rank in range(size):
        lo = cuts[rank - 1] if rank > 0 else -float("inf")
        hi = cuts[rank] if rank < size - 1 else float("inf")
        msgs = [[message(k, lo, hi, s) if ((s == 0 and rank > 0) or (s == 1 and rank < size - 1)) else None for s in (0, 1)] for k in range(K)]
        cap = max([m[0] for mk in msgs for m in mk if m is not None] + [1])
        cap = int(cap * 1.25) + 256
        nbytes = w.slab_message_bytes(cap)
        def stage(m):
            if m is None: return None
            buf = np.zeros(nbytes // 4, np.uint32); buf[0] = m[0]; buf[4:4 + m[1].size] = m[1].reshape(-1)
            return torch.from_numpy(buf.view(np.uint8)).to(dev)
        staged = [[stage(m) for m in mk] for mk in msgs]
        out = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) if staged[0][s] is not None else None for s in (0, 1)]
        ptr = lambda t: t.data_ptr() if t is not None else 0
        def prepare():
            w.write_state(rec_t[0], rec_v[0]); w.slab_configure(rank, size, axis, lo, hi, MARGIN)
        def per_step(k):
            w.slab_pack(ptr(out[0]), ptr(out[1]), cap)
            w.slab_unpack(ptr(staged[k][0]), ptr(staged[k][1]), cap)
            w.step_internal(scene.dt)
        ms = timed_run(w, prepare, per_step)
        codes = w.slab_codes(); st = w.stats()
        per_rank.append(ms); worst = max(worst, ms)
        print("  %d slabs, rank %d: %.3f ms/step; owned %d ghosts %d; ghost records per step and side <= %d; contacts %d; recoveries %d" % (
            size, rank, ms, int((codes == 1).sum()), int((codes >= 2).sum()), cap, st["numContacts"], st["numFlowRecoveries"]), flush=True)
    lines.append("%d slabs: slowest rank %.3f ms/step (ranks: %s) -> projected %.2fx of the unsplit %.3f ms/step" % (size, worst, " ".join("%.2f" % m for m in per_rank), ms_unsplit / worst, ms_unsplit))
print("\n".join(lines))
