"""Developer tool: timeline of the dataflow sweep: per colour when its manifolds see their inputs each iteration, and the hop times along the
busiest bodies' chains (consecutive users of one body)."""
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
raw = w.flow_trace(True, n).astype(np.int64)
tr = raw[:, :30]
t0 = tr[tr > 0].min()
tr = (tr - t0) * 0.01  # us
slept10 = (raw[:, 31] - t0) * 0.01
print("manifolds", n, "colours", nc, "sweep length %.1f us" % tr.max())
print("iteration end (max over all manifolds), us:", np.round(tr.max(axis=0), 1).tolist())
print("per-iteration period (median over manifolds): %.2f us" % np.median(np.diff(tr, axis=1)))
cs = cs.astype(np.int64)
for c in range(nc):
    blk = tr[cs[c]:cs[c + 1]]
    if len(blk):
        print("colour %2d n=%6d  it0 done: median %.1f max %.1f | it1: median %.1f max %.1f | it29: median %.1f max %.1f" % (c, len(blk), np.median(blk[:, 0]), blk[:, 0].max(), np.median(blk[:, 1]), blk[:, 1].max(), np.median(blk[:, 29]), blk[:, 29].max()))

# hop times along the busiest bodies' chains: consecutive users of one body, within one iteration
pairs, counts, contacts, bp = w.manifolds()
sched_bp = bp[slots[:n].astype(np.int64)]            # body pair of each schedule position
nb = w.num_bodies
deg = np.bincount(np.concatenate([sched_bp[:, 0], sched_bp[:, 1]]), minlength=nb + 1)[:nb]
order = np.argsort(-deg)[:3]
col_of_pos = np.searchsorted(cs[:66].astype(np.int64), np.arange(n), side="right") - 1
for body in order:
    users = np.nonzero((sched_bp[:, 0] == body) | (sched_bp[:, 1] == body))[0]   # schedule positions, ascending = colour order
    print("body %d: contact counts of its users, in colour order: %s" % (body, counts[slots[users].astype(np.int64)].tolist()))
    for it in (10,):
        t = tr[users, it]
        print("body %d degree %d iteration %d: finish times of its users (us) %s  hops %s" % (body, deg[body], it, np.round(t - t[0], 2).tolist(), np.round(np.diff(t), 2).tolist()))
    # what the other body of each user was waiting for: its previous user's finish time relative to this user's finish
    it = 10
    for u in users:
        other = sched_bp[u, 1] if sched_bp[u, 0] == body else sched_bp[u, 0]
        if other >= nb:
            continue
        ou = np.nonzero((sched_bp[:, 0] == other) | (sched_bp[:, 1] == other))[0]
        k = int(np.nonzero(ou == u)[0][0])
        prev_t = tr[ou[k - 1], it] if k > 0 else tr[ou[-1], it - 1]
        hub_k = int(np.nonzero(users == u)[0][0])
        hub_prev = tr[users[hub_k - 1], it] if hub_k > 0 else tr[users[-1], it - 1]
        ready = max(hub_prev, prev_t)
        print("   user colour %2d count %d: both predecessors had their inputs at %.2f | this one slept until %+.2f | saw its inputs %+.2f (relative to that)" % (col_of_pos[u], counts[slots[u]], ready - tr[users[0], it], slept10[u] - ready, tr[u, it] - ready))
