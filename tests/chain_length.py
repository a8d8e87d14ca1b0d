"""Developer experiment: depth of the per-iteration dependency chain for colour order vs random-priority order of the manifolds."""
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
pairs, counts, contacts, bp = w.manifolds()
act = counts > 0
bp = bp[act].astype(np.int64)
nb = w.num_bodies
print("manifolds", len(bp), "colours", w.stats()["numColors"])
def depth(order):
    last = np.zeros(nb + 1, np.int64); best = 0
    for m in order:
        a, b = bp[m]
        d = 1 + max(last[a] if a < nb else 0, last[b] if b < nb else 0)
        if a < nb: last[a] = d
        if b < nb: last[b] = d
        if d > best: best = d
    return best
rng = np.random.default_rng(1)
for t in range(3):
    print("random priority order: depth", depth(rng.permutation(len(bp))))
deg = np.bincount(np.concatenate([bp[:, 0], bp[:, 1]]), minlength=nb + 1)[:nb]
print("max degree", deg.max())
# hubs first: manifolds ordered by max degree of their bodies (descending), random ties
key = np.maximum(np.where(bp[:, 0] < nb, deg[np.minimum(bp[:, 0], nb - 1)], 0), np.where(bp[:, 1] < nb, deg[np.minimum(bp[:, 1], nb - 1)], 0))
print("hub-first order: depth", depth(np.lexsort((rng.random(len(bp)), -key))))
print("leaf-first order: depth", depth(np.lexsort((rng.random(len(bp)), key))))
