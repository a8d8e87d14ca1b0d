"""World-size-2/3 CPU tests (gloo) of the slab partition + ghost-body halo exchange used for multi-GPU runs.
The exchange logic is device-agnostic torch code; here the per-rank stepper is a trivial advection so that the ownership /
ghost invariants can be checked exactly without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world_size, port, n, steps, margin, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from directx_renderer_kurth_amd import parallel as par
    g = torch.Generator().manual_seed(1234)                     # same initial world on every rank
    pose = torch.zeros((n, 8)); vel = torch.zeros((n, 8))
    pose[:, 0] = torch.rand(n, generator=g) * 40.0 - 20.0; pose[:, 1] = torch.rand(n, generator=g) * 5.0; pose[:, 7] = 1.0
    vel[:, 0] = (torch.rand(n, generator=g) - 0.5) * 60.0         # up to 30 m/s: 0.25 m per step, lots of cut crossings
    vel[:, 3] = 1.0
    cuts = par.quantile_cuts(pose[:, 0].numpy(), world_size)
    ex = par.HaloExchanger(rank, world_size, cuts, axis=0, margin=margin)
    code = ex.initial_code(pose)
    dt = 1.0 / 120.0
    ok = True
    for step in range(steps):
        ex.exchange(pose, vel, code)
        # ---- invariants, checked collectively ----
        owned = (code == par.OWNED).to(torch.int32)
        total = owned.clone(); dist.all_reduce(total)
        ok &= bool((total == 1).all())                            # exactly one owner per body
        ref_pose = pose * owned[:, None].float(); dist.all_reduce(ref_pose)   # the owners' states
        active = code != par.INACTIVE
        ok &= bool(torch.equal(pose[active], ref_pose[active]))   # every simulated copy (owned or ghost) carries the owner's state
        x = ref_pose[:, 0]
        need = ((x >= ex.lo - margin) & (x < ex.hi + margin))     # everything within `margin` of my slab must be simulated here
        # bodies that just migrated are owned elsewhere and may sit up to one step's travel beyond the band: allow 0.3 m slack inward
        core = ((x >= ex.lo - margin + 0.3) & (x < ex.hi + margin - 0.3))
        ok &= bool(active[core].all())
        ok &= bool((~active | need | (torch.abs(x - ex.lo) < margin + 0.6) | (torch.abs(x - ex.hi) < margin + 0.6)).all())
        # ---- fake step: advect what this rank simulates; bounce at the world's ends ----
        pose[active, 0] += vel[active, 0] * dt
        flip = active & ((pose[:, 0] > 20.0) | (pose[:, 0] < -20.0))
        vel[flip, 0] = -vel[flip, 0]
    np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([ok, ex.bytes_sent]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2, 3])
def test_halo_exchange_invariants(tmp_path, world_size):
    port = _free_port()
    mp.spawn(_worker, args=(world_size, port, 2000, 150, 1.5, str(tmp_path)), nprocs=world_size, join=True)
    for r in range(world_size):
        ok, sent = np.load(os.path.join(str(tmp_path), "ok_%d.npy" % r))
        assert ok == 1, "rank %d violated an ownership/ghost invariant" % r
        assert sent > 0


def test_quantile_cuts_balance():
    from directx_renderer_kurth_amd import parallel as par
    x = np.random.default_rng(0).normal(size=10000)
    cuts = par.quantile_cuts(x, 8)
    counts = np.histogram(x, bins=[-np.inf] + cuts + [np.inf])[0]
    assert len(cuts) == 7 and counts.max() - counts.min() <= 2
