"""Two spatial slabs on ONE GPU (two processes, gloo for the rendezvous + host-staged messages): the multi-GPU code path
(mask, device state hand-over, ownership, ghosts) against the single-world run of the same scene."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world_size, port, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    torch.cuda.set_device(0)
    from directx_renderer_kurth_amd import scenes, parallel
    scene = scenes.by_name("c3_small")
    sw = parallel.SlabWorld(scene, device=0, rank=rank, world_size=world_size, margin=3.5, comm_on_cpu=True)
    owners = torch.zeros(scene.num_bodies, dtype=torch.int32, device="cuda")
    for _ in range(steps):
        sw.step_internal(scene.dt)
        own = (sw.code == parallel.OWNED).to(torch.int32)
        tot = own.cpu().clone(); dist.all_reduce(tot)
        assert bool((tot == 1).all()), "ownership is not a partition"
    t = sw.transforms(1); v = sw.velocities()
    if rank == 0:
        np.save(os.path.join(out_dir, "t.npy"), t); np.save(os.path.join(out_dir, "v.npy"), v)
        np.save(os.path.join(out_dir, "stats.npy"), np.array([sw.exchanger.bytes_sent, int((sw.code != parallel.INACTIVE).sum())]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_slabs_match_single_world(tmp_path, mi):
    from directx_renderer_kurth_amd import scenes
    steps = 40
    scene = scenes.by_name("c3_small")
    w = scene.instantiate(mi.World())
    for _ in range(steps):
        w.step_internal(scene.dt)
    ref_t, ref_v = w.transforms(1), w.velocities()
    w.close()
    mp.spawn(_worker, args=(2, _free_port(), steps, str(tmp_path)), nprocs=2, join=True)
    t = np.load(os.path.join(str(tmp_path), "t.npy")); v = np.load(os.path.join(str(tmp_path), "v.npy"))
    sent, active = np.load(os.path.join(str(tmp_path), "stats.npy"))
    assert np.isfinite(t).all() and sent > 0 and active < scene.num_bodies
    err = np.abs(t[:, :3] - ref_t[:, :3]).max(axis=1)
    # Gauss-Seidel inside a slab, block-Jacobi across the cut: bodies away from the cut follow the single-world trajectory closely,
    # the pile as a whole stays the same pile.
    print("slab vs single world after %d steps: median |dx| %.2e, 99th pct %.2e, max %.2e, halo bytes %d" % (steps, np.median(err), np.percentile(err, 99), err.max(), sent))
    frac_off = float((err > 0.01).mean())
    print("bodies off by more than 1 cm: %.2f %%" % (100 * frac_off))
    assert np.median(err) < 1e-3                  # most bodies follow the single-world trajectory to rounding level
    # The block is only 14 m wide and the ghost band 2 x 3.5 m: about half of all bodies are coupled through the cut within a
    # few contacts, and a collapsing pile amplifies the Jacobi-vs-Gauss-Seidel difference there to centimetres within 40 steps.
    assert frac_off < 0.60
    assert err.max() < 2.0                        # and none of them is ejected
    assert abs(t[:, 1].mean() - ref_t[:, 1].mean()) < 0.02
