"""Spatial slabs across the GPUs of one node with a ghost-body halo exchange (RCCL over xGMI via torch.distributed).

The reference is a single process with no notion of this; the design is ours.

* The world is cut into `world_size` slabs along one axis at body-count quantiles.  Rank r OWNS the bodies whose position lies in its
  slab.  Every rank allocates ALL bodies (same indices everywhere — 1 M bodies are 100 MB of state in a 288 GB HBM), but simulates
  only its owned bodies plus GHOST copies of the neighbours' bodies within `margin` of the cut; everything else is masked out on
  the device (no AABB, no integration).
* Once per step, before the step: each rank sends its owned bodies that lie within `margin` of a cut (pose + velocity, 64 B each,
  plus index and a flag) to that neighbour with point-to-point send/recv — two messages per neighbour, no all-reduce on the data
  path (xGMI links are point-to-point).  A body that crossed the cut is sent with flag MIGRATE: the receiver becomes its owner,
  the sender keeps it as a ghost.  Ghosts that dropped out of the neighbour's message are masked out.
* Contacts between an owned and a ghost body are generated and solved on both ranks from identical inputs; ghost results are
  discarded at the next exchange (block-Jacobi coupling across the cut, Gauss-Seidel inside a slab).

`SlabWorld` is the product path: the halo is packed and applied by HIP kernels inside the world (mi_slab_pack / mi_slab_unpack), torch only
moves the fixed-capacity messages.  `HaloExchanger` is the same protocol's bookkeeping written with torch tensors: the model the CPU
tests run with gloo (world_size 2 and 3, no GPU) and the device kernels are compared with (tests/test_gpu_slabs.py).
"""
import numpy as np
import torch
import torch.distributed as dist

INACTIVE, OWNED, GHOST_LEFT, GHOST_RIGHT = 0, 1, 2, 3
FLAG_GHOST, FLAG_MIGRATE = 0, 1


def quantile_cuts(x, world_size):
    """Cut positions (world_size - 1 values) giving equal body counts per slab."""
    xs = np.sort(np.asarray(x, np.float64))
    return [float(xs[(len(xs) * r) // world_size]) for r in range(1, world_size)]


class HaloExchanger:
    """Ownership + ghost bookkeeping on flat state tensors.

    pose: [N, 8] float32 (pos.xyz, 0, quat), vel: [N, 8] float32 (v.xyz, invMass, w.xyz, 0), code: [N] int8 in {INACTIVE, OWNED, GHOST_*}.
    """

    def __init__(self, rank, world_size, cuts, axis=0, margin=3.0, device="cpu", group=None, comm_on_cpu=False):
        self.rank, self.world_size, self.axis, self.margin, self.device, self.group = rank, world_size, axis, float(margin), device, group
        self.comm_device = "cpu" if comm_on_cpu else device  # gloo rehearsals of GPU worlds stage the messages through host memory
        self.lo = cuts[rank - 1] if rank > 0 else -float("inf")
        self.hi = cuts[rank] if rank < world_size - 1 else float("inf")
        self.bytes_sent = 0

    def initial_code(self, pose):
        x = pose[:, self.axis]
        code = torch.zeros(pose.shape[0], dtype=torch.int8, device=pose.device)
        code[(x >= self.lo) & (x < self.hi)] = OWNED
        # ghosts of the initial configuration: what the neighbours would send in their first exchange
        code[(x >= self.hi) & (x < self.hi + self.margin)] = GHOST_RIGHT
        code[(x < self.lo) & (x >= self.lo - self.margin)] = GHOST_LEFT
        return code

    def _pack(self, pose, vel, code, to_right):
        x = pose[:, self.axis]
        owned = code == OWNED
        if to_right:
            band = owned & (x >= self.hi - self.margin)
            migrate = owned & (x >= self.hi)
        else:
            band = owned & (x < self.lo + self.margin)
            migrate = owned & (x < self.lo)
        idx = torch.nonzero(band, as_tuple=False).flatten()
        meta = torch.stack([idx.to(torch.int32), migrate[idx].to(torch.int32)], dim=1).contiguous()
        payload = torch.cat([pose[idx], vel[idx]], dim=1).contiguous()
        return idx, migrate, meta, payload

    def exchange(self, pose, vel, code):
        """In-place update of pose/vel/code.  Collective over the group (neighbour point-to-point only)."""
        left = self.rank - 1 if self.rank > 0 else None
        right = self.rank + 1 if self.rank < self.world_size - 1 else None
        send = {}
        for nb, to_right in ((left, False), (right, True)):
            if nb is not None:
                send[nb] = self._pack(pose, vel, code, to_right)
        # 1) sizes
        cd = self.comm_device
        counts_out = {nb: torch.tensor([s[2].shape[0]], dtype=torch.int64, device=cd) for nb, s in send.items()}
        counts_in = {nb: torch.zeros(1, dtype=torch.int64, device=cd) for nb in send}
        ops = []
        for nb in send:
            ops.append(dist.P2POp(dist.isend, counts_out[nb], nb, group=self.group))
            ops.append(dist.P2POp(dist.irecv, counts_in[nb], nb, group=self.group))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        # 2) payloads
        recv = {}
        ops = []
        for nb, (idx, migrate, meta, payload) in send.items():
            n_in = int(counts_in[nb].item())
            recv[nb] = (torch.zeros((n_in, 2), dtype=torch.int32, device=cd), torch.zeros((n_in, 16), dtype=torch.float32, device=cd))
            if meta.shape[0]:
                ops.append(dist.P2POp(dist.isend, meta.to(cd), nb, group=self.group))
                ops.append(dist.P2POp(dist.isend, payload.to(cd), nb, group=self.group))
                self.bytes_sent += meta.numel() * 4 + payload.numel() * 4
            if n_in:
                ops.append(dist.P2POp(dist.irecv, recv[nb][0], nb, group=self.group))
                ops.append(dist.P2POp(dist.irecv, recv[nb][1], nb, group=self.group))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        # 3) apply: bodies I migrated out become ghosts of that neighbour (their state here is the freshest there is)
        for nb, (idx, migrate, meta, payload) in send.items():
            code[migrate] = GHOST_RIGHT if nb == right else GHOST_LEFT
        for nb, (meta, payload) in recv.items():
            meta, payload = meta.to(self.device), payload.to(self.device)
            ghost_code = GHOST_RIGHT if nb == right else GHOST_LEFT
            stale = code == ghost_code
            if meta.shape[0]:
                idx = meta[:, 0].to(torch.int64)
                keep = torch.zeros_like(stale)
                keep[idx] = True
                # a ghost I just created by migrating a body out is not in the neighbour's message yet: keep it this step
                mig_out = send[nb][1]
                stale = stale & ~keep & ~mig_out
                code[stale] = INACTIVE
                pose[idx] = payload[:, :8]
                vel[idx] = payload[:, 8:]
                new_code = torch.where(meta[:, 1] == FLAG_MIGRATE, torch.tensor(OWNED, dtype=torch.int8, device=self.device),
                                       torch.tensor(ghost_code, dtype=torch.int8, device=self.device))
                code[idx] = new_code
            else:
                code[stale & ~send[nb][1]] = INACTIVE
        return pose, vel, code


def max_variance_axis(positions):
    """The axis the reference's sweep would sort on: largest variance of the centres (collision_broad.cpp:374-376, 443-444)."""
    return int(np.argmax(np.var(np.asarray(positions, np.float64), axis=0)))


class SlabWorld:
    """One HIP world per rank; same interface subset as `World` for bench.py (step_internal / stats / synchronize / transforms).

    The halo runs on the device (mi_slab_pack / mi_slab_unpack, include/mi_physics.h): per step ONE pack kernel fills a
    fixed-capacity message per neighbour (count in its header: no size round trip), ONE batch of point-to-point sends / receives
    moves them (RCCL over xGMI, enqueued on the world's own stream: the host does not wait for it), two small kernels apply what
    arrived.  No host synchronisation is added to the step's own one.  `comm_on_cpu` (gloo rehearsals of several ranks on one GPU)
    stages the messages through host memory instead, which does synchronise.
    The slabs are cut along the axis of largest variance of the initial positions, at body-count quantiles, and re-cut every
    `recut_interval` steps at the quantiles of the CURRENT positions (a pile that flows away from its start would otherwise
    unbalance the slabs): every rank then needs every body's state, which is assembled with two all-reduces of the owners'
    contributions — the one place where the slab runner uses a collective, once per few hundred steps."""

    def __init__(self, scene, device, rank, world_size, axis=None, margin=3.0, comm_on_cpu=False, capacity=None, recut_interval=240):
        import directx_renderer_kurth_amd as mi
        self.rank, self.world_size, self.scene, self.comm_on_cpu = rank, world_size, scene, comm_on_cpu
        self.dev = torch.device("cuda", device)
        self.world = scene.instantiate(mi.World(device=device))
        n = scene.num_bodies
        x0 = np.array([b[0] for b in scene.bodies], np.float64)
        self.axis = max_variance_axis(x0) if axis is None else axis
        cuts = quantile_cuts(x0[:, self.axis], world_size)
        lo = cuts[rank - 1] if rank > 0 else -float("inf")
        hi = cuts[rank] if rank < world_size - 1 else float("inf")
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world_size - 1 else None
        self.margin, self.cuts, self.recut_interval, self.steps, self.recuts = margin, cuts, int(recut_interval), 0, 0
        self.world.slab_configure(rank, world_size, self.axis, lo, hi, margin)
        # Message capacity (records per neighbour and direction; the whole fixed-size message travels every step): twice the band
        # population, measured — at the start from the scene (bodies within `margin` of any cut, the fullest band of ALL ranks, so that
        # every rank arrives at the same size without talking), then every `resize_interval` steps from the messages' own headers
        # (one all-reduce, off the per-step path).  A message that overflowed is an error, never silent.
        self.fixed_capacity = capacity is not None
        if capacity is None:
            xs = x0[:, self.axis]
            band = max([int(((xs >= c - margin) & (xs < c + margin)).sum()) for c in cuts] + [0])
            capacity = max(1024, 2 * band)
        self.capacity, self.resize_interval = int(capacity), 32
        self.stream = torch.cuda.ExternalStream(self.world.device_pointers()[2], device=self.dev)  # the world's stream: torch's comm work is enqueued on it
        self._allocate_messages()
        self.bytes_sent = 0
        self.host_syncs = 0      # synchronisations the exchange itself adds (0 on the RCCL path between capacity checks)

    def _allocate_messages(self):
        nbytes = self.world.slab_message_bytes(self.capacity)
        with torch.cuda.stream(self.stream):
            self.out = {nb: torch.zeros(nbytes, dtype=torch.uint8, device=self.dev) for nb in (self.left, self.right) if nb is not None}
            self.inc = {nb: torch.zeros(nbytes, dtype=torch.uint8, device=self.dev) for nb in (self.left, self.right) if nb is not None}

    def check_capacity(self):
        """Collective, every `resize_interval` steps: the fullest message of any rank decides the next capacity (2 x, same on all ranks);
        a message that did not hold its band raises."""
        self.world.synchronize()
        head = [t[:8].cpu().view(torch.int32) for t in self.out.values()]
        self.host_syncs += 1
        mine = torch.tensor([max([int(h[0]) for h in head] + [0]), sum(int(h[1]) for h in head)], dtype=torch.int64, device="cpu" if self.comm_on_cpu else self.dev)
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        fullest, dropped = int(mine[0]), int(mine[1])
        if dropped:
            raise RuntimeError("slab halo: %d bodies did not fit a message of %d records (capacity is adapted every %d steps: the band filled faster than 2x)" % (dropped, self.capacity, self.resize_interval))
        want = max(1024, 2 * fullest)
        if not self.fixed_capacity and (want > self.capacity or want < self.capacity // 2):
            self.capacity = want
            self._allocate_messages()

    def _ptr(self, table, nb):
        return table[nb].data_ptr() if nb is not None else 0

    def exchange(self):
        w = self.world
        with torch.cuda.stream(self.stream):
            w.slab_pack(self._ptr(self.out, self.left), self._ptr(self.out, self.right), self.capacity)
            if self.comm_on_cpu:
                host_out = {nb: t.cpu() for nb, t in self.out.items()}       # (synchronises: rehearsal path only)
                host_in = {nb: torch.zeros_like(t) for nb, t in host_out.items()}
                self.host_syncs += 1
                ops = []
                for nb in host_out:
                    ops.append(dist.P2POp(dist.isend, host_out[nb], nb)); ops.append(dist.P2POp(dist.irecv, host_in[nb], nb))
                for r in (dist.batch_isend_irecv(ops) if ops else []):
                    r.wait()
                for nb, t in host_in.items():
                    self.inc[nb].copy_(t)
            else:
                ops = []
                for nb in self.out:
                    ops.append(dist.P2POp(dist.isend, self.out[nb], nb)); ops.append(dist.P2POp(dist.irecv, self.inc[nb], nb))
                for r in (dist.batch_isend_irecv(ops) if ops else []):
                    r.wait()                                                 # NCCL work: orders the stream, does not block the host
            self.bytes_sent += sum(t.numel() for t in self.out.values())
            w.slab_unpack(self._ptr(self.inc, self.left), self._ptr(self.inc, self.right), self.capacity)

    def recut(self):
        """New cuts at the body-count quantiles of the current positions.  Collective: every rank ends up with every body's current
        state (owners' contributions summed), writes it into its world and re-classifies ownership from it."""
        t, v = self.transforms(1), self.velocities()
        self.world.write_state(t, v)
        self.cuts = quantile_cuts(t[:, self.axis].astype(np.float64), self.world_size)
        lo = self.cuts[self.rank - 1] if self.rank > 0 else -float("inf")
        hi = self.cuts[self.rank] if self.rank < self.world_size - 1 else float("inf")
        self.world.slab_configure(self.rank, self.world_size, self.axis, lo, hi, self.margin)
        self.recuts += 1

    def step_internal(self, dt, iterations=30):
        if self.world_size > 1 and self.recut_interval and self.steps and self.steps % self.recut_interval == 0:
            self.recut()
        if self.world_size > 1 and self.steps and self.steps % self.resize_interval == 0:
            self.check_capacity()
        self.exchange()
        self.world.step_internal(dt, iterations)
        self.steps += 1

    def dropped(self):
        """Bodies that did not fit this rank's last outgoing messages (must be 0: raise `capacity`)."""
        return sum(int(t[4:8].cpu().view(torch.int32)[0]) for t in self.out.values())

    def synchronize(self):
        self.world.synchronize()

    def enable_stage_timing(self, on=True):
        self.world.enable_stage_timing(on)

    def stats(self):
        return self.world.stats()

    def gather_stats(self, acc):
        """Sum the count-like statistics over ranks (stage times: max), in place on rank 0's accumulator.  Pairs and contacts between
        an owned body and a ghost exist on both sides of a cut and are counted twice (a few per cent of a slab)."""
        keys = sorted(acc.keys())
        t = torch.tensor([acc[k] for k in keys], dtype=torch.float64, device=self.dev)
        tmax = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        for i, k in enumerate(keys):
            acc[k] = float(tmax[i]) if (k.startswith("ms") or k in ("numColors", "avgColors", "avgSteps", "numRigidBodies", "numColliders", "numInternalSteps", "numGraphBuilds", "coloringRounds")) else float(t[i])

    def owned_mask(self):
        return self.world.slab_codes() == OWNED

    def transforms(self, which=1):
        """Global transforms assembled from every rank's owned bodies (collective)."""
        own = torch.as_tensor(self.owned_mask().astype(np.float32), device=self.dev)[:, None]
        t = torch.as_tensor(self.world.transforms(which), device=self.dev) * own
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def velocities(self):
        own = torch.as_tensor(self.owned_mask().astype(np.float32), device=self.dev)[:, None]
        v = torch.as_tensor(self.world.velocities(), device=self.dev) * own
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        return v.cpu().numpy()
