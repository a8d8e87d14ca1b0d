"""bench.py — physics steps/sec of the MI355X rigid-body stepper on BASELINE.json's headline config.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c1|c3_mid] [--no-cpu-baseline]

N = 1: the workload is config 3 ("100k mixed colliders (sphere/capsule/OBB) + contacts, 1 MI355X"): 100 000 bodies poured as a
dense block, settled for 240 steps (untimed), then W warm-up steps and K timed steps of one physicsStepInternal each (dt = 1/120 s,
30 solver iterations), state resident in HBM, no host read-back inside the timed region.
N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the same world cut into N spatial slabs with a ghost-body halo
exchange per step (directx-renderer-kurth_amd/parallel.py); "scaling": "strong".

One JSON line on rank 0.  `roofline` prices the dominant kernel (contact solve: k_cl_solve, the LDS cluster sweep, all 30 iterations
in one launch) with the fixed algorithmic figure of BASELINE.md (240 B per contact per iteration) against HIP-event time measured on
the world's stream inside the timed region; `stage_roofline` does the same per stage with SURVEY section 8(d)'s per-unit bytes;
`cpu_baseline` times the oracle's 8-lane "AVX2 restatement" of the reference (8-wide sweep, greedy batch scheduler, 8-wide solver; single thread) on a bounded sample that starts
from the state at the BEGINNING of the timed window.  After the timed region the state is checked (finite, nothing below the ground,
no solver recovery): a number measured on a broken simulation is not printed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGORITHMIC_BYTES_PER_CONTACT_ITERATION = 240.0  # BASELINE.md / SURVEY §8(d)
HBM_PEAK_GBPS = 8000.0                            # MI355X_MICROARCH.md: 8.0 TB/s spec
WORKLOADS = {"c1": ("64 OBBs on a ground plane", 120), "c2": ("10k stacked spheres", 240), "c3": ("100k mixed colliders (sphere/capsule/OBB)", 240),
             "c3_mid": ("20k mixed colliders", 240), "c4": ("256 ragdolls (hinge + cone-twist chains)", 120), "c5": ("1M mixed colliders", 240)}


def cpu_baseline(scene, transforms, velocities, first_step, seconds=12.0, max_steps=40):
    """Oracle 8-lane path (liboracle_avx2.so, -O3 -mavx2 -mfma), one thread, started from the device's state at the first timed step."""
    from oracle import oracle as orc
    w = scene.instantiate(orc.OracleWorld(avx2=True, solver=orc.SOLVER_WIDE8))
    w.set_wide_broadphase(True)     # the reference's 8-wide sweep (collision_broad.cpp:168-295), like its 8-wide solver
    w.write_state(transforms, velocities, presort=True)
    w.step_internal(scene.dt)  # untimed: first broadphase after the state injection (an insertion sort from scratch)
    w.stage_seconds(reset=True)
    n = 0
    t0 = time.perf_counter()
    while n < max_steps and (n < 2 or time.perf_counter() - t0 < seconds):
        w.step_internal(scene.dt)
        n += 1
    dt = time.perf_counter() - t0
    stages = w.stage_seconds()
    contacts = len(w.contacts()[0])
    return {"value": n / dt, "unit": "steps/s", "cores": 1, "kind": "port",
            "stage_ms": dict(zip(("msCollidersBroad", "msNarrow", "msSolverSetup", "msSolve", "msIntegrate"), [round(float(x) / n * 1e3, 2) for x in stages])),
            "sample": "%d steps from the state at timed step %d (%d contacts at the end), oracle 8-lane AVX2 restatement of the reference solver incl. its greedy batch scheduler, "
                      "8-wide sort-and-sweep broadphase (the reference's determineOverlapsSIMD), 1 thread; the reference binary itself cannot be built (MSVC/Windows) and its u16 indices cannot hold this config" % (n, first_step, contacts)}


def pmc_traffic(contacts):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes of this round (profiles/), scaled to this run's
    contact count: a STATIC figure, labelled so (the counters need their own rocprofv3 passes and cannot be read inside this run)."""
    path = os.path.join(ROOT, "profiles", "r03_pmc_k_cl_solve.json")
    try:
        with open(path) as f:
            p = json.load(f)
        per_contact = (p["fetch_bytes_per_launch"] + p["write_bytes_per_launch"]) / p["contacts_per_step"]
        return per_contact * contacts, "static: %s (%d contacts), scaled by contact count" % (os.path.relpath(path, ROOT), p["contacts_per_step"])
    except (OSError, KeyError, ValueError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--settle", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus, args.gpus))
        args.gpus = world_size

    import torch
    import directx_renderer_kurth_amd as mi
    from directx_renderer_kurth_amd import scenes

    label, default_settle = WORKLOADS[args.workload]
    settle = default_settle if args.settle < 0 else args.settle
    scene = scenes.by_name(args.workload)
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))

    if world_size > 1:
        import torch.distributed as dist
        from directx_renderer_kurth_amd import parallel
        # RCCL over xGMI; MI_BENCH_BACKEND=gloo (messages staged through host memory) only exists to rehearse this path with
        # several ranks on ONE GPU, where RCCL refuses duplicate devices
        backend = os.environ.get("MI_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend)
        n_dev = torch.cuda.device_count()
        if local_rank >= n_dev:
            local_rank = local_rank % max(1, n_dev)
            torch.cuda.set_device(local_rank)
        stepper = parallel.SlabWorld(scene, device=local_rank, rank=rank, world_size=world_size, comm_on_cpu=(backend == "gloo"))
        barrier = dist.barrier
    else:
        dist = None
        stepper = scene.instantiate(mi.World(device=local_rank))
        barrier = lambda: None

    def sync():
        stepper.synchronize()
        torch.cuda.synchronize()

    for _ in range(settle):
        stepper.step_internal(scene.dt)
    for _ in range(args.warmup):
        stepper.step_internal(scene.dt)
    first = stepper.stats()            # clears the running means; counts of the last warm-up step
    start_state = (stepper.transforms(1), stepper.velocities()) if (world_size == 1 and not args.no_cpu_baseline) else None
    stepper.enable_stage_timing(True)  # HIP events on the world's stream around every stage of every timed step (read back in batches, no per-step stall)
    barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stepper.step_internal(scene.dt)
    barrier(); sync()
    elapsed = time.perf_counter() - t0
    st = stepper.stats()               # stage times and counts: means over exactly the timed steps
    stepper.enable_stage_timing(False)
    assert st["avgSteps"] == args.steps, st["avgSteps"]
    acc = {k: v * args.steps for k, v in st.items() if isinstance(v, (int, float))}
    for k, a in (("numContacts", "avgContacts"), ("numCollisions", "avgCollisions"), ("numColors", "avgColors"), ("numBroadphaseOverlaps", "avgBroadphaseOverlaps"), ("flowProbes", "avgFlowProbes")):
        acc[k] = st[a] * args.steps

    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        stepper.gather_stats(acc)

    # the state that was timed must be a sane simulation: finite, nothing through the ground (static AABB top at y = 0), no redone steps
    import numpy as np
    tr_end = stepper.transforms(1)
    assert np.isfinite(tr_end).all() and np.isfinite(stepper.velocities()).all(), "non-finite state after the timed region"
    assert float(tr_end[:, 1].min()) > -0.5, "a body fell through the ground: y_min = %g" % float(tr_end[:, 1].min())
    rehearsal = world_size > 1 and os.environ.get("MI_BENCH_BACKEND", "nccl") == "gloo"  # several ranks sharing ONE GPU: their persistent sweeps are not all resident and time out on each other
    assert rehearsal or st["numFlowRecoveries"] == first["numFlowRecoveries"], "%d steps of the timed region were redone by the fallback sweep (the cluster sweep gave up): not a measurement of the production path" % (st["numFlowRecoveries"] - first["numFlowRecoveries"])

    if rank == 0:
        K = args.steps
        mean = {k: v / K for k, v in acc.items()}
        ms_per_step = elapsed / K * 1e3
        contacts = mean["numContacts"]
        cluster = st["clusterTasks"][0] > 0 if "clusterTasks" in st else False  # cluster sweep: all 30 iterations in ONE launch of k_cl_solve
        launches_per_step = 1.0 if cluster else max(1.0, mean["numColors"]) * 30.0
        bytes_per_step = ALGORITHMIC_BYTES_PER_CONTACT_ITERATION * contacts * 30.0
        # joints solved by the same launch (config 4): SURVEY section 8(d)'s per joint-iteration figures (row + ids + 2 x (64 B body read + 24 B write))
        joint_bytes = {"hinge": 196 + 8 + 2 * (64 + 24), "cone_twist": 245 + 8 + 2 * (64 + 24)}
        joint_counts = {k: sum(1 for j in scene.joints if (j[0][:-6] if j[0].endswith("_local") else j[0]) == k) for k in joint_bytes}
        joint_bytes_per_step = 30.0 * sum(joint_bytes[k] * n for k, n in joint_counts.items())
        bytes_per_step += joint_bytes_per_step
        solve_s = mean["msSolve"] * 1e-3
        achieved = bytes_per_step / solve_s / 1e9 if solve_s > 0 else 0.0
        traffic, traffic_src = pmc_traffic(contacts) if cluster else (None, None)
        nb, nc, pairs, manifolds = scene.num_bodies, scene.num_bodies + 1, mean["numBroadphaseOverlaps"], mean["numCollisions"]
        # SURVEY section 8(d) per-unit algorithmic bytes, against the stage's HIP-event time (an upper bound of its kernels' time)
        stage_bytes = {
            "msCollidersBroad": (64 + 28 + 88) * nc + 24 * 27 * nc,          # collider build + ~27 candidate AABBs per collider
            "msNarrow": (2 * 64 + 96) * pairs,                               # two collider records in, one manifold record out per candidate pair
            "msSolverSetup": (32 + 4 + 2 * 104 + 120) * contacts + (140 + 104) * nb,  # contact rows + force integration
            "msIntegrate": (104 + 52) * nb,
        }
        out = {
            "metric": "physics steps/sec at 100k rigid bodies" if args.workload == "c3" else "physics steps/sec", "value": K / elapsed, "unit": "steps/s",
            "n_gpus": world_size, "steps": K, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s, dt 1/%d s, 30 solver iterations" % (args.workload, label, round(1.0 / scene.dt)),
                       "timed_steps": [settle + args.warmup, settle + args.warmup + K], "contacts_first_last": [int(first.get("numContacts", 0)), int(st["numContacts"])],
                       "bodies": scene.num_bodies, "broadphase_pairs": round(pairs), "manifolds": round(manifolds),
                       "contacts": round(contacts), "colors": round(mean["numColors"], 1), "joints": round(mean["numJoints"]),
                       "solver": ("cluster sweep: tasks per phase %s, %d bodies handed between tasks" % (st["clusterTasks"], st["clusterSharedBodies"])) if cluster else "launch sweep",
                       "recoveries": int(st["numFlowRecoveries"]),
                       "parallelism": "1 gpu" if world_size == 1 else "%d spatial slabs + ghost-body halo (%s)" % (world_size, "RCCL over xGMI" if os.environ.get("MI_BENCH_BACKEND", "nccl") == "nccl" else "gloo rehearsal")},
            "stage_ms": {k: round(mean[k], 4) for k in ("msCollidersBroad", "msNarrow", "msSolverSetup", "msSolve", "msIntegrate", "msTotal")},
            "roofline": {"bound": "hbm", "kernel": "k_cl_solve (contact PGS sweep, 30 iterations in one launch, LDS clusters)" if cluster else "k_solve_color (contact PGS sweep, one launch per colour and iteration)", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per_step / launches_per_step, "avg_launch_us": solve_s / launches_per_step * 1e6,
                         "launches_per_step": launches_per_step,
                         "joint_bytes_per_launch": joint_bytes_per_step / launches_per_step, "joints_priced": joint_counts},
            "stage_roofline": {k: {"algorithmic_bytes": round(b), "GBps": round(b / (mean[k] * 1e-3) / 1e9, 1) if mean[k] > 0 else 0.0, "frac": round(b / (mean[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if mean[k] > 0 else 0.0}
                               for k, b in stage_bytes.items()},
        }
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, start_state[0], start_state[1], settle + args.warmup)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
