"""Developer tool: profiles/r03_* from what tests/run_prof_r03.sh left under gpurun_out/r03 (run here after the GPU call)."""
import json, os, re, shutil
R, P = "gpurun_out/r03", "profiles"
for name in ("bench_c3_driver", "bench_c3_default", "bench_c4"):
    shutil.copy(os.path.join(R, name + ".json"), os.path.join(P, "r03_%s.json" % name))
shutil.copy(os.path.join(R, "stats", "c3_kernel_stats.csv"), os.path.join(P, "r03_c3_kernel_stats.csv"))
drv = json.load(open(os.path.join(R, "bench_c3_driver.json")))
prof_line = [l for l in open(os.path.join(R, "stats.log")) if l.startswith('{"metric"')][-1].strip()
with open(os.path.join(P, "r03_c3_rocprof_stats.txt"), "w") as f:
    f.write("rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline   (config 3, 100k bodies: the driver's command)\n")
    f.write("All 265 steps of the process (240 settle + 5 warm-up + 20 timed):\n")
    f.write(open(os.path.join(R, "c3_kernel_summary.txt")).read())
    f.write("\nTimed window only (tests/trace_gaps.py on the kernel trace, last 20 steps = steps 245..265):\n")
    f.write(open(os.path.join(R, "c3_last20.txt")).read())
    f.write("\nbench line of that (profiled) run:\n" + prof_line + "\n")
    f.write("\nbench line of the same command without the profiler (profiles/r03_bench_c3_driver.json): %.1f steps/s, %.3f ms/step, stage_ms %s\n" % (drv["value"], drv["ms_per_step"], drv["stage_ms"]))
shutil.copy(os.path.join(R, "c4_kernel_summary.txt"), os.path.join(P, "r03_c4_kernel_summary.txt"))
vals = {}
for l in open(os.path.join(R, "pmc_summary.txt")):
    m = re.match(r"(\S+)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)", l)
    if m:
        vals[(m.group(1), m.group(2))] = float(m.group(4))
fetch_kb, write_kb = vals[("k_cl_solve", "FETCH_SIZE")], vals[("k_cl_solve", "WRITE_SIZE")]
contacts = drv["config"]["contacts"]
fb, wb = fetch_kb * 1024 * 2.0, write_kb * 1024
per = (fb + wb) / contacts / 30.0
with open(os.path.join(P, "r03_pmc_k_cl_solve.txt"), "w") as f:
    f.write("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, no trace domains) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline\n")
    f.write("k_cl_solve, last 20 dispatches (steps 245..265, %d contacts per step on average).  Counter unit: KB per dispatch.\n" % contacts)
    f.write(open(os.path.join(R, "pmc_summary.txt")).read())
    f.write("\ngfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests as 64 B for 16-B-per-lane loads (what this kernel issues: float4 rows in the prologue, b128 hand-over polls) -> x2; WRITE_SIZE exact.\n")
    f.write("k_cl_solve: fetch = %.1f KB x 1024 x 2 = %.1f MB, write = %.1f KB x 1024 = %.1f MB per launch: %.1f MB = %.1f B per contact and iteration,\n" % (fetch_kb, fb / 1e6, write_kb, wb / 1e6, (fb + wb) / 1e6, per))
    f.write("against the algorithmic 240 B per contact and iteration (%.0f MB per launch at this contact count): traffic / algorithmic = %.2f.\n" % (240.0 * contacts * 30 / 1e6, per / 240.0))
    f.write("(The rows live in registers, the bodies in LDS for all 30 iterations; what moves per iteration is the hand-over of the bodies shared between tasks, 2 x 16 B published and polled per body and phase.)\n")
json.dump({"kernel": "k_cl_solve", "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no trace domains), last 20 dispatches of bench.py --steps 20 --warmup 5 --no-cpu-baseline (steps 245..265)",
           "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb, "gfx950_fetch_correction": 2.0, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "contacts_per_step": contacts},
          open(os.path.join(P, "r03_pmc_k_cl_solve.json"), "w"), indent=1)
with open(os.path.join(P, "r03_pmc_stage_kernels.txt"), "w") as f:
    f.write("rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS, then --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum\n")
    f.write("(two separate passes, no trace domains) -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline; per kernel, mean over its last 10 dispatches (SQ cycle counters in quad-cycles)\n")
    for l in open(os.path.join(R, "pmc_stage_kernels.txt")):
        if re.match(r"(kernel|k_pairs|k_epa|k_narrow|k_gjk|k_cl_|k_contact_init|k_active_list|k_classify|k_build_colliders)", l):
            f.write(l)
print("profiles written; driver window %.1f steps/s, traffic %.1f B per contact-iteration" % (drv["value"], per))
