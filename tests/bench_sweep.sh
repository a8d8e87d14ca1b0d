mkdir -p gpurun_out/r02
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r02/bench_now.json 2>/dev/null || exit 1
python - <<PY
import json
j=json.load(open("gpurun_out/r02/bench_now.json"))
print(round(j["ms_per_step"],4), j["stage_ms"], j["config"]["solver"], j["config"]["recoveries"])
PY
