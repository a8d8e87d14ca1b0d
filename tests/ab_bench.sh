# Developer tool: A/B of two builds of libmi_physics.so on ONE GPU box (box-to-box variance is several per cent).
# usage (on the box): bash tests/ab_bench.sh tmp_libs/old.so tmp_libs/new.so [rounds] [bench args...]
A=$1; B=$2; R=${3:-3}; shift 3 || true
for i in $(seq 1 $R); do
  for L in $A $B; do
    cp $L directx-renderer-kurth_amd/libmi_physics.so
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$L', round(d['value'],1), round(d['ms_per_step'],4), s['msCollidersBroad'], s['msNarrow'], s['msSolverSetup'], s['msSolve'])"
  done
done
cp $B directx-renderer-kurth_amd/libmi_physics.so
