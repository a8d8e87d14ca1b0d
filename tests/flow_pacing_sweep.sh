#!/bin/bash
# Developer tool: the bench workload and the small configurations under the current poll pacing defaults (override with MI_FLOW_HOP / MI_FLOW_HOP_LARGE).
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('c3 bench ms/step %.3f' % d['ms_per_step'], 'solve %.3f' % d['stage_ms']['msSolve'], 'steps/s %.1f' % d['value'])"
for sc in c2 c4 c1 c3_mid; do timeout -k 10 100 python tests/quick_bench.py $sc 120 300 2>&1 | grep "ms/step"; done
