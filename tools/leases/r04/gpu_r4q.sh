#!/bin/bash
# round 4, lease q: what LDS instructions cost (ubench_lds)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4q; mkdir -p $O
cd $R
timeout 120 tools/experiments/ubench_lds > $O/ubench_lds.jsonl 2> $O/ubench_lds.err
cat $O/ubench_lds.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['op'].ljust(16), d['in_flight'], str(d['active_lanes']).rjust(3), d['ms'], d['ns_per_wave_instruction_per_cu'])
"
