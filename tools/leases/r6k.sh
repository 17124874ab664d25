#!/bin/bash
# lease r6k: host-side diet of the -f and d >= 2 runs (parallel role flags, no serial zero fills, parallel d >= 2 swarms writer): parity, then the timelines of r6j again
O=gpurun_out/r6k; mkdir -p $O
(timeout 1200 python -m pytest tests/test_dn_gpu.py tests/test_cli_gpu.py tests/test_fastidious_gpu.py tests/test_derep.py tests/test_ref_gpu.py -x -q -n 3 > $O/tests.txt 2>&1; tail -4 $O/tests.txt)
bash tools/leases/r6j.sh > /dev/null 2>&1
cp gpurun_out/r6j/config2_timeline.txt $O/; cp gpurun_out/r6j/config3_timeline.txt $O/
grep -E "==|Clustering|Counting|Checking|Grafting|Writing|written|wall_ms|swarm table|device \+" $O/config2_timeline.txt
grep -E "==|graph: resident|walk on|swarm tables|Clustering|written|wall_ms" $O/config3_timeline.txt
