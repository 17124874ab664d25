#!/bin/bash
# lease r5c — the whole run after the host-side rework (worker pool, arrays sized beside the sort, pieces unmapped and freed
# by their threads, lean clustering results, writer beside the formatters, launcher instead of re-exec): timeline by run,
# sort threads, host allocation modes; the sample sort alone by thread count; then the suites the rework touches
O=gpurun_out/r5c; mkdir -p $O
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
cat $FA > /dev/null
run() {   # label, env...
  local label=$1; shift
  for i in 1 2 3; do
    echo "---- $label run $i"
    sleep 1
    s=${EPOCHREALTIME/./}
    env "$@" SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\["
    e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"
  done
}
{
run default X=1
md5sum /tmp/o.txt
run sort64 SWARM_AMD_SORT_THREADS=64
run sort16 SWARM_AMD_SORT_THREADS=16
run alloc_thp SWARM_AMD_HOST_ALLOC=1
run alloc16 SWARM_AMD_HOST_ALLOC=16
run pool32 SWARM_AMD_HOST_THREADS=32
echo "---- quiet (no timing output), 5 runs"
for i in 1 2 3 4 5; do sleep 1; s=${EPOCHREALTIME/./}; ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; done
echo "---- all outputs (-o -s -i -w -j), 2 runs"
for i in 1 2; do sleep 1; s=${EPOCHREALTIME/./}; SWARM_AMD_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o2.txt -s /tmp/s2.txt -i /tmp/i2.txt -w /tmp/w2.txt -j /tmp/j2.txt -l /dev/null $FA 2>&1 | grep -E "^\[t"; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; done
md5sum /tmp/o2.txt /tmp/s2.txt /tmp/j2.txt
} > $O/runs.txt 2>&1
{ for t in 8 16 32 64 128; do echo "threads $t: $(tools/experiments/sort_bench $t)"; done; } > $O/sort_bench.txt 2>&1
timeout 1200 python -m pytest tests/test_cli_gpu.py tests/test_d1_gpu.py tests/test_fastidious_gpu.py tests/test_multi_gpu.py tests/test_ref_gpu.py tests/test_guard_gpu.py tests/test_derep.py tests/test_dn_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "not 100" > $O/tests_fullsize.txt 2>&1
tail -3 $O/tests.txt $O/tests_fullsize.txt; cat $O/sort_bench.txt
