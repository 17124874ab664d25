#!/bin/bash
# lease r5d — the whole run with the database kept in file order on the host (pointers for db order, the GPU gathers the words,
# staged from the helper thread while the reader sorts); the suites the rework touches
O=gpurun_out/r5d; mkdir -p $O
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


run alloc_thp SWARM_AMD_HOST_ALLOC=1

run pool32 SWARM_AMD_HOST_THREADS=32
run alloc0 SWARM_AMD_HOST_ALLOC=0
echo "---- quiet (no timing output), 5 runs"
for i in 1 2 3 4 5; do sleep 1; s=${EPOCHREALTIME/./}; ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; done
echo "---- all outputs (-o -s -i -w -j), 2 runs"
for i in 1 2; do sleep 1; s=${EPOCHREALTIME/./}; SWARM_AMD_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o2.txt -s /tmp/s2.txt -i /tmp/i2.txt -w /tmp/w2.txt -j /tmp/j2.txt -l /dev/null $FA 2>&1 | grep -E "^\[t"; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; done
md5sum /tmp/o2.txt /tmp/s2.txt /tmp/j2.txt
} > $O/runs.txt 2>&1
timeout 1200 python -m pytest tests/test_cli_gpu.py tests/test_d1_gpu.py tests/test_fastidious_gpu.py tests/test_multi_gpu.py tests/test_ref_gpu.py tests/test_guard_gpu.py tests/test_derep.py tests/test_dn_gpu.py tests/test_bench_contract.py -x -q -m gpu > $O/tests.txt 2>&1
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "not 100" > $O/tests_fullsize.txt 2>&1
