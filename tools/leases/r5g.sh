#!/bin/bash
# lease r5g — after the CPU diet: identifier check by partitioned L2 tables, AVX2 packing, prefetching writer (CPU seconds per phase)
O=gpurun_out/r5g; mkdir -p $O
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
cat $FA > /dev/null
run() {
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
cat /sys/fs/cgroup/cpu.max; grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat
run default X=1
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat


run noavx SWARM_AMD_NO_AVX2=1
run threads64 SWARM_AMD_HOST_THREADS=64 OMP_NUM_THREADS=32
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat
echo "---- quiet, 5 runs"
for i in 1 2 3 4 5; do sleep 1; s=${EPOCHREALTIME/./}; ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; done
md5sum /tmp/o.txt
echo "---- the reference, -t 16"
s=${EPOCHREALTIME/./}; oracle/_ref/swarm -d 1 -t 16 -o /tmp/ro.txt -l /dev/null $FA; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; md5sum /tmp/ro.txt
} > $O/runs.txt 2>&1
grep -E "wall_ms|^----|thrott" $O/runs.txt | head -60
timeout 600 python -m pytest tests/test_cli_gpu.py tests/test_fastidious_gpu.py -x -q -m gpu > gpurun_out/r5g/tests.txt 2>&1; grep -E "passed|failed" gpurun_out/r5g/tests.txt
