#!/bin/bash
# lease r5b — host side of the whole run on the box's CPUs: NUMA layout, the reader's phases in detail against thread count,
# OpenMP team size, GPU start-up alone / beside the reader, placement on one socket
O=gpurun_out/r5b; mkdir -p $O
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
cat $FA > /dev/null
{
echo "== numa"; ls /sys/devices/system/node/ | grep node; for n in /sys/devices/system/node/node*; do echo "$n: $(cat $n/cpulist)  $(grep MemFree $n/meminfo)"; done
lscpu | grep -E "Model name|Socket|Thread|Core|NUMA|MHz" ; which taskset numactl perf strace 2>&1
cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | grep -E "cpu_cores_count|simd_count" | head -20
ls /sys/bus/pci/devices/*/numa_node 2>/dev/null | head -0; for d in /sys/class/drm/card*/device; do echo "$d numa $(cat $d/numa_node 2>/dev/null)"; done
} > $O/numa.txt 2>&1
run() {   # label, env...
  local label=$1; shift
  for i in 1 2; do
    echo "---- $label run $i"
    s=${EPOCHREALTIME/./}
    env "$@" SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\["
    e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"
    sleep 0.5
  done
}
{
run default X=1
run serial_init SWARM_AMD_SERIAL_INIT=1
run threads16 SWARM_AMD_HOST_THREADS=16 OMP_NUM_THREADS=16
run threads32 SWARM_AMD_HOST_THREADS=32 OMP_NUM_THREADS=32
run omp64 OMP_NUM_THREADS=64
run omp32 OMP_NUM_THREADS=32
run alloc0 SWARM_AMD_HOST_ALLOC=0
run alloc_thp SWARM_AMD_HOST_ALLOC=1
if which taskset > /dev/null; then
  N0=$(cat /sys/devices/system/node/node0/cpulist)
  echo "---- taskset node0 ($N0)"
  for i in 1 2; do s=${EPOCHREALTIME/./}; SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 taskset -c $N0 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\["; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; sleep 0.5; done
  echo "---- taskset node0, 32 threads"
  for i in 1 2; do s=${EPOCHREALTIME/./}; SWARM_AMD_HOST_THREADS=32 OMP_NUM_THREADS=32 SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 taskset -c $N0 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\["; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; sleep 0.5; done
fi
md5sum /tmp/o.txt
} > $O/runs.txt 2>&1
# the GPU side, while the box is here: the d1 suites on the new build (list-region, flag and occupancy fixes)
timeout 900 python -m pytest tests/test_d1_gpu.py tests/test_stream_gpu.py tests/test_lengths_gpu.py tests/test_guard_gpu.py tests/test_bench_contract.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt; cat $O/numa.txt | head -40
