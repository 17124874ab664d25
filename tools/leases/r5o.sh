#!/bin/bash
# lease r5o — is the long process exit the cgroup's CPU throttling?  per run: throttle events and throttled time of the
# cgroup around the run, by thread count
O=gpurun_out/r5o; mkdir -p $O
FA=/tmp/swa_bench_10000000x150_s1.fa
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
cat $FA > /dev/null
st() { awk '/nr_throttled/{a=$2} /throttled_usec/{b=$2} /usage_usec/{c=$2} END{print a, b, c}' /sys/fs/cgroup/cpu.stat; }
run() {
  local label=$1; shift
  echo "---- $label"
  for i in 1 2 3 4 5 6 7 8; do
    sleep 1; read t0 u0 c0 < <(st); s=${EPOCHREALTIME/./}
    env "$@" SWARM_AMD_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "results written|context created|read and ordered" | grep "^\[t" | tr '\n' ' '
    e=${EPOCHREALTIME/./}; read t1 u1 c1 < <(st)
    echo " wall_ms $(( (e - s) / 1000 )) throttles $((t1 - t0)) throttled_ms $(( (u1 - u0) / 1000 )) cpu_ms $(( (c1 - c0) / 1000 ))"
  done
}
{
cat /sys/fs/cgroup/cpu.max
run default X=1
run threads12 SWARM_AMD_HOST_THREADS=12 OMP_NUM_THREADS=12
run threads8 SWARM_AMD_HOST_THREADS=8 OMP_NUM_THREADS=8

} > $O/runs.txt 2>&1
cat $O/runs.txt | cut -c1-260
