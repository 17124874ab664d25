#!/bin/bash
# lease r5e — why the host phases of a whole run vary from run to run (19..96 ms for the same bucket sorts): kernel
# counters around a run (NUMA balancing, THP, compaction), the reader without the GPU start-up beside it, placements
O=gpurun_out/r5e; mkdir -p $O
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
cat $FA > /dev/null
VM="numa_hint_faults numa_hint_faults_local numa_pages_migrated pgmigrate_success numa_pte_updates thp_fault_alloc thp_fault_fallback thp_collapse_alloc compact_stall compact_fail pgfault pgmajfault pgalloc_normal pgfree allocstall_normal"
snap() { for k in $VM; do echo "$k $(awk -v k=$k '$1==k{print $2}' /proc/vmstat)"; done; }
run() {   # label, env... (3 runs, vmstat difference around each)
  local label=$1; shift
  for i in 1 2 3; do
    echo "---- $label run $i"
    sleep 1
    snap > /tmp/vm0
    s=${EPOCHREALTIME/./}
    env "$@" SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\["
    e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"
    snap > /tmp/vm1
    echo "vmstat: $(paste /tmp/vm0 /tmp/vm1 | awk '{d=$4-$2; if (d!=0) printf "%s=%d ", $1, d}')"
  done
}
{
echo "numa_balancing $(cat /proc/sys/kernel/numa_balancing 2>&1)  zone_reclaim $(cat /proc/sys/vm/zone_reclaim_mode 2>&1)  overcommit $(cat /proc/sys/vm/overcommit_memory)"
cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/cpu.max 2>&1 | head -3
grep -E "Cpus_allowed_list|Mems_allowed_list" /proc/self/status
run default X=1
run gpu_after_read SWARM_AMD_GPU_AFTER_READ=1
run gpu_after_read_alloc0 SWARM_AMD_GPU_AFTER_READ=1 SWARM_AMD_HOST_ALLOC=0
run free_at_exit SWARM_AMD_FREE_AT_EXIT=1
N0=$(cat /sys/devices/system/node/node0/cpulist)
echo "==== taskset node0 32 threads"
for i in 1 2 3; do sleep 1; snap > /tmp/vm0; s=${EPOCHREALTIME/./}; SWARM_AMD_HOST_THREADS=32 OMP_NUM_THREADS=32 SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 taskset -c $N0 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\["; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; snap > /tmp/vm1; echo "vmstat: $(paste /tmp/vm0 /tmp/vm1 | awk '{d=$4-$2; if (d!=0) printf "%s=%d ", $1, d}')"; done
echo "==== taskset cores 0-31 only (no SMT siblings), 32 threads"
for i in 1 2 3; do sleep 1; s=${EPOCHREALTIME/./}; SWARM_AMD_HOST_THREADS=32 OMP_NUM_THREADS=32 SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 taskset -c 0-47 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\["; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; done
} > $O/runs.txt 2>&1
tail -n 40 $O/runs.txt
