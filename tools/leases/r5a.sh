#!/bin/bash
# lease r5a — where a whole run's time goes outside the kernels: HIP start-up call by call, process exit against
# host / device memory held, THP availability; the new sort; whole-run timeline; baseline bench of the round's first build
O=gpurun_out/r5a; mkdir -p $O
{
echo "== box"; nproc; free -g | head -2; cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag /sys/kernel/mm/transparent_hugepage/khugepaged/defrag 2>&1
uname -r; rocm-smi --showserial 2>/dev/null | grep -i serial | head -2; ls /sys/class/kfd/kfd/topology/nodes | wc -l
E=tools/experiments
echo "== init (plain binary, then one that links librccl)"
for i in 1 2 3; do $E/init_cost 2>&1 | tr '\n' ' '; echo; done
for i in 1 2; do $E/init_cost_rccl 2>&1 | tr '\n' ' '; echo; done
echo "== init, env variants"
for v in "HIP_VISIBLE_DEVICES=0" "ROCR_VISIBLE_DEVICES=0" "HSA_ENABLE_SDMA=0" "GPU_MAX_HW_QUEUES=1" "HSA_ENABLE_INTERRUPT=0" "HIP_INITIAL_DM_SIZE=0" "HSA_NO_SCRATCH_RECLAIM=1" "AMD_DIRECT_DISPATCH=0" "HSA_DISABLE_CACHE=1"; do
  echo "-- $v"; env $v $E/init_cost 2>&1 | tr '\n' ' '; echo
done
echo "== exit cost: host MB, device MB, thp, nohip -> wall seconds of the whole process and of its last stamp"
for cfg in "0 0 0 0" "0 0 0 1" "4096 0 0 1" "4096 0 1 1" "4096 0 0 0" "4096 0 1 0" "0 3000 0 0" "4096 3000 0 0" "4096 3000 1 0"; do
  for i in 1 2; do
    s=${EPOCHREALTIME/./}; out=$($E/init_cost $cfg 2>&1 | tail -1); e=${EPOCHREALTIME/./}
    echo "cfg [$cfg]  wall_ms $(( (e - s) / 1000 ))  last: $out"
  done
done
} > $O/system.txt 2>&1
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
{
for mode in default "SWARM_AMD_HOST_ALLOC=1" "SWARM_AMD_HOST_ALLOC=0"; do
  for i in 1 2 3; do
    echo "---- $mode run $i"
    s=${EPOCHREALTIME/./}
    env ${mode/default/X=1} SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o_$i.txt -l /dev/null $FA 2>&1 | grep -E "^\[" 
    e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"
  done
done
md5sum /tmp/o_1.txt /tmp/o_3.txt; python - <<'PY'
import json; print(json.load(open('tests/golden/fullsize.json'))['10000000'])
PY
} > $O/whole_run.txt 2>&1
timeout 900 python -m pytest tests/test_multi_gpu.py tests/test_cli_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
tail -5 $O/tests.txt; tail -c 1500 $O/whole_run.txt
