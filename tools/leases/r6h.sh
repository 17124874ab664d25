#!/bin/bash
# lease r6h: agglomeration with chained level sweeps + the download path warmed beside the start-up; parity, laps, 40 whole runs
O=gpurun_out/r6h; mkdir -p $O
(timeout 900 python -m pytest tests/test_d1_gpu.py tests/test_cli_gpu.py -x -q -n 3 > $O/tests_d1_cli.txt 2>&1; tail -3 $O/tests_d1_cli.txt)
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
{
for mode in X=1 SWARM_AMD_WARM_DOWNLOADS=0 X=1 SWARM_AMD_WARM_DOWNLOADS=0; do
for i in 1 2 3; do
  echo "== $mode"
  s=${EPOCHREALTIME/./}
  env $mode SWARM_AMD_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\[cluster|pinned|warm|Clustering|Building|uploaded|read and ordered|code objects|written"
  e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"
done
done
md5sum /tmp/o.txt
} > $O/cluster_laps.txt 2>&1
grep -E "==|device \+|download order|generations|label sweeps|sort|wall_ms|usable|warm|uploaded" $O/cluster_laps.txt | head -80
{
for i in $(seq 1 40); do
  s=${EPOCHREALTIME/./}; ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA; e=${EPOCHREALTIME/./}; echo $(( (e - s) / 1000 )); sleep 0.7
done
} > $O/whole_run_40.txt 2>&1
sort -n $O/whole_run_40.txt | tr '\n' ' '
