#!/bin/bash
# lease r6b: the new agglomeration (frontier walk, chained sweeps, narrow sort, pinned result arrays): parity, then where the time goes
O=gpurun_out/r6b; mkdir -p $O
(timeout 900 python -m pytest tests/test_d1_gpu.py tests/test_cli_gpu.py tests/test_fastidious_gpu.py -x -q -n 3 > $O/tests_d1_cli.txt 2>&1; tail -5 $O/tests_d1_cli.txt)
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
run() { # tag, env...
  tag=$1; shift
  for i in 1 2 3 4 5; do
    s=${EPOCHREALTIME/./}
    env "$@" SWARM_AMD_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o_$tag.txt -l /dev/null $FA 2>&1 | grep -E "^\[cluster\]|^\[t .*(pinned|Clustering|Building|Writing|written|uploaded)" | tr '\n' ';'
    e=${EPOCHREALTIME/./}; echo " wall_ms $(( (e - s) / 1000 ))"
    sleep 0.5
  done
}
{
echo "== default (pinned, jump all)"; run a X=1
echo "== SWARM_AMD_PIN_RESULTS=0"; run b SWARM_AMD_PIN_RESULTS=0
echo "== SWA_CLUSTER_JUMP=active"; run c SWA_CLUSTER_JUMP=active
md5sum /tmp/o_a.txt /tmp/o_b.txt /tmp/o_c.txt
python - <<'PY'
import json
g=json.load(open('tests/golden/fullsize.json'))
print({k:(v.get('runs',{}).get('d1',{}) or {}).get('o_md5') for k,v in g.items() if isinstance(v,dict)})
PY
} > $O/whole_run_variants.txt 2>&1
tail -30 $O/whole_run_variants.txt
KSTATS_LINES=40 tools/kstats.sh r6b_whole ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA > $O/kstats_head.txt 2>&1
mv gpurun_out/r6b_whole_kernel_stats.csv $O/whole_run_10M_kernel_stats.csv
python - > $O/cluster_kernels.txt <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r6b/whole_run_10M_kernel_stats.csv')))
tot=0
for r in rows:
    n=r['Name']
    if any(k in n for k in ('k_label','k_level','radix_sort','k_swarm','scan_config','copyBuffer','fillBuffer')):
        print(n[:100].replace('rocprim::ROCPRIM_400200_NS::detail::',''), r['Calls'], int(r['TotalDurationNs'])/1e6); tot+=int(r['TotalDurationNs'])/1e6
print('total', tot)
PY
cat $O/cluster_kernels.txt
# the distribution of 40 whole runs
{
for i in $(seq 1 40); do
  s=${EPOCHREALTIME/./}; ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA; e=${EPOCHREALTIME/./}; echo $(( (e - s) / 1000 )); sleep 0.7
done
} > $O/whole_run_40.txt 2>&1
sort -n $O/whole_run_40.txt | tr '\n' ' '
