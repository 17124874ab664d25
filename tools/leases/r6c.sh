#!/bin/bash
# lease r6c: where the agglomeration's 13 ms go (laps between host synchronisations, kernel statistics); streamed (pread) against mapped input
O=gpurun_out/r6c; mkdir -p $O
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
{
for i in 1 2 3 4; do
  SWARM_AMD_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\[cluster|pinned|Clustering|Building" ; echo ---
done
} > $O/cluster_laps.txt 2>&1
tail -22 $O/cluster_laps.txt
KSTATS_LINES=60 tools/kstats.sh r6c_whole $PWD/swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA > $O/kstats_head.txt 2>&1
mv gpurun_out/r6c_whole_kernel_stats.csv $O/whole_run_10M_kernel_stats.csv
python - > $O/cluster_kernels.txt <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r6c/whole_run_10M_kernel_stats.csv')))
tot=0
for r in rows:
    n=r['Name']
    if any(k in n for k in ('k_label','k_level','radix_sort','k_swarm','scan_config','copyBuffer','fillBuffer')):
        print(n[:100].replace('rocprim::ROCPRIM_400200_NS::detail::',''), r['Calls'], int(r['TotalDurationNs'])/1e6); tot+=int(r['TotalDurationNs'])/1e6
print('total', tot)
PY
cat $O/cluster_kernels.txt
# streamed against mapped input: 24 runs each, alternating; reader lap + wall
{
for i in $(seq 1 24); do
  for m in pread mmap; do
    if [ $m = mmap ]; then export SWARM_AMD_INPUT=mmap; else unset SWARM_AMD_INPUT; fi
    s=${EPOCHREALTIME/./}; r=$(SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "map \+ parallel|database read and ordered|results written" | tr '\n' ' '); e=${EPOCHREALTIME/./}
    echo "$m wall_ms $(( (e - s) / 1000 )) $r"; sleep 0.6
  done
done
} > $O/input_pread_vs_mmap.txt 2>&1
unset SWARM_AMD_INPUT
for m in pread mmap; do echo $m $(grep "^$m" $O/input_pread_vs_mmap.txt | awk '{print $3}' | sort -n | tr '\n' ' '); done
grep -E "^pread|^mmap" $O/input_pread_vs_mmap.txt | head -6
