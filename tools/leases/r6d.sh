#!/bin/bash
# lease r6d: download cost by kind of host memory; kernel statistics of the whole run with the new agglomeration
O=gpurun_out/r6d; mkdir -p $O
tools/experiments/d2h_cost 40 > $O/d2h_cost_40MB.txt 2>&1; HSA_ENABLE_SDMA=0 tools/experiments/d2h_cost 40 > $O/d2h_cost_40MB_no_sdma.txt 2>&1
cat $O/d2h_cost_40MB.txt; echo "== no sdma"; grep -E "copy 1|copy 2|kernel store" $O/d2h_cost_40MB_no_sdma.txt | head -30
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
SWARM_AMD_FULL_TEARDOWN=1 KSTATS_LINES=70 tools/kstats.sh r6d_whole $PWD/swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA > $O/kstats_head.txt 2>&1
mv gpurun_out/r6d_whole_kernel_stats.csv $O/whole_run_10M_kernel_stats.csv
python - > $O/cluster_kernels.txt <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r6d/whole_run_10M_kernel_stats.csv')))
tot=0
for r in rows:
    n=r['Name']
    if any(k in n for k in ('k_label','k_level','radix_sort','k_swarm','scan_config','copyBuffer','fillBuffer')):
        print(n[:100].replace('rocprim::ROCPRIM_400200_NS::detail::',''), r['Calls'], int(r['TotalDurationNs'])/1e6, r['MinNs'], r['MaxNs']); tot+=int(r['TotalDurationNs'])/1e6
print('total', tot)
PY
cat $O/cluster_kernels.txt
