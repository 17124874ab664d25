#!/bin/bash
# lease r5r — the reader's remaining serial costs: input mapping kept with the handle; first touch by the 16 workers instead
# of populating (SWARM_AMD_HOST_ALLOC=0); buckets of the sample sort
O=gpurun_out/r5r; mkdir -p $O
FA=/tmp/swa_bench_10000000x150_s1.fa
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
cat $FA > /dev/null
run() {
  local label=$1; shift
  for i in 1 2 3 4; do
    echo "---- $label run $i"; sleep 1; s=${EPOCHREALTIME/./}
    env "$@" SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\[hostdb|read and ordered|results written"
    e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"
  done
}
{
run default X=1
run alloc0 SWARM_AMD_HOST_ALLOC=0
run buckets512 SWARM_AMD_SORT_BUCKETS=512
run buckets2048 SWARM_AMD_SORT_BUCKETS=2048
md5sum /tmp/o.txt
} > $O/runs.txt 2>&1
timeout 300 python -m pytest tests/test_cli_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
grep -E "passed|failed" $O/tests.txt; grep -E "^----|read and ordered|buckets|sized|scratch|wall_ms" $O/runs.txt | cut -c1-100 | tr '\n' ' ' | sed 's/----/\n----/g'
