#!/bin/bash
# lease r5i — what process exit costs after the run, and what moves out of it: word pools / scratch released in slices beside
# the GPU's work, the context and the HIP runtime released beside the writing; 8 runs per variant (the exit is bimodal)
O=gpurun_out/r5i; mkdir -p $O
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
cat $FA > /dev/null
run() {
  local label=$1; shift
  echo "---- $label"
  for i in 1 2 3 4 5 6 7 8; do
    sleep 1; s=${EPOCHREALTIME/./}
    env "$@" SWARM_AMD_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "results written|released|Clustering|uploaded|Hashing" | grep "^\[t" | tr '\n' ' '
    e=${EPOCHREALTIME/./}; echo " wall_ms $(( (e - s) / 1000 ))"
  done
}
{
run default X=1
run keep_gpu SWARM_AMD_KEEP_GPU=1
run no_trim SWARM_AMD_NO_TRIM=1
run neither SWARM_AMD_KEEP_GPU=1 SWARM_AMD_NO_TRIM=1
md5sum /tmp/o.txt
} > $O/runs.txt 2>&1
bash tools/stress/cold_runs.sh 160 > $O/cold.txt 2>&1
cat $O/runs.txt | cut -c1-400; tail -n 2 $O/cold.txt
