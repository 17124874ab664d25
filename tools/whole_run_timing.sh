python - <<'PY'
import bench, time
fa = bench.gen_fasta(10_000_000, 150, 1)
print(fa)
PY
FA=/tmp/swa_bench_10000000x150_s1.fa
ls -la $FA
for i in 1 2 3; do
  SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 bash -c "time ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA" 2>&1 | tail -60
  echo ------
done
if [ -x oracle/_ref/swarm ]; then
  for t in 8 16; do
    bash -c "time oracle/_ref/swarm -d 1 -t $t -o /tmp/ro.txt -l /dev/null $FA" 2>&1 | grep real
  done
  cmp /tmp/o.txt /tmp/ro.txt && echo "outputs identical"
fi
