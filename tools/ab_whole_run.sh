python - <<'PY'
import bench
print(bench.gen_fasta(10_000_000, 150, 1))
PY
FA=/tmp/swa_bench_10000000x150_s1.fa
for rep in 1 2 3 4 5; do
  SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 bash -c "time ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA" 2>&1 | grep -E "sort|gather|waited|context created|written|real" | tr '\n' ' '; echo
done
