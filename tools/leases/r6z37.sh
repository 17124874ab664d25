for ao in 0 1 0 1; do
  SWA_DN_ALIGN_ORDER=$ao timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 > /dev/null 2>&1
  python -c "
import json
d=json.load(open('bench_detail.json'))['config']['configs3']; print('order=$ao', d['clustering_seconds'], d['gpu_kernels_ms'])"
done
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 > /dev/null 2>&1
python -c "
import json
d=json.load(open('bench_detail.json'))['config']['configs3']; print('default', d['clustering_seconds'], d['gpu_kernels_ms'])"
