#!/bin/bash
# lease r6z21: the N > 1 flow of bench.py on the evening build: two and four ranks on one GPU over gloo at full size
# (20 M / 40 M amplicons replicated; the gathered CSR must equal the whole network: the script checks), and the GPU-side sharding tests
O=$PWD/gpurun_out/r6z21_out; mkdir -p $O
for n in 2 4; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 3 --warmup 1 --dev-backend gloo > $O/gpus$n.json 2> $O/gpus$n.err
  echo "gpus=$n rc=$?"; tail -c 700 $O/gpus$n.json; echo; tail -3 $O/gpus$n.err | cut -c1-300
done
(timeout 900 python -m pytest tests/test_bench_sharded_gpu.py -m gpu -x -q > $O/sharded_tests.txt 2>&1; tail -3 $O/sharded_tests.txt)
