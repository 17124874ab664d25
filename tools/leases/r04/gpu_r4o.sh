#!/bin/bash
# round 4, lease o: SQ / LDS counters of k_group1 (what it waits for)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'$R'); import bench; bench.gen_fasta(10000000,150,1)"
rocprofv3 --list-avail > $O/avail.txt 2>&1
want_sets=("SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_LDS"
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
           "SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU")
i=0
for set in "${want_sets[@]}"; do
  i=$((i+1)); have=""
  for c in $set; do grep -qw "$c" $O/avail.txt && have="$have $c"; done
  echo "pass $i:$have" >> $O/passes.txt
  [ -z "$have" ] && continue
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $have -d $O/pmc$i -o g -- python $R/tools/experiments/time_build.py --steps 3 > $O/pmc$i.log 2>&1
  find $O/pmc$i -name '*counter_collection.csv' -exec cp {} $O/pmc$i.csv \;
  rm -rf $O/pmc$i
done
python - <<PY > $O/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("$O/pmc*.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_group1" not in k and "k_keys" not in k: continue
        acc[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k[:40], r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(f, k)
        for c, v in d.items(): print("   ", c, v / n[(k, c)])
PY
cat $O/summary.txt
