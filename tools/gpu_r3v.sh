#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3v
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for group in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $group -d $O/p$i -o p -- python $R/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  echo "pass $i: $f"
  python - "$f" <<'PY'
import csv, sys, re, collections
if not sys.argv[1]:
    sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_\w+(<[^>(]*>)?)", r["Kernel_Name"]); name = m.group(1) if m else r["Kernel_Name"][:40]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(name, r["Counter_Name"])] += 1
for name in ("k_d1_group_pairs<0, 5>", "k_d1_group_pairs<1, 5>", "k_group1", "k_part_scatter<1, 4096u, 1024u>", "k_part_scatter<0, 4096u, 512u>", "k_csr_bucket<8>", "k_keys<5>"):
    if name in acc:
        n = max(calls[(name, c)] for c in acc[name])
        print(name, {c: round(v / n) for c, v in acc[name].items()})
PY
  rm -rf $O/p$i/*/*.db 2>/dev/null
done
