#!/bin/bash
# tools/closing_artefacts.sh [tag] — the artefacts a round closes with (on the GPU box: gpurun -- bash tools/closing_artefacts.sh):
# the default bench line as the driver runs it, rocprofv3 kernel stats of the step at 10 M and 1 M, the PMC passes, smoke(),
# the whole GPU suite.  Everything lands under gpurun_out/<tag>/ (default: closing); copy what is kept into profiles/rNN/.
TAG=${1:-closing}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/status.txt
cp $R/bench_detail.json $O/bench_detail.json 2>/dev/null     # (the side file of THIS run: the later --no-extras runs write their own)
tail -c 600 $O/bench_default.json; echo
KSTATS_LINES=40 timeout 600 bash tools/kstats.sh ${TAG}_step10M python $R/bench.py --steps 20 --warmup 3 --no-extras > $O/kstats.txt 2>&1; cp $R/gpurun_out/${TAG}_step10M_kernel_stats.csv $O/ 2>/dev/null
head -14 $O/kstats.txt | cut -c1-150
KSTATS_LINES=40 timeout 600 bash tools/kstats.sh ${TAG}_step1M python $R/bench.py --per-gpu 1000000 --steps 20 --warmup 3 --no-extras > $O/kstats1M.txt 2>&1; cp $R/gpurun_out/${TAG}_step1M_kernel_stats.csv $O/ 2>/dev/null
timeout 1200 bash tools/profile_d1.sh ${TAG}_pmc 3 > $O/profile.log 2>&1; echo "profile rc=$?" | tee -a $O/status.txt
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
timeout 1800 python -m pytest tests -q -m gpu > $O/tests_all.log 2>&1; echo "tests_all rc=$?" | tee -a $O/status.txt
tail -4 $O/tests_all.log
