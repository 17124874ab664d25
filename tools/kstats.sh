#!/bin/bash
# tools/kstats.sh <tag> <cmd...> — rocprofv3 --kernel-trace --stats of a command; CSV -> gpurun_out/<tag>_kernel_stats.csv
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o k -- "$@" > $R/gpurun_out/${TAG}_run.log 2>&1
find $R/gpurun_out/$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf $R/gpurun_out/$TAG
cut -c1-120 $R/gpurun_out/${TAG}_kernel_stats.csv | head -${KSTATS_LINES:-8}
