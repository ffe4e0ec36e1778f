#!/bin/bash
# rocprofv3 --kernel-trace --stats of a serial (--sync-steps) bench run -> gpurun_out/<name>_kernel_stats.txt
cd "$(dirname "$0")/.."
REPO=$PWD
name=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
d=/tmp/trace_$name
rm -rf $d
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $d -- python3 $REPO/bench.py --no-cpu-baseline "$@" > gpurun_out_$name.log 2>&1; tail -1 /tmp/gpurun_out_$name.log > $REPO/gpurun_out/${name}_bench_under_rocprof.json )
db=$(find $d -name "*.db" | head -1)
python3 scripts/rocprof_summary.py $db gpurun_out/${name}_kernel_stats.txt "python3 bench.py --no-cpu-baseline $*"
head -24 gpurun_out/${name}_kernel_stats.txt
