#!/bin/bash
# rocprofv3 --kernel-trace --stats of any python script of the repo -> gpurun_out/<name>_kernel_stats.txt
#   scripts/trace_script.sh <name> <script.py> [args...]
cd "$(dirname "$0")/.."
REPO=$PWD
name=$1; shift
script=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
d=/tmp/trace_$name
rm -rf $d
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $d -- python3 $REPO/$script "$@" > /tmp/trace_$name.log 2>&1; tail -5 /tmp/trace_$name.log )
db=$(find $d -name "*.db" | head -1)
python3 scripts/rocprof_summary.py $db gpurun_out/${name}_kernel_stats.txt "python3 $script $*" > /dev/null
head -30 gpurun_out/${name}_kernel_stats.txt
