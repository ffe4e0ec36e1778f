#!/bin/bash
# rocprofv3 --kernel-trace of N single-view render() calls -> gpurun_out/<name>_single_view_timeline.txt
cd "$(dirname "$0")/.."
REPO=$PWD
name=${1:-r03}; n=${2:-40}; wl=${3:-c3}; mode=${4:-}
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 scripts/single_view_calls.py $n $wl $mode > gpurun_out/${name}_single_view_timeline.txt 2>&1
d=/tmp/trace_sv_$name
rm -rf $d
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $d -- python3 $REPO/scripts/single_view_calls.py $n $wl $mode > /tmp/sv_$name.log 2>&1 )
db=$(find $d -name "*.db" | head -1)
{ echo "# under rocprofv3 --kernel-trace:"; tail -1 /tmp/sv_$name.log; python3 scripts/single_view_timeline.py $db; } >> gpurun_out/${name}_single_view_timeline.txt
cat gpurun_out/${name}_single_view_timeline.txt
