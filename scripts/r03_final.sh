#!/bin/bash
# Round-3 evidence run (GPU box): tests, bench lines, kernel traces, counter passes -> gpurun_out/r03_*
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=gpurun_out/r03
python -m pytest tests -m gpu -q 2>&1 | tail -5 > ${P}_pytest_gpu.txt
# counters first: bench.py reads profiles/pmc.json, so the lines below carry THIS code's traffic / issue numbers
bash scripts/pmc_profile.sh r03_pmc > ${P}_pmc.log 2>&1
cp gpurun_out/r03_pmc.json profiles/pmc.json
python bench.py > ${P}_bench_default.json 2> ${P}_bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > ${P}_bench_steps20_warmup5.json 2>/dev/null
python bench.py --workload c5 --dynamic --views 200 --no-drop-in > ${P}_bench_c5_dynamic.json 2>/dev/null
python bench.py --workload c5 --views 200 --no-drop-in --no-cpu-baseline > ${P}_bench_c5_static.json 2>/dev/null
python bench.py --dynamic --no-drop-in --no-cpu-baseline > ${P}_bench_c3_dynamic.json 2>/dev/null
python bench.py --workload c2 --views 64 --no-drop-in --no-cpu-baseline > ${P}_bench_c2.json 2>/dev/null
python bench.py --facade > ${P}_bench_facade.json 2>/dev/null
python bench.py --gpus 2 --share-devices --backend gloo --steps 6 --warmup 2 --no-drop-in > ${P}_bench_rehearsal_2ranks_1gpu.json 2>/dev/null
bash scripts/trace_run.sh r03 --no-drop-in > /dev/null 2>&1
bash scripts/trace_run.sh r03_sync --no-drop-in --sync-steps > /dev/null 2>&1
bash scripts/single_view_trace.sh r03 40 c3 > /dev/null 2>&1
for f in ${P}_bench_*.json; do echo "== $f"; tail -c 600 $f | head -c 300; echo; done
cat ${P}_pytest_gpu.txt
