cd $GRAFT_REPO_ROOT
python bench.py --views 4096 --steps 128 --no-drop-in --no-cpu-baseline > gpurun_out/r03_bench_c4_4096views_1gpu.json 2>/dev/null
python bench.py --workload c5 --dynamic --views 200 --batch 40 --steps 5 --no-drop-in --no-cpu-baseline > gpurun_out/r03_bench_c5_dynamic_200steps_1gpu.json 2>/dev/null
python scripts/fuzz_parity.py 60000 3000 2>&1 | tail -1
python scripts/fuzz_fused.py 2000 400 2>&1 | tail -1
python scripts/soak_determinism.py 2>&1 | tail -4
bash scripts/single_view_trace.sh r03 40 c3 > /dev/null 2>&1
tail -5 gpurun_out/r03_single_view_timeline.txt
for f in gpurun_out/r03_bench_c4_4096views_1gpu.json gpurun_out/r03_bench_c5_dynamic_200steps_1gpu.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['steps'], d['config'])"; done
