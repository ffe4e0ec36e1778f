#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
P=gpurun_out/r05d
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -60 > ${P}_pytest_gpu.txt
./scripts/microbench/valu_rates > ${P}_valu_rates.txt 2>&1
bash scripts/single_view_trace.sh r05d 40 c3 > /dev/null 2>&1
python bench.py --facade > ${P}_bench_facade.json 2> ${P}_bench_facade.err
python bench.py --width 640 --height 480 --objects 6 --data-points all --no-cpu-baseline > ${P}_bench_ref_default_640x480.json 2> ${P}_bench_ref.err
tail -15 ${P}_pytest_gpu.txt
cat ${P}_valu_rates.txt
cat gpurun_out/r05d_single_view_timeline.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05d_bench_facade.json').read().strip().splitlines()[-1])
di=d['drop_in']
print({k:di[k] for k in ('frames_per_s','ms_per_frame','frames_per_s_all_data_points','render_call_ms','ms_per_part')})
print(di['dynamic'])
d=json.loads(open('gpurun_out/r05d_bench_ref_default_640x480.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['workload'], d['roofline']['stage_ms_per_view'], (d.get('drop_in') or {}).get('frames_per_s'), ((d.get('drop_in') or {}).get('dynamic') or {}).get('frames_per_s'))
PY
