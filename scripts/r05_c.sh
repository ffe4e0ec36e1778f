#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
P=gpurun_out/r05c
timeout 1200 python -m pytest tests -m gpu -q -k "grazing or count_walk_verdicts or silhouette_masks or facade or semantic_wrappers or render_facade" 2>&1 | tail -150 > ${P}_pytest_sel.txt
bash scripts/single_view_trace.sh r05c 40 c3 > /dev/null 2>&1
python bench.py --facade > ${P}_bench_facade.json 2> ${P}_bench_facade.err
cat ${P}_pytest_sel.txt | tail -120
cat gpurun_out/r05c_single_view_timeline.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05c_bench_facade.json').read().strip().splitlines()[-1])
di=d['drop_in']
print({k:di[k] for k in ('frames_per_s','ms_per_frame','frames_per_s_all_data_points','render_call_ms','ms_per_part')})
print(di['dynamic'])
PY
