#!/bin/bash
# Round-5 first GPU run: the suite with the new tests, the VALU-rate microbenchmark, the headline on the stated camera set
# and on the rounds-1-4 subset beside it (A/B on one box), the drop-in loop static + dynamic, a single-view trace.
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set}"
mkdir -p gpurun_out
P=gpurun_out/r05a
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > ${P}_pytest_gpu.txt
./scripts/microbench/valu_rates > ${P}_valu_rates.txt 2>&1
python bench.py > ${P}_bench_default.json 2> ${P}_bench_default.err
python bench.py --camera-set fibonacci_above_9deg --no-cpu-baseline --no-drop-in > ${P}_bench_above9deg.json 2>/dev/null
python bench.py --no-cpu-baseline --no-drop-in > ${P}_bench_default_2.json 2>/dev/null
python bench.py --camera-set fibonacci_above_9deg --no-cpu-baseline --no-drop-in > ${P}_bench_above9deg_2.json 2>/dev/null
python bench.py --workload c5 --views 200 --no-drop-in --no-cpu-baseline > ${P}_bench_c5_static.json 2>/dev/null
python bench.py --workload c5 --views 200 --camera-set fibonacci_above_9deg --no-drop-in --no-cpu-baseline > ${P}_bench_c5_static_above9deg.json 2>/dev/null
python bench.py --facade > ${P}_bench_facade.json 2> ${P}_bench_facade.err
bash scripts/single_view_trace.sh r05a 40 c3 > /dev/null 2>&1
for f in ${P}_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1], d.get("value"), d.get("value_min"), d.get("value_max"), r.get("stage_ms_per_view"), r.get("bound"), r.get("frac"), r.get("hbm_frac"), (d.get("drop_in") or {}).get("frames_per_s"), ((d.get("drop_in") or {}).get("dynamic") or {}).get("frames_per_s"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
cat ${P}_pytest_gpu.txt
tail -30 ${P}_valu_rates.txt
