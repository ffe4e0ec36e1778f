cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --facade > gpurun_out/r03_c_bench_facade.json 2> gpurun_out/r03_c_bench_facade.err
python -m pytest tests -m gpu -x -q -k "facade or generate" 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads([x for x in open('gpurun_out/r03_c_bench_facade.json') if x.startswith('{')][-1])
print(d['value'], d['drop_in']['ms_per_part'], d['drop_in']['render_call_ms'])
PY
