cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "long or c5 or tie or hostile or c2 or c1 or fuzz or full_size" 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --no-drop-in 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('c3', d['value'], r['stage_ms_per_view'])"; done
python bench.py --workload c5 --views 200 --no-cpu-baseline --no-drop-in 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('c5', d['value'], r['stage_ms_per_view'])"
bash scripts/r03_sorttiming.sh
