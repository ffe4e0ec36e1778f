# A/B of the count walk's records (PGR_BIN_RECORDS=0: the scatter walk re-evaluates every candidate)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for r in 0 1; do
  echo "PGR_BIN_RECORDS=$r"
  PGR_BIN_RECORDS=$r python bench.py --no-drop-in --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['roofline']['stage_ms_per_view'])"
done
done
