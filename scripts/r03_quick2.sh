cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "fused or frame_renderer or semantic or posed or block" 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline --no-drop-in 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print(d['value'], r['stage_ms_per_view'])"; done
bash scripts/pmc_profile.sh r03_e_pmc > /dev/null 2>&1
head -8 gpurun_out/r03_e_pmc.txt
