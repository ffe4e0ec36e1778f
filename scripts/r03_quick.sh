cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do python bench.py --no-cpu-baseline --no-drop-in 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print(d['value'], r['stage_ms_per_view'], r['raster_only_views_per_s'])"; done
python -m pytest tests -m gpu -x -q -k "fused or frame_renderer or semantic" 2>&1 | tail -3
