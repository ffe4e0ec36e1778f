cd $GRAFT_REPO_ROOT
python scripts/single_view_calls.py 64 c3 2>&1 | tail -1
python scripts/single_view_calls.py 64 c3 2>&1 | tail -1
PGR_BLOCK_CULL=0 python scripts/single_view_calls.py 64 c3 2>&1 | tail -1
PGR_BLOCK_CULL=0 python scripts/single_view_calls.py 64 c3 2>&1 | tail -1
python -m pytest tests -m gpu -x -q -k "facade or backward" 2>&1 | tail -2
python -c "
import cProfile, pstats, sys, io
sys.argv=['x','48','c3']
import runpy
pr=cProfile.Profile(); pr.enable()
runpy.run_path('scripts/single_view_calls.py', run_name='__main__')
pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:6000])
" 2>&1 | grep -v "amdgpu.ids" | tail -45
