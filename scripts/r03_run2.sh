cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03_b_pytest.txt
bash scripts/single_view_trace.sh r03_b 40 c3 > /dev/null 2>&1
bash scripts/pmc_profile.sh r03_b_pmc > gpurun_out/r03_b_pmc.log 2>&1
python bench.py --facade > gpurun_out/r03_b_bench_facade.json 2> gpurun_out/r03_b_bench_facade.err
cat gpurun_out/r03_b_pytest.txt
