set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03_a_pytest.txt
python bench.py > gpurun_out/r03_a_bench_default.json 2> gpurun_out/r03_a_bench_default.err
python bench.py --gpus 2 --share-devices --backend gloo --steps 6 --warmup 2 > gpurun_out/r03_a_bench_rehearsal_2ranks_1gpu.json 2> gpurun_out/r03_a_bench_rehearsal.err
python bench.py --workload c5 --dynamic --views 200 --no-drop-in > gpurun_out/r03_a_bench_c5_dynamic.json 2> gpurun_out/r03_a_bench_c5_dynamic.err
python bench.py --facade > gpurun_out/r03_a_bench_facade.json 2> gpurun_out/r03_a_bench_facade.err
tail -3 gpurun_out/r03_a_*.err
cat gpurun_out/r03_a_pytest.txt
