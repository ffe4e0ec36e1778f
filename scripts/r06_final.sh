#!/bin/bash
# Round-6 evidence run (GPU box): tests, counter passes + issue model, bench lines, kernel traces -> gpurun_out/r06_*
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set}"
mkdir -p gpurun_out
P=gpurun_out/r06
python -m pytest tests -m gpu -q 2>&1 | tail -5 > ${P}_pytest_gpu.txt
# counters first: bench.py reads profiles/pmc*.json + profiles/issue_model.json, so the lines below carry THIS code's numbers
bash scripts/pmc_profile.sh r06_pmc > ${P}_pmc.log 2>&1
bash scripts/pmc_profile.sh r06_pmc_c5 --workload c5 --views 200 > ${P}_pmc_c5.log 2>&1
cp gpurun_out/r06_pmc.json profiles/pmc.json; cp gpurun_out/r06_pmc_c5.json profiles/pmc_c5.json
cp gpurun_out/r06_pmc.json profiles/r06_pmc.json; cp gpurun_out/r06_pmc_c5.json profiles/r06_pmc_c5.json
python scripts/issue_model.py profiles/r06_pmc.json profiles/r06_pmc_c5.json > ${P}_issue_model.log 2>&1
cp profiles/r06_issue_model.txt profiles/issue_model.json gpurun_out/
python bench.py > ${P}_bench_default.json 2> ${P}_bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > ${P}_bench_steps20_warmup5.json 2>/dev/null
python bench.py --data-points all --no-cpu-baseline --no-drop-in > ${P}_bench_all_data_points.json 2>/dev/null
python bench.py --workload c5 --dynamic --views 200 --no-drop-in > ${P}_bench_c5_dynamic.json 2>/dev/null
python bench.py --workload c5 --views 200 --no-drop-in --no-cpu-baseline > ${P}_bench_c5_static.json 2>/dev/null
python bench.py --dynamic --no-drop-in --no-cpu-baseline > ${P}_bench_c3_dynamic.json 2>/dev/null
python bench.py --workload c2 --views 64 --no-drop-in --no-cpu-baseline > ${P}_bench_c2.json 2>/dev/null
python bench.py --views 4096 --steps 128 --repeats 3 --no-drop-in --no-cpu-baseline > ${P}_bench_c4_4096views_1gpu.json 2>/dev/null
python bench.py --workload c5 --dynamic --views 200 --batch 40 --steps 5 --no-drop-in --no-cpu-baseline > ${P}_bench_c5_dynamic_200steps_1gpu.json 2>/dev/null
python bench.py --facade > ${P}_bench_facade.json 2>/dev/null
python bench.py --width 640 --height 480 --objects 6 --data-points all --no-cpu-baseline > ${P}_bench_ref_default_640x480.json 2>/dev/null
python bench.py --gpus 2 --share-devices --backend gloo --steps 6 --warmup 2 --no-drop-in > ${P}_bench_rehearsal_2ranks_1gpu.json 2>/dev/null
python bench.py --force-dist --backend nccl --no-drop-in --no-cpu-baseline > ${P}_bench_rccl_1rank_forced.json 2>/dev/null
python bench.py --force-dist --backend nccl --full-outputs --no-drop-in --no-cpu-baseline > ${P}_bench_rccl_1rank_forced_full_outputs.json 2>/dev/null
bash scripts/trace_run.sh r06 --no-drop-in > /dev/null 2>&1
bash scripts/trace_run.sh r06_sync --no-drop-in --sync-steps > /dev/null 2>&1
bash scripts/trace_run.sh r06_c5_sync --no-drop-in --sync-steps --workload c5 --views 200 > /dev/null 2>&1
bash scripts/single_view_trace.sh r06 40 c3 > /dev/null 2>&1
# records-only frame sets against full ones: frames/s interleaved, then the compositor's WRITE_SIZE per launch of each form
( python scripts/records_only_ab.py both 3 2>&1 | grep -v amdgpu.ids
  for kind in full records; do
    rm -rf /tmp/ro_$kind
    ( cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/ro_$kind -- python3 $GRAFT_REPO_ROOT/scripts/records_only_ab.py $kind 1 > /tmp/ro_$kind.log 2>&1 )
    python - $kind "$(find /tmp/ro_$kind -name '*.db' | head -1)" <<'PY'
import sys
sys.path.insert(0, "scripts")
import pmc_report
acc, disp = pmc_report.read(sys.argv[2])
for k in sorted(acc):
    if "WRITE_SIZE" in acc[k]:
        print(f"  {sys.argv[1]:8s} {k:44s} WRITE_SIZE {acc[k]['WRITE_SIZE'] / len(disp[k]) * 1024 / 1e6:9.3f} MB per launch ({len(disp[k])} launches of 32 views)")
PY
  done ) > ${P}_records_only.txt 2>&1
python scripts/silhouette_time.py 2>&1 | grep -v amdgpu.ids > ${P}_silhouette_time.txt
( python scripts/fuzz_parity.py 90000 1500 2>&1 | tail -1; python scripts/fuzz_fused.py 5000 200 2>&1 | tail -1; python scripts/fuzz_layered.py 3000 100 2>&1 | tail -1
  python scripts/soak_determinism.py 4 c3 2>&1 | tail -1; python scripts/full_size_parity.py 2>&1 | tail -1 ) | grep -v amdgpu.ids > ${P}_verification.txt
for f in ${P}_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1], d.get("value"), d.get("value_min"), d.get("value_max"), r.get("stage_ms_per_view"), r.get("bound"), r.get("frac"), r.get("hbm_frac"),
          (r.get("issue_model") or {}).get("clock_mhz"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
cat ${P}_pytest_gpu.txt ${P}_records_only.txt ${P}_silhouette_time.txt ${P}_verification.txt
