#!/bin/bash
# Copies what scripts/r06_final.sh left under gpurun_out/ into profiles/ (the judged, committed copies) and checks that the
# counter files carry the hash of the kernel sources in this tree.
set -e
cd "$(dirname "$0")/.."
for f in r06_bench_all_data_points.json r06_bench_c2.json r06_bench_c3_dynamic.json r06_bench_c4_4096views_1gpu.json \
         r06_bench_c5_dynamic.json r06_bench_c5_dynamic_200steps_1gpu.json r06_bench_c5_static.json r06_bench_default.json \
         r06_bench_facade.json r06_bench_rccl_1rank_forced.json r06_bench_rccl_1rank_forced_full_outputs.json \
         r06_bench_ref_default_640x480.json r06_bench_rehearsal_2ranks_1gpu.json r06_bench_steps20_warmup5.json \
         r06_c5_sync_kernel_stats.txt r06_sync_kernel_stats.txt r06_kernel_stats.txt r06_issue_model.txt r06_pmc.json r06_pmc.txt \
         r06_pmc_c5.json r06_pmc_c5.txt r06_pytest_gpu.txt r06_records_only.txt r06_silhouette_time.txt \
         r06_single_view_timeline.txt r06_verification.txt; do
  cp gpurun_out/$f profiles/$f
done
cp gpurun_out/r06_pmc.json profiles/pmc.json
cp gpurun_out/r06_pmc_c5.json profiles/pmc_c5.json
cp gpurun_out/issue_model.json profiles/issue_model.json
python3 - <<'PY'
import json
from pegasus_amd import build
want = build.source_hash()
got = {f: json.load(open(f"profiles/{f}")).get("library_sha16") for f in ("pmc.json", "pmc_c5.json")}
im = json.load(open("profiles/issue_model.json"))
got.update({f"issue_model.{k}": v for k, v in im["library_sha16"].items()})
print("sources", want, got)
assert all(v == want for v in got.values()), "counter files were taken with other kernel sources"
PY
