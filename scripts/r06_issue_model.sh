#!/bin/bash
# round 6: everything the issue model needs, on one box -- (1) per-kind issue costs + shader clock (valu_classes), (2) which SQ
# class counter every kind is counted under (the same binary under rocprofv3 --pmc), (3) the class counters of the bench
# command (pmc_profile.sh: C3 and C5), (4) list statistics
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
mkdir -p gpurun_out
B=$PWD/scripts/microbench/valu_classes
[ -x $B ] || ( cd scripts/microbench && hipcc --offload-arch=gfx950 -O3 valu_classes.hip -o valu_classes )
{ $B 8; $B 4; } > gpurun_out/r06_valu_classes.txt 2>&1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=/tmp/vc_$i; rm -rf $d
  ( cd /tmp && timeout 300 rocprofv3 --pmc $grp -d $d -- $B 8 > /tmp/vc_$i.log 2>&1 )
  db=$(find $d -name "*.db" | head -1)
  python3 - "$db" $i <<'PY' >> gpurun_out/r06_valu_classes_pmc.txt
import sqlite3, sys, collections
db, i = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db); cur = con.cursor()
cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
ix = {c: k for k, c in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ix else [c for c in cols if "kernel" in c and "name" in c][0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in cur.execute("select * from counters_collection"):
    k = r[ix[name_col]].split("(")[0]
    acc[k][r[ix["counter_name"]]] += float(r[ix["value"]]); disp[k].add(r[ix["dispatch_id"]])
print(f"# pass {i}: counter sums per dispatch (each kind launches twice: 2048 workgroups x 4 waves x 65536 instructions = 5.37e8 wave-instructions per dispatch)")
for k in acc:
    n = len(disp[k])
    print(k, " ".join(f"{c}={v / n:.4g}" for c, v in sorted(acc[k].items())))
PY
done
bash scripts/pmc_profile.sh r06_pmc > /dev/null 2>&1
bash scripts/pmc_profile.sh r06_pmc_c5 --workload c5 --views 200 > /dev/null 2>&1
python scripts/list_stats.py c3 > gpurun_out/r06_list_lengths.txt 2>/dev/null
python scripts/list_stats.py c5 >> gpurun_out/r06_list_lengths.txt 2>/dev/null
