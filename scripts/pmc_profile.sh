#!/bin/bash
# rocprofv3 counter passes over THE bench command (fused frames path, 32-view batches, C3) -- one counter group per pass,
# no tracing flags -- then ONE report: gpurun_out/<name>.json (what bench.py reads as profiles/pmc.json) and
# gpurun_out/<name>.txt (the human-readable table of the same numbers), so the two cannot diverge.
#   scripts/pmc_profile.sh <out-name> [extra bench flags...]
cd "$(dirname "$0")/.."
REPO=$PWD
name=${1:-pmc}; shift
# the counter passes describe the SINGLE-PROCESS path: a flag that turns the profiled bench.py into a launcher of a process
# tree (--gpus N), or into another measurement (--facade, --stub-renderer), would make pmc.json describe something else
for a in "$@"; do
  case "$a" in
    --gpus*|--facade|--stub-renderer|--share-devices|--force-dist|--backend*)
      echo "pmc_profile.sh: flag $a is not allowed here (single-process counter passes only)" >&2; exit 2;;
  esac
done
export TMPDIR=/tmp
mkdir -p gpurun_out
# (all 16 batches of the 512 cameras: the set is ordered by elevation, three batches would be the grazing views only)
BENCH_FLAGS="--steps 16 --warmup 1 --repeats 1 --no-cpu-baseline --no-drop-in --profile-steps 1 --sync-steps $*"
# (round 6: three more passes with the SQ instruction-CLASS counters -- what the issue model of bench.py's roofline prices with
#  the per-class costs of scripts/microbench/valu_classes.hip: profiles/r06_issue_model.txt)
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
         "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"
         "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU"
         "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT")
dbs=()
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  d=/tmp/pmc_${name}_$i
  rm -rf $d
  ( cd /tmp && timeout 600 rocprofv3 --pmc $grp -d $d -- python3 $REPO/bench.py $BENCH_FLAGS > /tmp/pmc_${name}_$i.log 2>&1 )
  db=$(find $d -name "*.db" | head -1)
  if [ -z "$db" ]; then echo "pass $i ($grp): no database; log tail:"; tail -5 /tmp/pmc_${name}_$i.log; fi
  dbs+=("${db:-MISSING}")
done
python3 scripts/pmc_report.py gpurun_out/$name.json gpurun_out/$name.txt "python3 bench.py $BENCH_FLAGS" "${dbs[@]}"
