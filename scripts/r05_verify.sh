#!/bin/bash
# Long verification run with the final kernels: random scenes vs the oracle, fused / layered equivalences, determinism soak,
# full-size views spread over the camera sets -> gpurun_out/r05_verification_long.txt
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
( python scripts/fuzz_parity.py 100000 8000 2>&1 | tail -3; python scripts/fuzz_fused.py 6000 1000 2>&1 | tail -2; python scripts/fuzz_layered.py 4000 500 2>&1 | tail -2
  python scripts/soak_determinism.py 10 c3 2>&1 | tail -1; python scripts/soak_determinism.py 4 c5 2>&1 | tail -1; python scripts/full_size_parity.py 16 4 2>&1 | tail -22 ) | grep -v amdgpu.ids > gpurun_out/r05_verification_long.txt
cat gpurun_out/r05_verification_long.txt
