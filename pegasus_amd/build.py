"""Builds pegasus_amd/csrc/libpegasus_raster.so for gfx950 with hipcc (in-tree; the .so is git-ignored
but travels to the GPU box with the working-tree snapshot)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libpegasus_raster.so"
ARCH = "gfx950"
# -ffp-contract=off: every fused multiply-add in the kernels is an explicit fmaf(), which is what
# makes the integer stages bit-exact against the oracle (DESIGN.md "Arithmetic contract").
# -munsafe-fp-atomics: the backward pass accumulates with hardware global_atomic_add_f32 instead of a CAS loop.
# -fno-slp-vectorize: the SLP vectoriser pairs scalar fp32 operations into v_pk_* and pays for it in v_mov to line the
# operands up (61 moves in the 614 VALU instructions of the preprocess' per-view body): preprocess -3 %, binning -1.5 %.
# The compositor's packed arithmetic is written with vector types and does not depend on it.
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-munsafe-fp-atomics", "-fPIC", "-shared",
         f"--offload-arch={ARCH}"]


def sources():
    return sorted(CSRC.glob("*.hip")), sorted(CSRC.glob("*.h")) + [CSRC.parents[1] / "include" / "pegasus_raster.h"]


def source_hash() -> str:
    """16 hex digits over the kernel sources, the C header and the compiler flags: WHICH kernels a library holds.  Compiled into
    the library (pgr_version() ends with it) and stamped into the counter files of scripts/pmc_report.py, so that bench.py prices
    a kernel's live duration only with counters of the same kernels.  (The .so itself is not byte-reproducible.)"""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    srcs, hdrs = sources()
    for f in sorted(srcs + hdrs):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not LIB.exists():
        return True
    srcs, hdrs = sources()
    newest = max(p.stat().st_mtime for p in srcs + hdrs + [Path(__file__)])      # (the flags live in this file)
    return LIB.stat().st_mtime < newest


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs, _ = sources()
    tmp = LIB.with_suffix(f".tmp{os.getpid()}.so")
    cmd = [hipcc, *FLAGS, f'-DPGR_SOURCE_HASH="{source_hash()}"', "-o", str(tmp), *map(str, srcs)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
