#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel of libpegasus_raster (hipcc -Rpass-analysis=kernel-resource-usage).
    python scripts/kernel_resources.py [extra hipcc flags]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pegasus_amd.build import FLAGS  # noqa: E402

cmd = ["hipcc", *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/pgr_resources.so",
       str(ROOT / "pegasus_amd/csrc/pegasus_raster.hip"), *sys.argv[1:]]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip(" []")
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = re.sub(r"\(.*", "", d).replace("void pgr::", "")

    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", b)
        return m.group(1) if m else "?"
    print(f"{d[:84]:84s} vgpr {g('VGPRs'):>3} agpr {g('AGPRs'):>3} spill {g('VGPR Spill'):>3} scratch {g('ScratchSize [bytes/lane]'):>4} "
          f"occ {g('Occupancy [waves/SIMD]'):>2} lds {g('LDS Size [bytes/block]'):>6}")
