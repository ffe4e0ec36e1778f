"""SH <-> RGB helpers (the build's versions of ``utils.sh_utils.RGB2SH / SH2RGB`` that PEGASUS imports
from the missing submodule: /root/reference/pegasus.py:231, /root/reference/src/gs/render.py:8,51)."""
C0 = 0.28209479177387814


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
