from pegasus_amd.sh_utils import RGB2SH, SH2RGB, eval_sh, C0, C1, C2, C3  # noqa: F401
