"""SH helpers (the build's versions of ``utils.sh_utils`` that PEGASUS imports from the missing submodule:
/root/reference/pegasus.py:231, /root/reference/src/gs/render.py:8,51, /root/reference/src/gs/gaussian_model.py:28)."""
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5


def sh_basis(deg: int, dirs):
    """Real SH basis values [..., (deg+1)^2] for unit directions [..., 3] (numpy or torch), in the sign and
    ordering convention of the rasterizer (oracle/pgr_oracle.c sh_basis)."""
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    one = x * 0 + 1
    b = [C0 * one]
    if deg > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
        if deg > 2:
            b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
                  C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
                  C3[6] * x * (xx - 3 * yy)]
    if hasattr(x, "numpy") or type(x).__module__.startswith("torch"):
        import torch
        return torch.stack(b, dim=-1)
    import numpy as np
    return np.stack(b, axis=-1)


def eval_sh(deg: int, sh, dirs):
    """sh [..., C, (max_deg+1)^2], dirs [..., 3] -> [..., C]  (torch; the convert_SHs_python path of render())."""
    b = sh_basis(deg, dirs)                       # [..., nc]
    nc = (deg + 1) ** 2
    return (sh[..., :nc] * b[..., None, :]).sum(-1)
