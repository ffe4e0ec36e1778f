from typing import NamedTuple

import numpy as np
import torch

from pegasus_amd.graphics import focal2fov, fov2focal, getWorld2View, getWorld2View2  # noqa: F401
from pegasus_amd.graphics import getProjectionMatrix as _proj


class BasicPointCloud(NamedTuple):
    points: np.array
    colors: np.array
    normals: np.array


def getProjectionMatrix(znear, zfar, fovX, fovY):
    return torch.from_numpy(_proj(znear, zfar, fovX, fovY))


def geom_transform_points(points, transf_matrix):
    P, _ = points.shape
    ones = torch.ones(P, 1, dtype=points.dtype, device=points.device)
    points_out = torch.matmul(torch.cat([points, ones], dim=1), transf_matrix.unsqueeze(0))
    denom = points_out[..., 3:] + 0.0000001
    return (points_out[..., :3] / denom).squeeze(dim=0)
