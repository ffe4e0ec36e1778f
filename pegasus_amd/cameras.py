"""``scene.cameras.Camera`` / ``MiniCam`` counterparts (the build's own versions of the classes PEGASUS
imports from the missing submodule: /root/reference/src/gs/pegasus_setup.py:10-15,130-140 builds
``Camera(colmap_id, R, T, FoVx, FoVy, image=torch.empty((3,H,W)), gt_alpha_mask, image_name, uid)``; callers read
``world_view_transform, projection_matrix, full_proj_transform, camera_center, R, T, image_width/height``:
/root/reference/src/tools/pegasus_manipulation.py:235,248, pegasus_working.py:464-466).
``R`` is the camera-to-world rotation, ``T`` the world-to-camera translation (COLMAP convention)."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from . import graphics as G


class Camera(nn.Module):
    def __init__(self, colmap_id, R, T, FoVx, FoVy, image, gt_alpha_mask, image_name, uid,
                 trans=np.array([0.0, 0.0, 0.0]), scale=1.0, data_device="cuda", image_width=None, image_height=None):
        """``image=None`` with ``image_width`` / ``image_height``: a render-only camera (no ground-truth image is kept:
        the reference's own cameras are built around ``torch.empty((3, H, W))`` placeholders, pegasus_setup.py:130-140)."""
        super().__init__()
        self.uid = uid
        self.colmap_id = colmap_id
        self.R = R
        self.T = T
        self.FoVx = FoVx
        self.FoVy = FoVy
        self.image_name = image_name
        try:
            self.data_device = torch.device(data_device)
        except Exception as e:   # same behaviour as upstream: fall back to the default device name
            print(e)
            print(f"[Warning] Custom device {data_device} failed, fallback to default cuda device")
            self.data_device = torch.device("cuda")
        if image is None:
            if image_width is None or image_height is None:
                raise ValueError("Camera needs an image or image_width / image_height")
            self.original_image = None
            self.image_width, self.image_height = int(image_width), int(image_height)
        else:
            self.original_image = image.clamp(0.0, 1.0).to(self.data_device)
            self.image_width = self.original_image.shape[2]
            self.image_height = self.original_image.shape[1]
            if gt_alpha_mask is not None:
                self.original_image *= gt_alpha_mask.to(self.data_device)
        self.zfar = G.ZFAR
        self.znear = G.ZNEAR
        self.trans = trans
        self.scale = scale
        dev = self.data_device
        # Same values as upstream's Camera; the transposes are made CONTIGUOUS here, once, on the host: upstream keeps
        # `.transpose(0, 1)` views, whose inverse is column-major too, so every render() of a camera paid two device copy
        # kernels (view matrix, camera centre) to hand the rasterizer contiguous memory.
        self.world_view_transform = torch.tensor(G.getWorld2View2(R, T, trans, scale)).transpose(0, 1).contiguous().to(dev)
        self.projection_matrix = torch.tensor(
            G.getProjectionMatrix(self.znear, self.zfar, self.FoVx, self.FoVy)).transpose(0, 1).contiguous().to(dev)
        self.full_proj_transform = (self.world_view_transform.unsqueeze(0).bmm(
            self.projection_matrix.unsqueeze(0))).squeeze(0)
        self.camera_center = self.world_view_transform.inverse()[3, :3].contiguous()


class MiniCam:
    def __init__(self, width, height, fovy, fovx, znear, zfar, world_view_transform, full_proj_transform):
        self.image_width = width
        self.image_height = height
        self.FoVy = fovy
        self.FoVx = fovx
        self.znear = znear
        self.zfar = zfar
        self.world_view_transform = world_view_transform
        self.full_proj_transform = full_proj_transform
        view_inv = torch.inverse(self.world_view_transform)
        self.camera_center = view_inv[3][:3]
