"""``scene.Scene`` counterpart (SURVEY.md section 8b companion names): the trained-model directory layout the upstream
3DGS code writes and PEGASUS's viewer / rotation scripts open through ``Scene(dataset, gaussians, load_iteration=...)``
(/root/reference/src/gs/gs_object_rotation.py:20,79; commented-out twins in gs_viewer.py:54,
object_visualization.py:606):

    <model_path>/point_cloud/iteration_<N>/point_cloud.ply     the Gaussians (GaussianModel.load_ply / save_ply)
    <model_path>/cameras.json                                  one record per training camera: id, img_name, width,
                                                               height, position (camera centre), rotation (camera-to-
                                                               world, row-major 3x3), fx, fy

Only what rendering needs is implemented: no COLMAP / Blender source readers (offline asset pipeline, out of scope),
no ground-truth images -- cameras are render-only (Camera(image=None, image_width=, image_height=))."""
from __future__ import annotations

import json
import os
import random
from pathlib import Path

import numpy as np

from .cameras import Camera
from .graphics import focal2fov


def searchForMaxIteration(folder) -> int:
    """Largest N among the ``iteration_<N>`` entries of ``folder`` (upstream utils.system_utils)."""
    return max(int(name.split("_")[-1]) for name in os.listdir(folder))


def camera_to_JSON(cam_id: int, camera) -> dict:
    """The cameras.json record of a Camera (upstream utils.camera_utils.camera_to_JSON)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(camera.R).transpose()
    Rt[:3, 3] = np.asarray(camera.T)
    Rt[3, 3] = 1.0
    W2C = np.linalg.inv(Rt)
    from .graphics import fov2focal
    return {"id": cam_id, "img_name": camera.image_name, "width": int(camera.image_width), "height": int(camera.image_height),
            "position": W2C[:3, 3].tolist(), "rotation": [row.tolist() for row in W2C[:3, :3]],
            "fy": fov2focal(camera.FoVy, camera.image_height), "fx": fov2focal(camera.FoVx, camera.image_width)}


def cameras_from_json(path, data_device="cuda"):
    cams = []
    for rec in json.loads(Path(path).read_text()):
        R_c2w = np.asarray(rec["rotation"], dtype=np.float64)
        pos = np.asarray(rec["position"], dtype=np.float64)
        T = -R_c2w.transpose() @ pos                             # world-to-camera translation
        w, h = int(rec["width"]), int(rec["height"])
        cams.append(Camera(colmap_id=rec["id"], R=R_c2w, T=T, FoVx=focal2fov(rec["fx"], w), FoVy=focal2fov(rec["fy"], h),
                           image=None, gt_alpha_mask=None, image_name=rec.get("img_name", str(rec["id"])), uid=rec["id"],
                           data_device=data_device, image_width=w, image_height=h))
    return cams


class Scene:
    def __init__(self, args, gaussians, load_iteration=None, shuffle=True, resolution_scales=(1.0,)):
        self.model_path = args.model_path
        self.loaded_iter = None
        self.gaussians = gaussians
        pc_dir = os.path.join(self.model_path, "point_cloud")
        if load_iteration:
            self.loaded_iter = searchForMaxIteration(pc_dir) if load_iteration == -1 else load_iteration
            print("Loading trained model at iteration {}".format(self.loaded_iter))
        device = getattr(args, "data_device", "cuda")
        cam_file = os.path.join(self.model_path, "cameras.json")
        cams = cameras_from_json(cam_file, device) if os.path.exists(cam_file) else []
        if shuffle:
            random.shuffle(cams)
        self.train_cameras = {float(s): cams for s in resolution_scales}
        self.test_cameras = {float(s): [] for s in resolution_scales}
        # scene radius as the training code expects it (spatial_lr_scale, prune thresholds): 1.1 x the largest distance of a
        # camera centre from the mean centre
        self.cameras_extent = 1.0
        if cams:
            centres = np.stack([np.asarray(c.camera_center.cpu(), dtype=np.float64) for c in cams])
            self.cameras_extent = float(1.1 * np.linalg.norm(centres - centres.mean(0), axis=1).max())
        if self.loaded_iter:
            self.gaussians.load_ply(os.path.join(pc_dir, "iteration_" + str(self.loaded_iter), "point_cloud.ply"))

    def save(self, iteration):
        self.gaussians.save_ply(os.path.join(self.model_path, "point_cloud/iteration_{}".format(iteration), "point_cloud.ply"))

    def getTrainCameras(self, scale=1.0):
        return self.train_cameras[float(scale)]

    def getTestCameras(self, scale=1.0):
        return self.test_cameras[float(scale)]
