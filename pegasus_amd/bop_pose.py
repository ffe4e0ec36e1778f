"""BOP ground-truth pose records on the batch path.

Reference (the surviving copy of the writer): /root/reference/src/tools/pegasus_working.py:441-455,457-576.
For every frame and object:  T_m2c = T_w2c @ T_m2w  with  T_w2c[:3,:3] = cam.R.T, T_w2c[:3,3] = cam.T
(pegasus_working.py:464-466);  scene_gt entry (pegasus_working.py:565-576) = {"cam_R_m2c": 9 floats row-major,
"cam_t_m2c": 3 floats, "T_w2c": 16, "T_m2w": 16, "obj_id", "bullet_obj_id"} and -- when the caller supplies the object's
model-space bounding box (the reference takes the 8 corners of the mesh's minimal oriented box from open3d, which is out of
scope here, in the NDDS corner order of pegasus_working.py:470-495) -- "3d_bounding_box_model_coord" [8,3],
"3d_bounding_center" [3], "projected_points" [8,2] and "projected_center" [1,2] with P = K @ T_m2c[:3] applied to the
homogeneous points (pegasus_working.py:547-563);  scene_camera entry = {"cam_K": 9 floats, "depth_scale"} with K from the FoV
(fov2focal, pegasus_working.py:349-356).  Format: submodules/bop_toolkit/docs/bop_datasets_format.md:75-109.

Quirk kept from the reference and flagged: translations stay in METRES while depth images are written in
millimetres (pegasus.py:355); set ``translation_scale=1000.0`` for BOP-conformant millimetres.
"""
from __future__ import annotations

import numpy as np

from .graphics import fov2focal


def world_to_camera(R_c2w, t_w2c) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = np.asarray(R_c2w, dtype=np.float64).T
    T[:3, 3] = np.asarray(t_w2c, dtype=np.float64).reshape(3)
    return T


def camera_K(fovx, fovy, width, height) -> np.ndarray:
    fx, fy = fov2focal(fovx, width), fov2focal(fovy, height)
    return np.array([[fx, 0, width / 2.0], [0, fy, height / 2.0], [0, 0, 1.0]])


def project_points(K, T_m2c, points) -> np.ndarray:
    """[n,2] pixel coordinates of model-space points: P = K @ T_m2c[:3] on homogeneous points, then the division
    cv2.convertPointsFromHomogeneous does (a zero last coordinate divides by 1)   (pegasus_working.py:547-563)."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    hom = np.concatenate([pts, np.ones((pts.shape[0], 1))], axis=1)
    proj = (np.asarray(K, dtype=np.float64).reshape(3, 3) @ np.asarray(T_m2c, dtype=np.float64)[:3]) @ hom.T      # [3,n]
    w = np.where(proj[2] != 0.0, proj[2], 1.0)
    return (proj[:2] / w).T


def scene_gt_entry(R_c2w, t_w2c, object_poses_m2w: dict, translation_scale: float = 1.0, K=None, boxes: dict = None,
                   dataset_ids: dict = None) -> list:
    """object_poses_m2w: {bullet_obj_id: 4x4 model-to-world}.  Returns the scene_gt list for one image with the fields of
    the reference's writer (pegasus_working.py:565-576).  ``dataset_ids``: {bullet_obj_id: the object's dataset ID (the
    reference's meta_info.ID)}, default: the same number.  ``boxes``: {bullet_obj_id: (corners [8,3], center [3])} in
    model coordinates + ``K`` [3,3] add the box and its projection."""
    T_w2c = world_to_camera(R_c2w, t_w2c)
    out = []
    for obj_id, T_m2w in object_poses_m2w.items():
        T_m2w = np.asarray(T_m2w, dtype=np.float64).reshape(4, 4)
        T = T_w2c @ T_m2w
        e = {"cam_R_m2c": T[:3, :3].reshape(-1).tolist(),
             "cam_t_m2c": (T[:3, 3] * translation_scale).tolist(),
             "T_w2c": T_w2c.reshape(-1).tolist(), "T_m2w": T_m2w.reshape(-1).tolist(),
             "obj_id": int((dataset_ids or {}).get(obj_id, obj_id)), "bullet_obj_id": int(obj_id)}
        if boxes is not None and obj_id in boxes:
            if K is None:
                raise ValueError("boxes need the camera matrix K for their projection")
            corners, center = boxes[obj_id]
            corners = np.asarray(corners, dtype=np.float64).reshape(8, 3)
            center = np.asarray(center, dtype=np.float64).reshape(3)
            e["3d_bounding_box_model_coord"] = corners.tolist()
            e["3d_bounding_center"] = center.tolist()
            e["projected_center"] = project_points(K, T, center[None]).tolist()
            e["projected_points"] = project_points(K, T, corners).tolist()
        out.append(e)
    return out


def scene_camera_entry(fovx, fovy, width, height, R_c2w=None, t_w2c=None, depth_scale: float = 1.0) -> dict:
    e = {"cam_K": camera_K(fovx, fovy, width, height).reshape(-1).tolist(), "depth_scale": depth_scale}
    if R_c2w is not None:
        T = world_to_camera(R_c2w, t_w2c)
        e["cam_R_w2c"] = T[:3, :3].reshape(-1).tolist()
        e["cam_t_w2c"] = T[:3, 3].tolist()
    return e


def batch_pose_records(views, object_poses_per_frame, translation_scale: float = 1.0, boxes: dict = None,
                       dataset_ids: dict = None):
    """views: sequence with R_c2w, t_w2c, fovx, fovy, width, height (pegasus_amd.scenes.View or Camera-like);
    object_poses_per_frame: one {obj_id: 4x4} dict per view (or a single dict for a static scene); ``boxes`` /
    ``dataset_ids`` as in scene_gt_entry."""
    gt, cam = {}, {}
    for i, v in enumerate(views):
        poses = object_poses_per_frame if isinstance(object_poses_per_frame, dict) else object_poses_per_frame[i]
        R = getattr(v, "R_c2w", getattr(v, "R", None))
        t = getattr(v, "t_w2c", getattr(v, "T", None))
        fx, fy = getattr(v, "fovx", getattr(v, "FoVx", None)), getattr(v, "fovy", getattr(v, "FoVy", None))
        w, h = getattr(v, "width", getattr(v, "image_width", None)), getattr(v, "height", getattr(v, "image_height", None))
        gt[str(i)] = scene_gt_entry(R, t, poses, translation_scale, K=camera_K(fx, fy, w, h) if boxes is not None else None,
                                    boxes=boxes, dataset_ids=dataset_ids)
        cam[str(i)] = scene_camera_entry(fx, fy, w, h, R, t)
    return gt, cam


def gt_info_from_masks(visible, silhouette, depth_valid=None):
    """scene_gt_info entries (/root/reference/submodules/bop_toolkit/docs/bop_datasets_format.md:116-129) from the masks the
    frame path already holds: ``silhouette`` [..., K, H, W] (FrameRenderer.render_silhouettes: every object alone) and
    ``visible`` [..., K, H, W] (the frames' masks), ``depth_valid`` [..., H, W] (depth image != 0; default: all valid).
    Torch tensors (any device) or numpy arrays of 0/1.  Follows calc_gt_info.py:139-172 of the BOP toolkit: px_count_all /
    px_count_valid / px_count_visib, visib_fract = visible / all, bounding boxes (x, y, w, h) with w = x_max - x_min
    (bop_toolkit_lib.misc.calc_2d_bbox) and both boxes [-1, -1, -1, -1] when no pixel is visible.  The toolkit counts the
    silhouette of a mesh render including the part beyond the image border; a mask only has the part inside.
    Returns a dict of integer / float arrays shaped [..., K] (boxes [..., K, 4]) on the host."""
    import torch
    vis = torch.as_tensor(visible).bool()
    sil = torch.as_tensor(silhouette).bool().to(vis.device)
    H, W = vis.shape[-2:]
    valid = sil if depth_valid is None else sil & torch.as_tensor(depth_valid).bool().to(vis.device).unsqueeze(-3)

    def box(m):
        rows, cols = m.any(-1), m.any(-2)                        # [..., K, H], [..., K, W]
        first = lambda t: t.to(torch.uint8).argmax(-1)           # index of the first True (0 when there is none)
        y0, x0 = first(rows), first(cols)
        y1, x1 = H - 1 - first(rows.flip(-1)), W - 1 - first(cols.flip(-1))
        return torch.stack([x0, y0, x1 - x0, y1 - y0], -1)
    count = lambda m: m.flatten(-2).sum(-1)
    px_all, px_valid, px_visib = count(sil), count(valid), count(vis)
    seen = (px_visib > 0).unsqueeze(-1)
    none = torch.full_like(box(sil), -1)
    out = dict(px_count_all=px_all, px_count_valid=px_valid, px_count_visib=px_visib,
               visib_fract=torch.where(px_all > 0, px_visib.double() / px_all.clamp(min=1).double(), torch.zeros_like(px_all, dtype=torch.float64)),
               bbox_obj=torch.where(seen, box(sil), none), bbox_visib=torch.where(seen, box(vis), none))
    return {k: v.cpu().numpy() for k, v in out.items()}


def scene_gt_info_entry(info: dict, index) -> list:
    """The scene_gt_info.json list of one image from gt_info_from_masks' arrays: ``index`` selects the image (e.g. a batch
    position); one dict per object, in the order of the masks' K axis (= the order of scene_gt's entries)."""
    sel = {k: v[index] for k, v in info.items()}
    return [{"px_count_all": int(sel["px_count_all"][k]), "px_count_valid": int(sel["px_count_valid"][k]),
             "px_count_visib": int(sel["px_count_visib"][k]), "visib_fract": float(sel["visib_fract"][k]),
             "bbox_obj": [int(e) for e in sel["bbox_obj"][k]], "bbox_visib": [int(e) for e in sel["bbox_visib"][k]]}
            for k in range(sel["px_count_all"].shape[0])]
