"""Blender / nerf_synthetic loader -- counterpart of the reference's dataLoader/blender.py (BlenderDataset :21-260) and
the pinhole ray construction of dataLoader/ray_utils.py:23-41,65-85.  Reads `transforms_{split}.json` and the RGBA PNG
frames; produces the tensors train.py / renderer.py consume:

    all_rays [N*h*w, 6] (origin | unit direction)     all_rgbs [N*h*w, 4] RGBA in [0,1] (train) / [N,h,w,3] (is_stack)
    img_wh, near_far, scene_bbox (+-1.5 * aabb_scale), white_bg, poses [N,4,4], fx / fy, intrinsics

Pixel centres at +0.5, camera looks down +z after the blender->opencv flip (:45-47), directions unit-normalised (:120-122),
fx = 0.5 w / tan(camera_angle_x / 2) (:97-104).  EXR frames / normal maps / depth are outside the microfacet_tensorf2 path.
"""
import json
import os

import numpy as np
import torch

BLENDER2OPENCV = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=np.float64)


def get_ray_directions(H, W, focal, center=None):
    """dataLoader/ray_utils.py:23-41 (kornia.create_meshgrid restated: x = column, y = row, +0.5 pixel centre)"""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    i, j = xs + 0.5, ys + 0.5
    cent = center if center is not None else [W / 2, H / 2]
    return torch.stack([(i - cent[0]) / focal[0], (j - cent[1]) / focal[1], torch.ones_like(i)], -1)


def get_rays(directions, c2w):
    """dataLoader/ray_utils.py:65-85: rotate camera-space directions into the world, origin = camera centre"""
    rays_d = directions @ c2w[:3, :3].T
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    return rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)


def _read_image(path, wh=None):
    from PIL import Image
    img = Image.open(path)
    if wh is not None and tuple(img.size) != tuple(wh):
        img = img.resize(tuple(wh), Image.LANCZOS)
    a = np.asarray(img)
    if a.dtype == np.uint8:
        a = a.astype(np.float32) / 255.0                      # torchvision ToTensor
    elif a.dtype == np.uint16:
        a = a.astype(np.float32) / 65535.0
    if a.ndim == 2:
        a = a[..., None]
    return torch.from_numpy(np.ascontiguousarray(a)).float()   # [h, w, c]


class BlenderDataset(torch.utils.data.Dataset):
    def __init__(self, datadir, stack_norms=False, split="train", downsample=1.0, is_stack=False, N_vis=-1, white_bg=True,
                 is_testing=False):
        if stack_norms:
            raise NotImplementedError("normal maps are not used by model=microfacet_tensorf2")
        self.downsample, self.N_vis, self.root_dir, self.split, self.is_stack = downsample, N_vis, datadir, split, is_stack
        self.white_bg = white_bg
        self.is_testing = is_testing or split == "test"
        self.scene_bbox = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
        self.center = torch.mean(self.scene_bbox, dim=0).float().view(1, 1, 3)
        self.radius = (self.scene_bbox[1] - self.center).float().view(1, 1, 3)
        self.hdr = False
        self.read_meta()
        self.proj_mat = self.intrinsics.unsqueeze(0) @ torch.inverse(self.poses)[:, :3]

    def read_meta(self):
        with open(os.path.join(self.root_dir, f"transforms_{self.split}.json")) as f:
            meta = self.meta = json.load(f)
        ext = meta.get("ext", ".png")
        if "exr" in ext:
            raise NotImplementedError("EXR frames need an OpenEXR reader (not on the microfacet_tensorf2 benchmark path)")
        self.near_far = meta.get("near_far", [2.0, 6.0])
        self.white_bg = meta.get("white_bg", self.white_bg)
        meta.setdefault("w", 800)
        meta.setdefault("h", 800)
        w, h = int(meta["w"] / self.downsample), int(meta["h"] / self.downsample)
        self.img_wh = [w, h]
        if "aabb_scale" in meta:
            self.scene_bbox = self.scene_bbox * meta["aabb_scale"]
            self.radius = self.radius * meta["aabb_scale"]
        if "camera_angle_x" in meta:
            self.fx = self.fy = 0.5 * w / np.tan(0.5 * meta["camera_angle_x"])
        else:
            self.fx, self.fy = meta["fl_x"], meta["fl_y"]
        directions = get_ray_directions(h, w, [self.fx, self.fy])
        self.directions = directions / torch.norm(directions, dim=-1, keepdim=True)
        self.intrinsics = torch.tensor([[self.fx, 0, w / 2], [0, self.fy, h / 2], [0, 0, 1]]).float()
        self.image_paths, self.poses, self.all_rays, self.all_rgbs, self.acc_maps = [], [], [], [], []
        frames = meta["frames"]
        interval = 1 if self.N_vis < 0 else max(len(frames) // self.N_vis, 1)
        for i in range(0, len(frames), interval):
            frame = frames[i]
            pose = np.array(frame["transform_matrix"], dtype=np.float64) @ BLENDER2OPENCV
            c2w = torch.FloatTensor(pose)
            self.poses.append(c2w)
            path = os.path.join(self.root_dir, f"{frame['file_path']}{ext}")
            self.image_paths.append(path)
            img = _read_image(path, self.img_wh if self.downsample != 1.0 else None)          # [h, w, c]
            if img.shape[-1] == 4:
                self.acc_maps.append(img[..., -1])
            img = img.reshape(-1, img.shape[-1])
            if img.shape[1] == 4 and self.is_testing:
                img[:, :3] = img[:, :3] * img[:, -1:] + (1 - img[:, -1:])                      # blend A onto white (:165-169)
            rays_o, rays_d = get_rays(self.directions, c2w)
            self.all_rays.append(torch.cat([rays_o, rays_d], 1))
            self.all_rgbs.append(img)
        self.poses = torch.stack(self.poses)
        c = self.all_rgbs[0].shape[1]
        if not self.is_stack:
            self.all_rays = torch.cat(self.all_rays, 0)
            self.all_rgbs = torch.cat(self.all_rgbs, 0)
        else:
            self.all_rays = torch.stack(self.all_rays, 0)
            self.all_rgbs = torch.stack(self.all_rgbs, 0).reshape(-1, *self.img_wh[::-1], c)[..., :3]

    def world2ndc(self, points, lindisp=None):
        return (points - self.center.to(points.device)) / self.radius.to(points.device)

    def __len__(self):
        return len(self.all_rgbs)

    def __getitem__(self, idx):
        return {"rays": self.all_rays[idx], "rgbs": self.all_rgbs[idx]}


dataset_dict = {"blender": BlenderDataset}
