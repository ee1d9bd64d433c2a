"""HBM-resident ray batcher.

Replaces the step right before the render path: `ReconDataset.__getitem__` + `collate_fn` + the 4-worker DataLoader
(dataset/train_dataset.py:166-209, model/trainer/recon.py:210-211), which assemble every batch on the host out of per-ray
dicts that each carry their own 4x4 intrinsics and pose (136 B/ray).  Here the camera tables and the ground-truth images are
uploaded once; a batch is a tensor of global pixel indices, and ONE kernel (`i2sdf_ray_batch`) turns it into rays
(cam_loc, unit dir, ||dir||) and gathers the ground truth.  The tuple returned by `batch()` has the reference's collate layout
`(tidx, image_idx, sample, ground_truth)`; `sample` additionally carries the ready rays so `I2SDFNetwork.forward` skips its own
ray set-up, and materialises the per-ray `intrinsics` / `pose` stacks only if somebody asks for them.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterator, Optional, Tuple

import torch

from . import lib as L_


class RaySample(dict):
    """`sample` dict of a batch.  'uv' (B,1,2) and 'rays' are stored; 'intrinsics' / 'pose' (the reference's per-ray stacks,
    dataset/train_dataset.py:173-177) are gathered from the tables on first access."""

    def __init__(self, batcher: "RayBatcher", image_idx: torch.Tensor, **kw):
        super().__init__(**kw)
        self._batcher, self._image_idx = batcher, image_idx

    def __missing__(self, key):
        if key == "intrinsics":
            v = self._batcher.intrinsics_all.index_select(0, self._image_idx)
        elif key == "pose":
            v = self._batcher.pose_all.index_select(0, self._image_idx)
        else:
            raise KeyError(key)
        self[key] = v
        return v


class RayBatcher:
    """Device-resident stand-in for ReconDataset as a batch source.

    intrinsics_all (n,4,4); pose_all (n,4,4) or (n,7); img_res = [H, W]; image tables in the dataset's layouts:
    rgb_images (n,HW,3) f32, mask_images / lightmask_images (n,HW,1) f32, depth_images (n,HW) f32, depth_masks (n,HW) bool,
    normal_images (n,HW,3) f32 (already world space, :158-160), normal_masks (n,HW) bool.
    """

    def __init__(self, intrinsics_all, pose_all, img_res, rgb_images=None, mask_images=None, lightmask_images=None, depth_images=None,
                 depth_masks=None, normal_images=None, normal_masks=None, device="cuda"):
        self._lib = L_.load()
        dev = torch.device(device)
        if dev.type != "cuda":
            raise L_.I2SDFError("RayBatcher needs a ROCm device (there is no CPU path)")
        f32 = lambda t: None if t is None else torch.as_tensor(t).to(dev, torch.float32).contiguous()
        u8 = lambda t: None if t is None else torch.as_tensor(t).to(dev, torch.bool).contiguous()
        self.device = dev
        self.intrinsics_all, self.pose_all = f32(intrinsics_all), f32(pose_all)
        self.n_images = self.intrinsics_all.shape[0]
        self.img_res = [int(img_res[0]), int(img_res[1])]
        self.total_pixels = self.img_res[0] * self.img_res[1]
        self.pose_is_quat = self.pose_all.dim() == 2 and self.pose_all.shape[1] == 7
        if not self.pose_is_quat and tuple(self.pose_all.shape[1:]) != (4, 4):
            raise ValueError(f"pose_all must be (n,4,4) or (n,7), got {tuple(self.pose_all.shape)}")
        if tuple(self.intrinsics_all.shape) != (self.n_images, 4, 4) or self.pose_all.shape[0] != self.n_images:
            raise ValueError("intrinsics_all must be (n,4,4) with one pose per image")
        self.rgb_images, self.mask_images, self.lightmask_images = f32(rgb_images), f32(mask_images), f32(lightmask_images)
        self.depth_images, self.normal_images = f32(depth_images), f32(normal_images)
        self.depth_masks, self.normal_masks = u8(depth_masks), u8(normal_masks)
        for name, t, tail in (("rgb_images", self.rgb_images, 3), ("mask_images", self.mask_images, 1),
                              ("lightmask_images", self.lightmask_images, 1), ("depth_images", self.depth_images, 1),
                              ("normal_images", self.normal_images, 3), ("depth_masks", self.depth_masks, 1),
                              ("normal_masks", self.normal_masks, 1)):
            if t is not None and t.numel() != self.n_images * self.total_pixels * tail:
                raise ValueError(f"{name} has {t.numel()} elements, expected {self.n_images}x{self.total_pixels}x{tail}")
        self.use_mask, self.use_lightmask = self.mask_images is not None, self.lightmask_images is not None
        self.use_depth, self.use_normal = self.depth_images is not None, self.normal_images is not None
        self._n_bad = torch.zeros(1, dtype=torch.int32, device=dev)
        self._tables = L_.RayTables(L_.ptr(self.intrinsics_all), L_.ptr(self.pose_all), int(self.pose_is_quat), self.n_images,
                                    self.img_res[0], self.img_res[1], L_.ptr(self.rgb_images), L_.ptr(self.depth_images),
                                    L_.ptr(self.normal_images), L_.ptr(self.mask_images), L_.ptr(self.lightmask_images),
                                    L_.ptr(self.depth_masks), L_.ptr(self.normal_masks))

    @classmethod
    def from_dataset(cls, ds, device="cuda") -> "RayBatcher":
        """Upload a (reference-style) ReconDataset's tables; attribute names follow dataset/train_dataset.py."""
        g = lambda flag, name: getattr(ds, name) if getattr(ds, flag, False) else None
        has_depth = getattr(ds, "use_depth", False) or getattr(ds, "use_bubble", False)
        return cls(ds.intrinsics_all, ds.pose_all, ds.img_res, rgb_images=ds.rgb_images, mask_images=g("use_mask", "mask_images"),
                   lightmask_images=g("use_lightmask", "lightmask_images"),
                   depth_images=ds.depth_images if has_depth else None, depth_masks=ds.depth_masks if has_depth else None,
                   normal_images=g("use_normal", "normal_images"), normal_masks=g("use_normal", "normal_masks"), device=device)

    def __len__(self) -> int:
        return self.n_images * self.total_pixels

    def bad_indices(self) -> int:
        """Number of out-of-range pixel indices seen in device-side index tensors so far (one device->host read)."""
        return int(self._n_bad.item())

    def batch(self, tidx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, RaySample, Dict[str, torch.Tensor]]:
        """tidx: global pixel indices (what the DataLoader's sampler yields) -> (tidx, image_idx, sample, ground_truth)."""
        tidx = torch.as_tensor(tidx)
        if tidx.dim() != 1:
            raise ValueError("tidx must be 1-D")
        if not tidx.is_cuda and tidx.numel():
            # host indices are validated here (free); device indices are clamped by the kernel and counted in `bad_indices`
            lo, hi = int(tidx.min()), int(tidx.max())
            if lo < 0 or hi >= len(self):
                raise IndexError(f"pixel index out of range [0, {len(self)}): min {lo}, max {hi}")
        tidx = tidx.to(self.device, torch.int64).contiguous()
        B, dev = tidx.shape[0], self.device
        e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=dev)
        image_idx, uv = e(B, dt=torch.int64), e(B, 1, 2)
        cam, dirs, dnorm = e(B, 3), e(B, 3), e(B)
        gt: Dict[str, torch.Tensor] = {}
        if self.rgb_images is not None:
            gt["rgb"] = e(B, 3)
        if self.use_mask:
            gt["mask"] = e(B, 1)
        if self.use_lightmask:
            gt["light_mask"] = e(B, 1)
        if self.use_depth:
            gt["depth"], gt["depth_mask"] = e(B), e(B, dt=torch.bool)
        if self.use_normal:
            gt["normal"], gt["normal_mask"] = e(B, 3), e(B, dt=torch.bool)
        out = L_.RayBatch(L_.ptr(image_idx), L_.ptr(uv), L_.ptr(cam), L_.ptr(dirs), L_.ptr(dnorm), L_.ptr(gt.get("rgb")),
                          L_.ptr(gt.get("depth")), L_.ptr(gt.get("normal")), L_.ptr(gt.get("mask")), L_.ptr(gt.get("light_mask")),
                          L_.ptr(gt.get("depth_mask")), L_.ptr(gt.get("normal_mask")), L_.ptr(self._n_bad))
        with torch.cuda.device(dev):
            L_.check(self._lib.i2sdf_ray_batch(C.byref(self._tables), L_.ptr(tidx), B, C.byref(out), L_.stream_ptr()), "i2sdf_ray_batch")
        sample = RaySample(self, image_idx, uv=uv, rays={"cam_loc": cam, "dirs": dirs, "dnorm": dnorm})
        return tidx, image_idx, sample, gt

    def epoch(self, batch_size: int, generator: Optional[torch.Generator] = None, rank: int = 0, world_size: int = 1,
              drop_last: bool = False) -> Iterator[Tuple[torch.Tensor, torch.Tensor, RaySample, Dict[str, torch.Tensor]]]:
        """One shuffled pass over all pixels of all images, like DataLoader(shuffle=True) (model/trainer/recon.py:210-211).
        With world_size > 1 every rank must pass a generator seeded identically: the permutation is cut into disjoint
        strided slices (DistributedSampler semantics), so no collective is needed."""
        perm = torch.randperm(len(self), generator=generator, device=generator.device if generator is not None else "cpu")
        if world_size > 1:
            per = len(self) // world_size
            perm = perm[: per * world_size][rank::world_size]
        perm = perm.to(self.device)
        for s in range(0, perm.shape[0], batch_size):
            chunk = perm[s:s + batch_size]
            if drop_last and chunk.shape[0] < batch_size:
                return
            yield self.batch(chunk)
