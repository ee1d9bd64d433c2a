"""I2SDFLoss with the reference's semantics (model/network/__init__.py:289-406), including its quirks:
`angular_loss` is the same L1 form as `normal_loss` (:368-369) and `angular_weight` defaults to 0.05 although the
shipped configs omit it (:290).  Plain torch ops on tiny per-ray tensors (outside the kernel path)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class I2SDFLoss(nn.Module):
    def __init__(self, eikonal_weight=0.1, smooth_weight=0.0, mask_weight=0.0, depth_weight=0.1, normal_weight=0.05, angular_weight=0.05,
                 bubble_weight=0.0, min_bubble_iter=0, max_bubble_iter=None, smooth_iter=None, light_mask_weight=0.0,
                 eikonal_weight_bubble=0.0):
        super().__init__()
        self.eikonal_weight, self.smooth_weight, self.mask_weight = eikonal_weight, smooth_weight, mask_weight
        self.depth_weight, self.normal_weight, self.angular_weight = depth_weight, normal_weight, angular_weight
        self.bubble_weight, self.min_bubble_iter, self.max_bubble_iter = bubble_weight, min_bubble_iter, max_bubble_iter
        self.smooth_iter = smooth_iter
        if self.bubble_weight > 0 and self.max_bubble_iter is not None and self.smooth_iter < self.max_bubble_iter:
            self.smooth_iter = self.max_bubble_iter
        self.light_mask_weight = light_mask_weight

    @staticmethod
    def _normal_l1(normal, normal_gt, mask):
        m = mask.flatten()
        return torch.abs(1 - torch.sum(normal[m] * normal_gt.reshape(-1, 3)[m], dim=-1)).mean()

    def forward(self, out, gt, current_step):
        dev = out["rgb_values"].device
        zero = lambda: torch.tensor(0.0, device=dev).float()
        rgb_loss = F.l1_loss(out["rgb_values"], gt["rgb"].reshape(-1, 3))
        eik = ((out["grad_theta"].norm(2, dim=1) - 1) ** 2).mean() if "grad_theta" in out else zero()
        smooth_on = self.smooth_iter is None or current_step > self.smooth_iter
        smooth = out["diff_norm"].mean() if (smooth_on and self.smooth_weight > 0 and "diff_norm" in out) else zero()
        if "mask" in gt and self.mask_weight > 0:
            mask = F.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1.0 - 1e-3), gt["mask"])
        else:
            mask = zero()
        if "depth" in gt and self.depth_weight > 0:
            dm = gt["depth_mask"].flatten()
            depth = F.mse_loss(out["depth_values"][dm], gt["depth"].flatten()[dm])
        else:
            depth = zero()
        normal = self._normal_l1(out["normal_values"], gt["normal"], gt["normal_mask"]) if ("normal" in gt and self.normal_weight > 0) else zero()
        angular = self._normal_l1(out["normal_values"], gt["normal"], gt["normal_mask"]) if ("normal" in gt and self.angular_weight > 0) else zero()
        bubble = out["surface_sdf"].abs().mean() if ("surface_sdf" in out and self.bubble_weight > 0) else zero()
        if "light_mask" in out and self.light_mask_weight > 0:
            lm = F.binary_cross_entropy(out["light_mask"].reshape(-1, 1).clip(1e-3, 1.0 - 1e-3), gt["light_mask"].reshape(-1, 1))
        else:
            lm = zero()
        loss = (rgb_loss + self.eikonal_weight * eik + self.smooth_weight * smooth + self.mask_weight * mask + self.depth_weight * depth
                + self.normal_weight * normal + self.angular_weight * angular + self.bubble_weight * bubble + self.light_mask_weight * lm)
        return {"loss": loss, "rgb_loss": rgb_loss, "eikonal_loss": eik, "smooth_loss": smooth, "mask_loss": mask, "depth_loss": depth,
                "normal_loss": normal, "angular_loss": angular, "bubble_loss": bubble, "light_mask_loss": lm}
