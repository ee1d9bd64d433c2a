"""Bubble-loss point-cloud PDF: the per-point sampling density the trainer maintains next to the render path.

Mirrors the three methods of VolumeRenderSystem that touch it (model/trainer/recon.py):
  update_pdf           :142-152   value clamp / prune / scatter through the pixel->point links -- here ONE kernel fused with the
                                  error it is fed (i2sdf_pdf_update; the reference spends ~10 launches + an index_put per call)
  sample_bubble        :154-170   uniform or importance sampling of bubble_batch_size points
  initialize_bubble_pdf:172-199   the sweep over every pixel of every training image with model.forward(data, True); here the
                                  rays come from the HBM-resident RayBatcher (no per-ray K / pose stacks) and each split costs one
                                  render + one update launch.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import lib as L_


class BubblePDF:
    """pointcloud (n_points,3), pointlinks (n_images*H*W,) int64 with -1 = no point (dataset/train_dataset.py:105-138)."""

    def __init__(self, pointcloud: torch.Tensor, pointlinks: torch.Tensor, pdf_criterion: str = "DEPTH", pdf_max: Optional[float] = None,
                 pdf_prune: float = 0.0, uniform_bubble: bool = False, device="cuda"):
        assert pdf_criterion in ("RGB", "DEPTH")                         # model/trainer/recon.py:55-56
        dev = torch.device(device)
        if dev.type != "cuda":
            raise L_.I2SDFError("BubblePDF needs a ROCm device (there is no CPU path)")
        self._lib = L_.load()
        self.device = dev
        self.pointcloud = torch.as_tensor(pointcloud).to(dev, torch.float32).contiguous()
        self.pointlinks = torch.as_tensor(pointlinks).to(dev, torch.int64).contiguous()
        self.pdf_criterion, self.pdf_max, self.pdf_prune, self.uniform_bubble = pdf_criterion, pdf_max, float(pdf_prune), uniform_bubble
        self.pdf = torch.zeros(self.pointcloud.shape[0], dtype=torch.float32, device=dev)
        self.sample_count = torch.zeros(self.pointcloud.shape[0], dtype=torch.float32, device=dev)
        self._n_bad = torch.zeros(1, dtype=torch.int32, device=dev)

    def bad_indices(self) -> int:
        return int(self._n_bad.item())

    # ------------------------------------------------------------------------------------------
    def update_pdf(self, model_outputs: Dict[str, torch.Tensor], ground_truth: Dict[str, torch.Tensor], indices=None, first_pixel: int = 0):
        """pdf[pointlinks[indices]] = pruned, clamped error of this batch.  `indices`: global pixel indices (B,), or None for the
        run first_pixel .. first_pixel+B-1."""
        if self.pdf_criterion == "RGB":
            pred, tgt, ch = model_outputs["rgb_values"], ground_truth["rgb"], 3
        else:
            pred, tgt, ch = model_outputs["depth_values"], ground_truth["depth"], 1
        pred = pred.detach().to(torch.float32).reshape(-1, ch).contiguous()
        tgt = tgt.detach().to(self.device, torch.float32).reshape(-1, ch).contiguous()
        if not pred.is_cuda:
            raise L_.I2SDFError("BubblePDF.update_pdf needs device tensors (there is no CPU path)")
        n = pred.shape[0]
        assert tgt.shape[0] == n
        if indices is not None:
            indices = torch.as_tensor(indices).to(self.device, torch.int64).contiguous()
            assert indices.numel() == n
        with torch.cuda.device(self.device):
            L_.check(self._lib.i2sdf_pdf_update(L_.ptr(pred), L_.ptr(tgt), ch, L_.ptr(indices), int(first_pixel), n, L_.ptr(self.pointlinks),
                                                self.pointlinks.numel(), math.nan if self.pdf_max is None else float(self.pdf_max),
                                                self.pdf_prune, L_.ptr(self.pdf), self.pdf.numel(), L_.ptr(self._n_bad), L_.stream_ptr()),
                     "i2sdf_pdf_update")

    def sample_bubble(self, batch_size: int) -> torch.Tensor:
        """model/trainer/recon.py:154-170 (torch.multinomial keeps the reference's random stream and its 2^24 category limit)."""
        if self.uniform_bubble:
            return self.pointcloud[torch.randperm(self.pointcloud.shape[0], device=self.device)[:batch_size]]
        sample_idx = torch.where(self.pdf > 0)[0]
        if sample_idx.shape[0] >= (1 << 24):
            raise RuntimeError("PDF capacity exceeds the 2^24 category limit of torch.multinomial")
        idx = torch.multinomial(self.pdf[sample_idx], batch_size, replacement=False)
        self.sample_count[sample_idx[idx]] += 1
        return self.pointcloud[sample_idx[idx]]

    @torch.no_grad()
    def initialize_bubble_pdf(self, model, batcher, split_size: int, images=None, draws_for=None):
        """The initial sweep: every pixel of every image rendered with model.forward(data, True) in the model's CURRENT mode (the
        reference calls it from training_step, i.e. training mode under no_grad: perturbed sampling, no eikonal / normal outputs)
        and scattered into the PDF.  `batcher`: the RayBatcher holding the training views.  `draws_for(image, first_pixel, n)`
        may return the sampler draws to use (tests)."""
        tp = batcher.total_pixels
        for i in (range(batcher.n_images) if images is None else images):
            for lo in range(0, tp, split_size):
                n = min(split_size, tp - lo)
                tidx = torch.arange(i * tp + lo, i * tp + lo + n, dtype=torch.int64, device=self.device)
                _, _, sample, gt = batcher.batch(tidx)
                out = model(sample, True) if draws_for is None else model(sample, True, draws=draws_for(i, lo, n))
                self.update_pdf(out, gt, None, first_pixel=i * tp + lo)
