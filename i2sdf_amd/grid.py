"""Marching-cubes grids: the axis vectors of the reference's two grids, and nothing else on the host.

The reference builds every grid point with np.meshgrid and ships the (n,3) tensor through a DataLoader (utils/plots.py:440-489,
model/eval/recon.py:46-51,75-103).  `I2SDFNetwork.sdf_volume` only needs the three axis vectors returned here; the points are
generated on the device (csrc/grid.hip).  Axis arithmetic is float64 numpy like the reference's, cast to fp32 at upload.
"""
from __future__ import annotations

from typing import NamedTuple, Optional, Sequence, Tuple

import numpy as np


class GridAxes(NamedTuple):
    """x, y, z axis coordinates (float64) + the two extras the reference's grid dicts carry."""
    x: np.ndarray
    y: np.ndarray
    z: np.ndarray
    shortest_axis_length: float
    shortest_axis_index: int

    @property
    def xyz(self):
        return [self.x, self.y, self.z]

    @property
    def shape_volume(self) -> Tuple[int, int, int]:
        """(nx, ny, nz): the shape of the volume handed to marching cubes (the reference's reshape(ny,nx,nz).transpose(1,0,2))."""
        return (self.x.shape[0], self.y.shape[0], self.z.shape[0])

    @property
    def spacing(self) -> Tuple[float, float, float]:
        """marching_cubes `spacing`: the reference uses the x step x[2]-x[1] for all three axes (model/eval/recon.py:57-59,97-99)."""
        s = float(self.x[2] - self.x[1])
        return (s, s, s)

    @property
    def origin(self) -> np.ndarray:
        return np.array([self.x[0], self.y[0], self.z[0]])


def uniform_axes(resolution: int, grid_boundary: Sequence[float] = (-2.0, 2.0)) -> GridAxes:
    """Cube grid of resolution^3 points over [lo, hi]^3 -- get_grid_uniform (utils/plots.py:440-451)."""
    a = np.linspace(grid_boundary[0], grid_boundary[1], resolution)
    return GridAxes(a, a, a, 2.0, 0)


def aligned_axes(points, resolution: int, input_min: Optional[np.ndarray] = None, input_max: Optional[np.ndarray] = None,
                 eps: float = 0.1) -> GridAxes:
    """Grid over the bounding box of `points` (n,3) grown by eps: `resolution` points along the SHORTEST axis, the same step along
    the other two -- get_grid (utils/plots.py:453-489)."""
    if input_min is None or input_max is None:
        pts = np.asarray(points.detach().cpu().numpy() if hasattr(points, "detach") else points)
        input_min, input_max = pts.min(axis=0), pts.max(axis=0)
    input_min, input_max = np.asarray(input_min), np.asarray(input_max)
    s = int(np.argmin(input_max - input_min))
    short = np.linspace(input_min[s] - eps, input_max[s] + eps, resolution)
    length = np.max(short) - np.min(short)
    step = length / (short.shape[0] - 1)
    axes = [short if a == s else np.arange(input_min[a] - eps, input_max[a] + step + eps, step) for a in range(3)]
    return GridAxes(axes[0], axes[1], axes[2], float(length), s)
