"""i2sdf_amd -- MI355X-native volume-rendering core for I2-SDF (hand-written HIP kernels behind the reference's
`I2SDFNetwork` module contract).  See DESIGN.md / INTEGRATION.md."""
from .config import NetConfig, SamplerConfig, synthetic_conf, plumbing_conf  # noqa: F401


def __getattr__(name):
    # torch-dependent pieces are imported lazily so that `import i2sdf_amd` stays cheap
    if name in ("I2SDFNetwork", "ImplicitNetwork", "RenderingNetwork", "LaplaceDensity"):
        from . import network
        return getattr(network, name)
    if name == "I2SDFLoss":
        from .loss import I2SDFLoss
        return I2SDFLoss
    if name in ("RayBatcher", "RaySample"):
        from . import batcher
        return getattr(batcher, name)
    if name == "FusedAdam":
        from .optim import FusedAdam
        return FusedAdam
    if name == "BubblePDF":
        from .bubble import BubblePDF
        return BubblePDF
    if name in ("GridAxes", "uniform_axes", "aligned_axes"):
        from . import grid
        return getattr(grid, name)
    if name == "RenderEngine":
        from .engine import RenderEngine
        return RenderEngine
    raise AttributeError(name)
