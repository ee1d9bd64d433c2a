"""i2sdf_amd -- MI355X-native volume-rendering core for I2-SDF (hand-written HIP kernels behind the reference's
`I2SDFNetwork` module contract).  See DESIGN.md / INTEGRATION.md."""
from .config import NetConfig, SamplerConfig, synthetic_conf, plumbing_conf  # noqa: F401
