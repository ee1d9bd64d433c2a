"""Import the upstream reference (/root/reference) in THIS container with stub modules
for its absent, off-path dependencies (pytorch_lightning, cv2, ...).

Used ONLY by tests/golden/gen_golden.py (fixture generation) and by
tests/test_oracle_vs_reference.py (skipped when /root/reference is absent, i.e. on the GPU box).
Nothing here is shipped or imported by the product path.
"""
import importlib.machinery
import os
import sys
import types

REF_ROOT = os.environ.get("I2SDF_REFERENCE", "/root/reference")


class _Anything:
    """Attribute/call-tolerant placeholder (default args like cv2.COLORMAP_VIRIDIS)."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()


def _stub(name):
    m = _StubModule(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    return m


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "model", "network"))


def import_reference():
    """Returns (model_module, utils_module) of the reference."""
    if "model" in sys.modules and getattr(sys.modules["model"], "__file__", "").startswith(REF_ROOT):
        return sys.modules["model"], sys.modules["utils"]
    import torch.nn as nn

    for name in [
        "cv2", "imageio", "skimage", "skimage.measure", "skimage.transform", "GPUtil", "torchmetrics",
        "torchmetrics.image", "torchmetrics.image.lpip", "torchmetrics.functional", "trimesh", "mcubes", "open3d",
        "lpips", "fast_pytorch_kmeans", "torchvision", "torchvision.utils", "torchvision.transforms", "tensorboard",
        "pytorch_lightning", "pytorch_lightning.callbacks", "pytorch_lightning.loggers",
        "pytorch_lightning.callbacks.progress", "pytorch_lightning.callbacks.progress.rich_progress",
        "plotly", "plotly.graph_objs", "plotly.offline", "plotly.subplots", "matplotlib", "matplotlib.pyplot",
        "PIL", "PIL.Image", "pyrender", "ffmpeg",
    ]:
        try:
            __import__(name)
        except Exception:
            _stub(name)
    pl = sys.modules["pytorch_lightning"]
    if isinstance(pl, _StubModule):
        class LightningModule(nn.Module):
            def log(self, *a, **k):
                pass
        pl.LightningModule = LightningModule
        class _Bar:
            def __init__(self, *a, **k):
                pass
        sys.modules["pytorch_lightning.callbacks"].RichProgressBar = _Bar
        sys.modules["pytorch_lightning.callbacks"].ModelCheckpoint = _Bar
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import model as ref_model  # noqa
    import utils as ref_utils  # noqa
    return ref_model, ref_utils


def load_cfg(name="synthetic.yml"):
    import yaml
    _, ref_utils = import_reference()
    with open(os.path.join(REF_ROOT, "config", name)) as f:
        return ref_utils.CfgNode(yaml.safe_load(f))
