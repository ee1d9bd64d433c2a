"""Flat parameter buffer: layout (the reference's state_dict order), C descriptor, reference-scheme initialisation."""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

from .config import MlpShape, NetConfig
from .lib import MAX_LAYERS, MlpDesc, NetDesc


class ParamLayout:
    """name -> (offset, shape) inside one flat fp32 buffer, in the order of the reference's state_dict:
    implicit_network.lin{l}.{bias,weight_g,weight_v}, rendering_network..., [light_network...], density.beta."""

    def __init__(self, cfg: NetConfig):
        self.cfg = cfg
        self.entries: List[Tuple[str, int, Tuple[int, ...]]] = []
        off = 0
        for net in self.nets():
            for l, (out, inn) in enumerate(net.dims):
                for suffix, shape in (("bias", (out,)), ("weight_g", (out, 1)), ("weight_v", (out, inn))):
                    self.entries.append((f"{net.name}.lin{l}.{suffix}", off, shape))
                    off += int(np.prod(shape))
        self.entries.append(("density.beta", off, ()))
        off += 1
        self.n_params = off
        self.index: Dict[str, Tuple[int, Tuple[int, ...]]] = {n: (o, s) for n, o, s in self.entries}

    def nets(self) -> List[MlpShape]:
        c = self.cfg
        return [c.sdf, c.rgb] + ([c.light] if c.light is not None else [])

    def offset(self, name: str) -> int:
        return self.index[name][0]

    def _mlp_desc(self, net: MlpShape, d_out: int) -> MlpDesc:
        d = MlpDesc()
        if net.n_lin > MAX_LAYERS:
            raise ValueError("too many layers")
        d.n_lin, d.hidden, d.d_in, d.in0, d.d_out = net.n_lin, net.hidden, net.d_in, net.dims[0][1], d_out
        d.multires, d.skip_layer = net.multires, net.skip_layer
        for l, (out, inn) in enumerate(net.dims):
            d.out_dim[l], d.in_dim[l] = out, inn
            d.off_bias[l] = self.offset(f"{net.name}.lin{l}.bias")
            d.off_g[l] = self.offset(f"{net.name}.lin{l}.weight_g")
            d.off_v[l] = self.offset(f"{net.name}.lin{l}.weight_v")
        return d

    def net_desc(self) -> NetDesc:
        c = self.cfg
        nd = NetDesc()
        nd.sdf = self._mlp_desc(c.sdf, c.sdf.dims[-1][0])
        nd.rgb = self._mlp_desc(c.rgb, c.rgb.dims[-1][0])
        if c.light is not None:
            nd.light = self._mlp_desc(c.light, 1)
        nd.off_beta = self.offset("density.beta")
        nd.n_params = self.n_params
        nd.beta_min = c.beta_min
        nd.scene_bounding_sphere = c.scene_bounding_sphere
        return nd

    # ------------------------------------------------------------------------------------------
    def init_flat(self, generator: torch.Generator = None) -> torch.Tensor:
        """Reference initialisation scheme, written into a new flat CPU tensor.
        SDF net: geometric init (model/network/mlp.py:55-69); radiance / light nets: nn.Linear default
        (U(+-1/sqrt(in)) for weight and bias); weight_norm: v = W, g = row norms (mlp.py:71-72);
        density.beta from the config (density.py:6-8)."""
        c = self.cfg
        flat = torch.zeros(self.n_params, dtype=torch.float32)

        def put(prefix, W, b):
            for suffix, val in (("bias", b), ("weight_g", W.norm(dim=1, keepdim=True)), ("weight_v", W)):
                off, shape = self.index[f"{prefix}.{suffix}"]
                flat[off:off + val.numel()] = val.reshape(-1).to(torch.float32)

        pe = c.sdf.pe_dim
        n = c.sdf.n_lin
        for l, (out, inn) in enumerate(c.sdf.dims):
            if l == n - 1:
                W = torch.randn(out, inn, generator=generator) * 1e-4 + math.sqrt(math.pi) / math.sqrt(inn)
                b = torch.full((out,), -c.sdf_bias)
            else:
                W = torch.randn(out, inn, generator=generator) * (math.sqrt(2) / math.sqrt(out))
                b = torch.zeros(out)
                if l == 0:
                    W[:, 3:] = 0.0
                elif l == c.sdf.skip_layer:
                    W[:, -(pe - 3):] = 0.0
            put(f"implicit_network.lin{l}", W, b)
        for net in self.nets()[1:]:
            for l, (out, inn) in enumerate(net.dims):
                bound = 1.0 / math.sqrt(inn)
                W = (torch.rand(out, inn, generator=generator) * 2 - 1) * bound
                b = (torch.rand(out, generator=generator) * 2 - 1) * bound
                put(f"{net.name}.lin{l}", W, b)
        flat[self.offset("density.beta")] = c.beta_init
        return flat

    def flat_from_state_dict(self, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
        flat = torch.zeros(self.n_params, dtype=torch.float32)
        for name, off, shape in self.entries:
            v = sd[name].detach().to(torch.float32).reshape(-1).cpu()
            assert v.numel() == int(np.prod(shape)) if shape else v.numel() == 1, name
            flat[off:off + v.numel()] = v
        return flat

    def state_dict_from_flat(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = {}
        for name, off, shape in self.entries:
            n = int(np.prod(shape)) if shape else 1
            out[name] = flat[off:off + n].reshape(shape).clone()
        return out
