"""Data parallelism for the render core: one process per GPU, rays sharded, ONE collective per training step.

The reference has no distributed code at all (main_recon.py:111-112 `strategy=None`).  Rays are
independent, so the only exchange step of the path is the parameter-gradient mean: the backward writes every gradient into one
flat fp32 buffer (800 955 floats = 3.2 MB at synthetic.yml shapes) and that buffer is all-reduced once -- by the library's own
RCCL communicator (`i2sdf_allreduce_grads`, include/i2sdf.h) when the process group's backend is nccl (= RCCL on ROCm), through
torch.distributed otherwise (gloo: CPU tests, or several ranks sharing one GPU).  Inference shards the pixel list; results are
concatenated (no collective on the data path).

`equivalent=True` additionally makes a sharded training step EQUAL to the single-GPU step on the concatenated batch (SURVEY.md
8e): the sampler's batch-global convergence OR (ray_sampler.py:151) is reduced over all ranks on the device, the 32 extra
`randperm` columns are shared (ray_sampler.py:223: one draw for all rays of a batch), and the loss uses global denominators for
its (masked) means (model/network/__init__.py:320-336).  Throughput runs leave it off: per-rank flags, no extra exchanges."""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from . import lib as L


# ---------------------------------------------------------------------------------------------------------------------
# the library's RCCL communicator
# ---------------------------------------------------------------------------------------------------------------------
class RcclComm:
    """`i2sdf_comm` of the C ABI: created collectively by all ranks of `group`; the 128-byte unique id travels through the
    torch.distributed group (any backend).  The calling thread's current device must be this rank's GPU."""

    def __init__(self, group=None, device=None, init_timeout_s: Optional[float] = 180.0):
        self._lib = L.load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # i2sdf_comm_init_rank binds the communicator to the calling thread's CURRENT device: pin it to the module's device, so a
        # caller that passes 'cuda:N' everywhere without torch.cuda.set_device does not put every rank's communicator on GPU 0
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        box = [None]
        if self.rank == 0:
            buf = (C.c_ubyte * L.COMM_UNIQUE_ID_BYTES)()
            rc = self._lib.i2sdf_comm_unique_id(buf, L.COMM_UNIQUE_ID_BYTES)
            box[0] = bytes(buf) if rc == 0 else ("error", self._lib.i2sdf_last_comm_error().decode())
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        if not isinstance(box[0], bytes):         # rank 0 could not make the id: EVERY rank stops here, none enters the collective init
            raise L.I2SDFError(f"i2sdf_comm_unique_id failed on rank 0: {box[0]}")
        uid = (C.c_ubyte * L.COMM_UNIQUE_ID_BYTES).from_buffer_copy(box[0])
        h = C.c_void_p()
        # ncclCommInitRank is collective: if a peer never arrives it does not return.  A job that cannot start must FAIL, not hang:
        import os
        import threading

        def _abort():
            print(f"[i2sdf_amd.dist] rank {self.rank} of {self.world} (device {self.device}): i2sdf_comm_init_rank did not return within "
                  f"{init_timeout_s} s (a peer rank is missing or RCCL cannot reach it): aborting the process.  "
                  "attach_data_parallel(init_timeout_s=...) / I2SDF_COMM_INIT_TIMEOUT_S change the limit, 0 or None disable the watchdog", flush=True)
            os._exit(3)

        # the watchdog hard-kills the process (there is no way to cancel a collective init that a peer never joins): the limit is the
        # caller's to choose -- large multi-node jobs may need more than the default, a notebook may prefer no watchdog at all
        timer = None
        if init_timeout_s is not None and init_timeout_s > 0:
            timer = threading.Timer(init_timeout_s, _abort)
            timer.daemon = True
            timer.start()
        try:
            with torch.cuda.device(self.device):
                L.check(self._lib.i2sdf_comm_init_rank(uid, self.world, self.rank, C.byref(h)), "i2sdf_comm_init_rank")
        finally:
            if timer is not None:
                timer.cancel()
        self._h = h
        self._ex = L.Exchange()
        L.check(self._lib.i2sdf_comm_as_exchange(self._h, C.byref(self._ex)), "i2sdf_comm_as_exchange")

    def allreduce_mean(self, flat: torch.Tensor):
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        assert flat.device == self.device, f"communicator lives on {self.device}, gradient buffer on {flat.device}"
        with torch.cuda.device(flat.device):
            L.check(self._lib.i2sdf_allreduce_grads(L.ptr(flat), flat.numel(), self._h, L.stream_ptr()), "i2sdf_allreduce_grads")

    def broadcast(self, t: torch.Tensor, root: int = 0):
        assert t.is_cuda and t.is_contiguous()
        with torch.cuda.device(t.device):
            L.check(self._lib.i2sdf_broadcast(L.ptr(t), t.numel() * t.element_size(), root, self._h, L.stream_ptr()), "i2sdf_broadcast")

    def exchange(self) -> L.Exchange:
        return self._ex

    def nranks(self) -> int:
        """ranks of the RCCL communicator as the LIBRARY reports them (i2sdf_comm_size): what the collectives really span"""
        return int(self._lib.i2sdf_comm_size(self._h)) if getattr(self, "_h", None) else 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.i2sdf_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DevView:
    """A raw device pointer as a torch tensor (torch.as_tensor understands __cuda_array_interface__)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class TorchExchange:
    """The `i2sdf_exchange` hook routed through torch.distributed (any backend): what the gloo tests and ranks that share one GPU
    use for the small exchanges of `equivalent` mode.  Runs on torch's current stream -- the stream the entry points are given."""

    def __init__(self, group=None):
        self.group, self.world = group, dist.get_world_size(group)
        self.calls = 0

        def cb(ctx, buf, n, dtype, op, stream):
            try:
                self.reduce_(torch.as_tensor(_DevView(buf, n, "<i4" if dtype == L.XCHG_I32 else "<f4"), device="cuda"), op)
                return 0
            except Exception:      # never let an exception cross the C boundary
                return -5

        self._cb = L.EXCHANGE_FN(cb)              # keep the trampoline alive as long as the hook is installed
        self._ex = L.Exchange(self._cb, None)

    def reduce_(self, t: torch.Tensor, op: int):
        """the hook's collective on a tensor, in place (MAX / SUM / AVG over the group) -- what the C callback does with the device buffer
        it is handed; bench.py --selftest-launch --equivalent calls it directly on host tensors (no kernels on a CPU box)"""
        if op == L.XCHG_MAX:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            if op == L.XCHG_AVG:
                t.div_(self.world)
        self.calls += 1
        return t

    def exchange(self) -> L.Exchange:
        return self._ex


# ---------------------------------------------------------------------------------------------------------------------
# module-level wiring
# ---------------------------------------------------------------------------------------------------------------------
class DataParallelState:
    def __init__(self, group, world, comm, xchg, equivalent):
        self.group, self.world, self.comm, self.xchg, self.equivalent = group, world, comm, xchg, equivalent
        self.enabled = True


def attach_data_parallel(net, group=None, equivalent: bool = False, native=None, init_timeout_s="env"):
    """Make `net` (i2sdf_amd.I2SDFNetwork) average its flat gradient over the process group inside backward.

    native: use the library's own RCCL communicator for the collectives (default: when the group's backend is nccl); otherwise
    torch.distributed carries them.  Do not ALSO wrap the module in torch DistributedDataParallel / a Lightning DDP strategy:
    the gradients would be reduced twice -- this hook IS the data-parallel strategy of the module.  Under
    gradient accumulation use `with no_sync(net):` for all but the last micro-batch, as with DDP (in `equivalent` mode the micro-batches
    under no_sync() use rank-local loss denominators and a rank-local sampler test: the accumulated step is then NOT the 1-GPU step on
    the concatenated batch -- i2sdf_amd.I2SDFLoss warns once when that happens).

    init_timeout_s: watchdog of the library communicator's collective init (a process that cannot join is killed instead of hanging
    the job): seconds, 0 / None = no watchdog; default: I2SDF_COMM_INIT_TIMEOUT_S from the environment, else 180."""
    if init_timeout_s == "env":
        import os
        init_timeout_s = float(os.environ.get("I2SDF_COMM_INIT_TIMEOUT_S", "180"))
    world = dist.get_world_size(group)
    explicit = native is True
    if native is None:
        # I2SDF_DP_TRANSPORT=torch keeps the gradient mean on torch.distributed's own communicator (also RCCL under backend nccl)
        import os
        native = dist.get_backend(group) == "nccl" and os.environ.get("I2SDF_DP_TRANSPORT", "library") != "torch"
    comm = None
    if native:
        # all ranks must end up on the same transport.  First agree that RCCL can be bound on EVERY rank -- before anyone enters the
        # collective init, where a rank that cannot follow would leave the others waiting -- then create the communicator, then agree
        # on the outcome through the group
        on_gpu = dist.get_backend(group) == "nccl"
        avail = torch.tensor([float(L.load().i2sdf_comm_available())], device="cuda" if on_gpu else "cpu")
        dist.all_reduce(avail, op=dist.ReduceOp.MIN, group=group)
        if float(avail.item()) < 1.0:
            if explicit:
                raise RuntimeError("i2sdf_amd.dist: librccl could not be bound on every rank (i2sdf_comm_available)")
            import warnings
            warnings.warn("i2sdf_amd.dist: librccl not loadable on every rank; the gradient mean goes through torch.distributed.all_reduce")
            native = False
    if native:
        err = None
        try:
            dev = next((p.device for p in net.parameters() if p.is_cuda), None)
            comm = RcclComm(group, device=dev, init_timeout_s=init_timeout_s)
        except Exception as e:       # RCCL not loadable / init failed on this rank
            err = e
            print(f"[i2sdf_amd.dist] rank {dist.get_rank(group)}: library RCCL communicator failed: {e}", flush=True)
        ok = torch.tensor([0.0 if comm is None else 1.0], device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if float(ok.item()) < 1.0:
            if comm is not None:
                comm.close()
                comm = None
            if explicit:
                raise RuntimeError(f"i2sdf_amd.dist: the library's RCCL communicator could not be created on every rank ({err})")
            import warnings
            warnings.warn(f"i2sdf_amd.dist: library RCCL communicator unavailable ({err}); the gradient mean goes through "
                          "torch.distributed.all_reduce (same collective, torch's communicator)")
    xchg = None
    if equivalent:
        xchg = comm if comm is not None else TorchExchange(group)
    state = DataParallelState(group, world, comm, xchg, equivalent)

    def sync(flat_grad: torch.Tensor):
        # always through the collective once attached (a 1-rank group is a no-op for RCCL): the N=1 and N>1 code paths are the same
        if not state.enabled:
            return
        if comm is not None:
            comm.allreduce_mean(flat_grad)
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
            if world > 1:
                flat_grad.mul_(1.0 / world)

    net.grad_sync = sync
    net.dp_state = state
    return net


@contextlib.contextmanager
def no_sync(net):
    """Skip the gradient all-reduce inside the block (gradient accumulation: all micro-batches but the last)."""
    st = getattr(net, "dp_state", None)
    if st is None:
        yield
        return
    old, st.enabled = st.enabled, False
    try:
        yield
    finally:
        st.enabled = old


def attach_loss(loss_fn, net):
    """`equivalent` mode: the loss of this rank uses the global denominators (count-weighted masked means)."""
    st = getattr(net, "dp_state", None)
    loss_fn.exchange = st.xchg.exchange() if (st is not None and st.equivalent) else None
    loss_fn._exchange_owner = st.xchg if st is not None else None
    loss_fn._dp_state = st        # the loss exchanges only in training mode and while st.enabled (not under no_sync())
    return loss_fn


def broadcast_parameters(net, src: int = 0, group=None):
    """Replicate rank `src`'s parameters (one broadcast of the flat buffer when it exists, else per tensor)."""
    flat = getattr(net, "_flat", None)
    if flat is not None:
        dist.broadcast(flat, src=src, group=group)
    else:
        for p in net.parameters():
            dist.broadcast(p.data, src=src, group=group)


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` rays for `rank`: ceil(total/world) per rank."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def shard_pixels(model_input: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Eval layout (uv is (1, P, 2)): keep this rank's pixel slice; pose / intrinsics are shared."""
    P = model_input["uv"].shape[1]
    lo, hi = shard_range(P, rank, world)
    out = dict(model_input)
    out["uv"] = model_input["uv"][:, lo:hi].contiguous()
    return out


def gather_outputs(outputs: Dict[str, torch.Tensor], total: int, group=None) -> Dict[str, torch.Tensor]:
    """Concatenate per-rank render outputs back into image order (all ranks get the full result)."""
    world = dist.get_world_size(group)
    if world == 1:
        return outputs
    per = (total + world - 1) // world
    merged = {}
    for k, v in outputs.items():
        pad = torch.zeros((per,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[: v.shape[0]] = v
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        merged[k] = torch.cat(parts, 0)[:total]
    return merged


def render_image(net, model_input: Dict[str, torch.Tensor], split_n_pixels: int = 12000, group=None) -> Dict[str, torch.Tensor]:
    """Multi-GPU full-image inference (BASELINE config 4): whole chunks of the view are dealt to the ranks in contiguous runs,
    every rank renders its run with ONE library call, and the rows are all-gathered into image order (32 B/ray).  Identical
    to the single-GPU image because the chunk composition does not change."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    P = model_input["uv"].shape[1]
    mine = net.render_image(model_input, split_n_pixels, rank=rank, world_size=world)
    if world == 1:
        return mine
    n_chunks = (P + split_n_pixels - 1) // split_n_pixels
    per = (n_chunks + world - 1) // world * split_n_pixels           # rows per rank (last ranks may hold fewer / none)
    merged = {}
    for k, v in mine.items():
        pad = torch.zeros((per,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[: v.shape[0]] = v
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        merged[k] = torch.cat(parts, 0)[:P]
    return merged


def sdf_volume(net, axes, rot=None, trans=None, order: str = "volume", chunk: int = 1 << 21, group=None) -> torch.Tensor:
    """Multi-GPU SDF volume for marching cubes (row N4): every rank evaluates one contiguous slab of the flat output with ONE
    library call and the slabs are all-gathered (4 B/point).  Points do not interact, so the volume is identical to the 1-GPU one."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = net.sdf_volume(axes, rot, trans, order, chunk, rank=rank, world_size=world)
    if world == 1:
        return mine
    x, y, z = (axes.x, axes.y, axes.z) if hasattr(axes, "shortest_axis_index") else axes
    nx, ny, nz = len(x), len(y), len(z)
    total = nx * ny * nz
    per = (total + world - 1) // world
    pad = torch.zeros(per, dtype=mine.dtype, device=mine.device)
    pad[: mine.shape[0]] = mine
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    flat = torch.cat(parts)[:total]
    return flat.view(nx, ny, nz) if order == "volume" else flat


def global_any(flag: torch.Tensor, group=None) -> torch.Tensor:
    """MAX-reduce a small flag tensor on the host side of the ABI (kept for callers that run their own loop; the module's
    sampler uses the device-side exchange hook instead)."""
    if dist.get_world_size(group) > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return flag
