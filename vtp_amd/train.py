"""Data-parallel training step for the VTP hot path on MI355X.

The reference ships the model side of training only (VTP.forward(forward_type=...), vtp/models/vtp.py:323-338): no
loss, optimizer or DDP wrapper.  This module is that missing driver, MI355X-first:

  * one process per GPU, full replica, minibatch sharded by rank (the reference's only strategy is DP);
  * forward/backward are hand-scheduled kernel sequences (vtp_amd.engine), gradients land in ONE flat fp32 buffer;
  * gradient all-reduce = a few large contiguous RCCL buckets launched *during* backward (the c10d NCCL/RCCL backend
    runs them on its own HIP stream; they overlap the remaining backward kernels) and waited on before the optimizer;
    xGMI is point-to-point, so few large messages beat many small ones;
  * one fused AdamW launch per contiguous parameter range + one batched bf16-weight refresh launch.

Loss heads are OUR spec (parity unpinned by the reference): rec = mean |decode(encode(x)) - x| (L1).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import ops
from .engine import BF, F32


class GradBucketer:
    """Bucketed asynchronous all-reduce over contiguous ranges of a flat gradient buffer.

    Pure torch.distributed (works with gloo on CPU for tests, nccl==RCCL on MI355X)."""

    def __init__(self, flat_g: torch.Tensor, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.flat_g = flat_g
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.works = []
        self.reduced_elems = 0

    def reduce_range(self, lo: int, hi: int):
        """Launch (async) sum-all-reduce of flat_g[lo:hi]."""
        if self.world == 1 or hi <= lo:
            return
        w = self.dist.all_reduce(self.flat_g[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.works.append(w)
        self.reduced_elems += hi - lo

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []


def merge_ranges(ranges: Sequence[Tuple[int, int]]) -> List[Tuple[int, int]]:
    out: List[Tuple[int, int]] = []
    for lo, hi in sorted(ranges):
        if out and lo <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], hi))
        else:
            out.append((lo, hi))
    return out


def param_ranges(offsets, prefixes: Sequence[str]) -> List[Tuple[int, int]]:
    """Contiguous [lo,hi) ranges (4-element aligned) of the flat buffers covering every parameter whose name starts
    with one of `prefixes`."""
    r = []
    for name, (o, k) in offsets.items():
        if any(name.startswith(p) for p in prefixes):
            r.append((o, o + (k + 3) // 4 * 4))
    return merge_ranges(r)


class VTPTrainer:
    """Reconstruction-path trainer (BASELINE config 2 and the `rec` third of config 3)."""

    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.05,
                 group=None, bucket_blocks: int = 3, use_graphs: bool = False):
        self.model = model
        self.store = model._engine()
        self.trunk, self.decoder = model._trunk, model._decoder
        if self.decoder is None:
            raise RuntimeError("VTPTrainer needs train_reconstruction=True")
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        st = self.store
        self.ranges = param_ranges(st.offsets, ("trunk.", "pixel_decoder."))
        # visual_proj sits between trunk.* and pixel_decoder.* in registration order; it takes no gradient on the rec path
        self.m = torch.zeros_like(st.flat_p)
        self.v = torch.zeros_like(st.flat_p)
        self.step_no = 0
        self.loss_sum = torch.zeros(1, dtype=F32, device=st.device)
        self.bucketer = GradBucketer(st.flat_g, group)
        self.world = self.bucketer.world
        self.bucket_blocks = bucket_blocks
        self._bucket_plan = self._plan_buckets()
        # hyper-parameters live in device memory so that captured hipGraphs replay with per-step values
        self.hyper = torch.zeros(8, dtype=F32, device=st.device)
        self._hyper_host = torch.zeros(8, dtype=F32).pin_memory()
        self.use_graphs = use_graphs
        self._graphs = {}

    # gradient buckets in backward-completion order: decoder tail, decoder blocks (high->low), trunk blocks, trunk head
    def _plan_buckets(self):
        off = self.store.offsets

        def rng(prefix):
            return param_ranges(off, (prefix,))

        plan = {"dec_tail": rng("pixel_decoder.norm.") + rng("pixel_decoder.proj_out."),
                "dec_head": rng("pixel_decoder.proj_in."),
                "trunk_tail": rng("trunk.norm.") + rng("trunk.feature_bottleneck."),
                "trunk_head": rng("trunk.cls_token") + rng("trunk.mask_token") + rng("trunk.patch_embed.")}
        for key, tower, depth in (("dec", "pixel_decoder", self.decoder.depth), ("trunk", "trunk", self.trunk.depth)):
            for i in range(depth):
                plan[f"{key}.{i}"] = rng(f"{tower}.blocks.{i}.")
        return plan

    def _reduce(self, keys: Sequence[str]):
        if self.world == 1 or not keys:
            return
        rs = merge_ranges([r for k in keys for r in self._bucket_plan[k]])
        for lo, hi in rs:
            self.bucketer.reduce_range(lo, hi)

    # ------------------------------------------------------------------------------------------------------------
    # The step is written ONCE as a generator.  Every `yield keys` is a point where the gradients of the parameter
    # groups `keys` are complete: the eager driver launches their RCCL all-reduce there; the graph driver ends a
    # hipGraph segment there (collectives stay outside the graphs, between segment launches).
    # ------------------------------------------------------------------------------------------------------------
    def _tower_backward(self, tower: str, gen, depth: int):
        nb = self.bucket_blocks
        result = None
        try:
            while True:
                ev = next(gen)
                if ev == "tail":
                    yield [f"{tower}_tail"]
                elif ev[0] == "block" and ev[1] % nb == 0:
                    i = ev[1]
                    yield [f"{tower}.{j}" for j in range(i, min(i + nb, depth))]
        except StopIteration as stop:
            result = stop.value
        return result

    def _step_gen(self, images: torch.Tensor):
        st = self.store
        B, _, H, W = images.shape
        h, w = H // 16, W // 16
        st.zero_grad()
        self.loss_sum.zero_()
        self.trunk.forward(images, train=True)
        lat = self.trunk.latents()
        t = self.decoder.forward(lat, B, h, w, train=True)
        dt = self.decoder._ctx[0].get("b.dt", (B * h * w, 768), BF)
        ops.l1_loss_fwd_bwd(t, images, dt, self.loss_sum, B, h, w, 1.0 / (B * 3 * H * W))
        d_lat = yield from self._tower_backward("dec", self.decoder.backward(dt), self.decoder.depth)
        yield ["dec_head"]
        yield from self._tower_backward("trunk", self.trunk.backward(d_lat), self.trunk.depth)
        yield ["trunk_head"]
        # ---- optimizer (after every bucket has been reduced)
        for lo, hi in self.ranges:
            ops.adamw_dev(st.flat_p[lo:hi], st.flat_g[lo:hi], self.m[lo:hi], self.v[lo:hi], None, hi - lo, self.hyper)
        st.prep()

    def _set_hyper(self):
        self.step_no += 1
        b1, b2 = self.betas
        vals = [self.lr, b1, b2, self.eps, self.wd, 1.0 - b1 ** self.step_no, (1.0 - b2 ** self.step_no) ** 0.5,
                1.0 / self.world]
        self._hyper_host.copy_(torch.tensor(vals, dtype=torch.float32))
        self.hyper.copy_(self._hyper_host, non_blocking=True)

    def step_rec(self, images: torch.Tensor) -> torch.Tensor:
        """One optimizer step on the L1 reconstruction loss.  images: f32 [B,3,H,W] on the device.
        Returns the (local) loss as a device scalar tensor (no host sync)."""
        B, _, H, W = images.shape
        self._set_hyper()
        if self.use_graphs:
            self._step_graphs(images)
        else:
            gen = self._step_gen(images)
            pending_wait = False
            for keys in gen:
                if keys == ["trunk_head"]:
                    self._reduce(keys)
                    self.bucketer.wait()  # the generator's next (last) leg is the optimizer
                    pending_wait = True
                else:
                    self._reduce(keys)
            assert pending_wait
        self.model._pver = self.model._param_version()
        return self.loss_sum / float(B * 3 * H * W)

    # ---- hipGraph path: one captured graph per segment, replayed every step; collectives between segments ----------
    def _step_graphs(self, images: torch.Tensor):
        key = tuple(images.shape)
        plan = self._graphs.get(key)
        if plan is None:
            # warm-up in eager mode on a side stream (allocates every workspace buffer, sets kernel attributes)
            static_img = torch.empty_like(images)
            static_img.copy_(images)
            snap = (self.store.flat_p.clone(), self.m.clone(), self.v.clone())
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in self._step_gen(static_img):
                    pass
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            # the warm-up performed a real (local, un-reduced) optimizer step: roll parameters and moments back
            st = self.store
            st.flat_p.copy_(snap[0])
            self.m.copy_(snap[1])
            self.v.copy_(snap[2])
            st.prep()
            del snap
            segs = []
            pool = torch.cuda.graph_pool_handle()
            gen = self._step_gen(static_img)
            done = False
            while not done:
                g = torch.cuda.CUDAGraph()
                keys = None
                with torch.cuda.graph(g, pool=pool):
                    try:
                        keys = next(gen)
                    except StopIteration:
                        done = True
                segs.append((g, keys))
            plan = (static_img, segs)
            self._graphs[key] = plan
        static_img, segs = plan
        static_img.copy_(images)
        for g, keys in segs:
            g.replay()
            if keys is not None:
                self._reduce(keys)
                if keys == ["trunk_head"]:
                    self.bucketer.wait()
