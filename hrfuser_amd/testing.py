"""Harness that runs a SUB-BLOCK of the backbone (Bottleneck, HRFormerBlock, fusion block, HR
module ...) on the HIP engine with NCHW torch tensors in/out, so block-level parity tests can
compare each piece against the oracle exactly like the reference's own module boundaries."""
import torch

from . import runtime as R
from .backbone import HipModule


class BlockHarness(HipModule):
    """`runner(ctx, block, acts) -> list[Act]`; inputs/outputs are logical NCHW tensors."""

    def __init__(self, block, runner):
        super().__init__()
        self.block = block
        self._runner = runner

    def forward(self, *inputs):
        return self._call_engine(tuple(inputs))

    def _wrap_inputs(self, inputs):
        return [R.Act(t.permute(0, 2, 3, 1).contiguous(), bool(t.requires_grad)) for t in inputs]

    def _run(self, ctx, srcs):
        outs = self._runner(ctx, self.block, srcs)
        return list(outs) if isinstance(outs, (list, tuple)) else [outs]
