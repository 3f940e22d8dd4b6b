"""Host transport for the CPU tests of mpc4quantum_amd.distributed: torch.distributed "gloo" stands in for RCCL so that
the partition / layout / padding / status-word / unpack code of the product runs with world_size 2 on a box without GPUs.
Test infrastructure only: the product never imports torch."""
import numpy as np


class GlooTransport:
    on_device = False

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def gather_host(self, buf, dst):
        """buf: uint8 ndarray -> list of per-rank uint8 ndarrays on dst, None elsewhere."""
        t = self.torch.from_numpy(buf)
        outs = [self.torch.empty_like(t) for _ in range(self.world)] if self.rank == dst else None
        self.dist.gather(t, outs, dst=dst, group=self.group)
        return [o.numpy() for o in outs] if outs is not None else None

    def wait(self, slot=-1):
        pass
