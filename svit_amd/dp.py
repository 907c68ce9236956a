"""Data parallelism for the SViT hot path: one process per GPU, weights replicated, the global
batch sharded, ONE exchange per step -- the gradient all-reduce.

Replaces torch DistributedDataParallel (slowfast/models/build.py:67-74).  The engine
accumulates gradients into a flat buffer laid out in readiness order, so the all-reduce is a
handful of large contiguous slices launched from the backward schedule itself (after block 15,
14, ... 0) on RCCL's stream while the remaining blocks are still back-propagating.  xGMI is
point-to-point: few, large messages are the right shape for it (no 25 MB bucket heuristics).
"""
import collections

import torch
import torch.distributed as dist
import torch.nn as nn

RankRole = collections.namedtuple("RankRole", "is_image data_rank replicas batch_size dataset")


def rank_role(cfg, local_rank):
    """Which data a rank trains on in the published recipe (slowfast/datasets/loader.py:186-201):
    GPUs listed in cfg.IMAGE_TRAIN.GPU_IDS train still images with the HAOG losses, the others
    train clips with CE; each group shards ITS global batch over its own members.  The gradient
    exchange is the same flat all-reduce over all ranks -- every `param.grad` is a view of the
    flat buffer that `zero_grad` clears, so heads a rank's loss does not reach contribute exact
    zeros (what the reference obtains with its `0 * sum(params)` touches,
    video_model_builder.py:359,514)."""
    img = sorted(cfg.IMAGE_TRAIN.GPU_IDS)
    vid = [i for i in range(cfg.NUM_GPUS) if i not in img]
    is_image = local_rank in img
    group = img if is_image else vid
    bs = cfg.IMAGE_TRAIN.BATCH_SIZE if is_image else cfg.TRAIN.BATCH_SIZE
    return RankRole(is_image, group.index(local_rank), len(group), int(bs / max(1, len(group))),
                    "multi_images" if is_image else cfg.TRAIN.DATASET)


class DataParallel(nn.Module):
    """Exposes the surface callers of the reference touch on a DDP-wrapped model: `.module`,
    `.device`, `train()/eval()`, `parameters()` (checkpoint.py:140,236; train_net.py:117-118)."""

    def __init__(self, module, device_ids=None, output_device=None, find_unused_parameters=False,
                 process_group=None, bucket_ranks=4, average=True, force_collectives=False):
        super().__init__()
        self.module = module
        self.device = module.cls_token.device
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_ranks = max(1, bucket_ranks)
        self.average = average
        # test / rehearsal switch: run the whole exchange (broadcast, bucketed async all-reduce, finish) even when the
        # process group has ONE rank -- the production branch (backend "nccl" = RCCL, ReduceOp.AVG) on a one-GPU box
        self.force_collectives = bool(force_collectives) and dist.is_initialized()
        self._works = []
        self._pending = []
        module._grad_ready_hook = self._on_ready
        module._grad_ready_ranks = self.launch_ranks()
        if self.world_size > 1 or self.force_collectives:
            self._broadcast_parameters()

    def _broadcast_parameters(self):
        """rank 0's weights everywhere (DDP does the same at construction)."""
        flat = self.module.flat
        dist.broadcast(flat.data, src=0, group=self.process_group)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    # ------------------------------------------------------------------ gradient exchange ---
    def _reduce_slice(self, a, b):
        g = self.module.flat.grad[a:b]
        backend = dist.get_backend(self.process_group)
        if self.average and backend == "nccl":
            op = dist.ReduceOp.AVG
            self._works.append((dist.all_reduce(g, op=op, group=self.process_group, async_op=True), None))
        else:
            w = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
            self._works.append((w, g if self.average else None))

    def launch_ranks(self):
        """readiness ranks after which `_on_ready` launches collectives (graph.py cuts the
        backward graph there)."""
        n = self.module.flat.n_ranks
        return {r for r in range(n) if (r + 1) % self.bucket_ranks == 0 or r == n - 1}

    def _on_ready(self, rank):
        """engine callback: gradients of readiness rank `rank` are final on this GPU."""
        if self.world_size == 1 and not self.force_collectives:
            return
        flat = self.module.flat
        self._pending.extend(flat.ready_ranges[rank])
        last = rank == flat.n_ranks - 1
        if (rank + 1) % self.bucket_ranks == 0 or last:
            # merge adjacent slices, then launch one collective per contiguous slice
            merged = []
            for a, b in sorted(self._pending):
                if merged and merged[-1][1] == a:
                    merged[-1] = (merged[-1][0], b)
                else:
                    merged.append((a, b))
            self._pending = []
            for a, b in merged:
                self._reduce_slice(a, b)
        if last:
            self.finish()

    def finish(self):
        """Make the current stream wait for every outstanding all-reduce."""
        for w, g in self._works:
            w.wait()
            if g is not None:
                g.div_(self.world_size)
        self._works = []
