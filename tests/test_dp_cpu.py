"""Data-parallel exchange (svit_amd/dp.py) on CPU: gloo, world size 2.  The wrapper's logic --
readiness-ordered flat slices, bucketed async all-reduce, mean semantics, rank-0 broadcast -- is
device independent, so it is exercised here on the real SViT parameter table with CPU buffers."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubModel(torch.nn.Module):
    """Carries what DataParallel touches on an SViT: .flat, .cls_token, ._grad_ready_hook."""

    def __init__(self, flat):
        super().__init__()
        self.flat = flat
        self.cls_token = torch.nn.Parameter(flat.p("cls_token"))
        self._grad_ready_hook = None

    def forward(self, x):
        return x


def _worker(rank, world, port, ok):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from svit_amd import arch, config
    from svit_amd.dp import DataParallel
    from svit_amd.engine import FlatParams
    from svit_amd.model import _weight_decayed
    torch.set_num_threads(2)
    cfg = config.ssv2_cfg(16, 224)
    plan = arch.build_plan(cfg)
    shapes = arch.param_shapes(plan)
    flat = FlatParams(shapes, _weight_decayed, torch.device("cpu"), arch.readiness_rank)
    # every element of the flat buffer belongs to exactly one readiness slice
    cover = torch.zeros(flat.total, dtype=torch.int32)
    for rr in flat.ready_ranges:
        for a, b in rr:
            cover[a:b] += 1
    assert int(cover.min()) == 1 and int(cover.max()) == 1
    assert flat.n_ranks == 18 and flat.total >= 34373560
    torch.manual_seed(rank)
    flat.data.copy_(torch.randn(flat.total))
    model = _StubModel(flat)
    dp = DataParallel(model, bucket_ranks=4)
    # construction broadcasts rank 0's weights
    ref = torch.randn(flat.total, generator=torch.Generator().manual_seed(0))
    assert torch.equal(flat.data, ref)
    assert dp.module is model and dp.device == torch.device("cpu")
    # a backward: each rank holds different grads; the engine reports ranks 0..17 in order
    base = torch.arange(flat.total, dtype=torch.float32) % 97
    flat.grad.copy_(base * (rank + 1))
    for r in range(flat.n_ranks):
        dp._on_ready(r)
    expect = base * sum(range(1, world + 1)) / world
    assert torch.allclose(flat.grad, expect), float((flat.grad - expect).abs().max())
    assert dp._works == [] and dp._pending == []
    # slices of a readiness rank are final before later ranks are reported: launching the
    # collective for rank r must not touch rank r+1's slice
    flat.grad.zero_()
    a, b = flat.ready_ranges[0][0]
    flat.grad[a:b] = float(rank + 1)
    for r in range(4):       # first bucket only
        dp._on_ready(r)
    dp.finish()
    assert torch.allclose(flat.grad[a:b], torch.full((b - a,), (1 + world) / 2.0))
    ok[rank] = 1
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 8])
def test_flat_allreduce_gloo(world):
    """world 2: the N > 1 path; world 8: the rank / port / bucket bookkeeping at the world size the
    driver's scaling run uses (SCALE_rNN.json: --gpus 8), on the real 34.4 M-element parameter table."""
    port = _free_port()
    # spawn, not fork: earlier tests of the same pytest process may have started OpenMP threads
    # (any multi-threaded torch op), and a forked child then deadlocks in its first parallel region
    ctx = mp.get_context("spawn")
    ok = ctx.Array("i", [0] * world)
    procs = [ctx.Process(target=_worker, args=(r, world, port, ok)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert list(ok) == [1] * world


def test_bucket_message_sizes_of_the_real_layout():
    """What the exchange looks like on the wire at 16x224^2: with bucket_ranks = 4
    the 137.5 MB fp32 gradient leaves in 5 launches -- sizes and order pinned here so that the first
    real SCALE line can be read against them."""
    sys.path.insert(0, ROOT)
    from svit_amd import arch, config
    from svit_amd.engine import FlatParams
    from svit_amd.model import _weight_decayed
    cfg = config.ssv2_cfg(16, 224)
    shapes = arch.param_shapes(arch.build_plan(cfg))
    flat = FlatParams(shapes, _weight_decayed, torch.device("cpu"), arch.readiness_rank)
    launches, pending = [], []
    for r in range(flat.n_ranks):
        pending.extend(flat.ready_ranges[r])
        if (r + 1) % 4 == 0 or r == flat.n_ranks - 1:
            merged = []
            for a, b in sorted(pending):
                if merged and merged[-1][1] == a:
                    merged[-1] = (merged[-1][0], b)
                else:
                    merged.append((a, b))
            launches.append([4 * (b - a) for a, b in merged])
            pending = []
    total = sum(sum(l) for l in launches)
    assert total == 4 * flat.total and abs(total - 137.5e6) < 1.0e6
    assert len(launches) == 5
    # every bucket is at most two contiguous messages (the decayed and the not-decayed group)
    assert all(len(l) <= 2 for l in launches), launches
    # the first bucket (head, final norm, blocks 15-13) carries ~40 % of the bytes: the 768-wide
    # blocks; the last one (blocks 1-0, stem, tokens) less than 1 % -- the un-overlappable tail
    first, last = sum(launches[0]), sum(launches[-1])
    assert 0.30 * total < first < 0.50 * total, first / total
    assert last < 0.01 * total, last / total
