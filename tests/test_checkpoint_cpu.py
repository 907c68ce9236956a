"""Host logic of svit_amd/checkpoint.py (slowfast/utils/checkpoint.py:37-55,99-121,198-385) on a
stand-in module, and the layout fixture recorded from the reference (tests/golden/layout.json)."""
import json
import os

import pytest
import torch

from svit_amd import arch, checkpoint, config


class _Tiny(torch.nn.Module):
    def __init__(self, width=4):
        super().__init__()
        self.a = torch.nn.Linear(3, width)
        self.b = torch.nn.Linear(width, 2)


def _cfg():
    cfg = config.ssv2_cfg(4, 64)
    cfg.NUM_GPUS = 1
    return cfg


def test_paths_and_epoch_rule(tmp_path):
    job = str(tmp_path)
    assert checkpoint.get_path_to_checkpoint(job, 5).endswith("checkpoints/checkpoint_epoch_00005.pyth")
    assert not checkpoint.has_checkpoint(job)
    cfg = _cfg()
    cfg.TRAIN.CHECKPOINT_PERIOD, cfg.SOLVER.MAX_EPOCH = 5, 50
    assert [e for e in range(50) if checkpoint.is_checkpoint_epoch(cfg, e)] == [4, 9, 14, 19, 24, 29, 34, 39, 44, 49]
    cfg.SOLVER.MAX_EPOCH = 48
    assert checkpoint.is_checkpoint_epoch(cfg, 47)


def test_save_then_load_with_shape_matching(tmp_path):
    cfg = _cfg()
    src, opt = _Tiny(4), None
    opt = torch.optim.AdamW(src.parameters(), lr=1e-3)
    src(torch.randn(2, 3)).sum().backward() if False else None
    for p in src.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    path = checkpoint.save_checkpoint(str(tmp_path), src, opt, 2, cfg)
    assert os.path.basename(path) == "checkpoint_epoch_00003.pyth"
    assert checkpoint.has_checkpoint(str(tmp_path)) and checkpoint.get_last_checkpoint(str(tmp_path)) == path
    ck = torch.load(path, weights_only=False)
    assert sorted(ck) == ["cfg", "epoch", "model_state", "optimizer_state", "scaler_state"]
    assert isinstance(ck["cfg"], str) and ck["epoch"] == 2
    # the reference's trainer resumes through an ENABLED GradScaler (train_net.py:501-505,
    # checkpoint.py:381-382); an empty state would make its load_state_dict raise
    scaler = torch.amp.GradScaler("cpu", enabled=True)
    scaler.load_state_dict(ck["scaler_state"])
    assert scaler.get_scale() == 65536.0 and scaler.get_growth_interval() == 2000
    assert ck["scaler_state"] == torch.amp.GradScaler("cpu", enabled=True).state_dict()
    path2 = checkpoint.save_checkpoint(str(tmp_path / "s"), src, opt, 2, cfg, scaler=scaler)
    assert torch.load(path2, weights_only=False)["scaler_state"] == scaler.state_dict()
    # same architecture: everything loads, optimizer moments restored, epoch returned
    dst = _Tiny(4)
    opt2 = torch.optim.AdamW(dst.parameters(), lr=1e-3)
    assert checkpoint.load_checkpoint(path, dst, data_parallel=False, optimizer=opt2) == 2
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))
    assert torch.equal(opt2.state_dict()["state"][0]["exp_avg"], opt.state_dict()["state"][0]["exp_avg"])
    # a wider model: tensors whose shape differs keep their init, epoch_reset -> -1
    wide = _Tiny(6)
    before = {k: v.clone() for k, v in wide.state_dict().items()}
    assert checkpoint.load_checkpoint(path, wide, data_parallel=False, epoch_reset=True) == -1
    assert set(checkpoint.load_checkpoint.not_loaded) == {"a.weight", "a.bias", "b.weight"}
    assert torch.equal(wide.b.bias, src.b.bias) and torch.equal(wide.a.weight, before["a.weight"])
    # renaming patterns
    ck["model_state"] = {"module." + k: v for k, v in ck["model_state"].items()}
    torch.save(ck, path)
    dst2 = _Tiny(4)
    checkpoint.load_checkpoint(path, dst2, data_parallel=False, clear_name_pattern=("module.",))
    assert torch.equal(dst2.a.weight, src.a.weight)
    with pytest.raises(NotImplementedError):
        checkpoint.load_checkpoint(path, dst2, data_parallel=False, inflation=True)
    with pytest.raises(AssertionError):
        checkpoint.load_checkpoint(path + ".missing", dst2, data_parallel=False)


def test_parameter_order_is_the_references(golden_dir):
    """arch.param_shapes = named_parameters() order of the reference model (state index of the
    optimizer moments in a released .pyth depends on it)."""
    layout = json.load(open(os.path.join(golden_dir, "layout.json")))
    shapes = arch.param_shapes(arch.build_plan(config.ssv2_cfg(16, 224)))
    assert list(shapes) == layout["named_parameters"] == layout["model_state_keys"]
    assert {k: list(v) for k, v in shapes.items()} == layout["shapes"]
    from oracle import svit_ref as R
    assert list(R.param_shapes(R.make_spec(16, 224))) == layout["named_parameters"]


def test_zero_decay_pos_cls_follows_the_reference_rule():
    """optimizer.py:35-52: `name in skip` is an exact match on the dotted name, so of
    no_weight_decay()'s names only cls_token / object_queries / pos_embed_temporal ever leave the
    decayed group (single GPU); behind the DDP wrapper (NUM_GPUS > 1) the reference finds no
    `no_weight_decay` attribute and skips nothing."""
    from svit_amd.model import SViT
    cfg = config.ssv2_cfg(4, 64)
    cfg.MVIT.ZERO_DECAY_POS_CLS = True
    m = SViT(cfg)
    shapes = {n: tuple(p.shape) for n, p in m.named_parameters()}
    zero = {n for n, s in shapes.items() if not m.weight_decayed(n, s)}
    plain = {n for n, s in shapes.items() if len(s) == 1 or n.endswith(".bias")}
    assert zero == plain | {"cls_token", "object_queries", "pos_embed_temporal"}
    assert "blocks.0.attn.rel_pos_h" not in zero
    cfg.NUM_GPUS = 8
    assert {n for n, s in shapes.items() if not m.weight_decayed(n, s)} == plain
    cfg.NUM_GPUS, cfg.MVIT.ZERO_DECAY_POS_CLS = 1, False
    assert {n for n, s in shapes.items() if not m.weight_decayed(n, s)} == plain
