"""`.pyth` compatibility of the HIP model + fused optimizer (svit_amd/checkpoint.py, optim.py)
with what the reference's save_checkpoint wrote (tests/golden/layout.json) and with
torch.optim.AdamW itself (what a released checkpoint's optimizer_state comes from)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import procedural as P
from tests import smoke_impl as S


def _close(got, d, rtol=2e-4):
    g = P.digest(got.detach().cpu())
    assert abs(g["l2"] - d["l2"]) <= rtol * max(d["l2"], 1e-12), (g["l2"], d["l2"])
    assert all(abs(a - b) <= rtol * max(abs(b), 1e-6) + 1e-9 for a, b in zip(g["head"], d["head"]))


def _grads_into(model):
    for n, p in model.named_parameters():
        model.flat.g(n).copy_(1e-3 * P.tensor("layout:grad:" + n, tuple(p.shape)))


def test_checkpoint_file_equals_the_references(tmp_path, golden_dir):
    from svit_amd import checkpoint, optim
    layout = json.load(open(os.path.join(golden_dir, "layout.json")))
    cfg, model, spec, sd = S.build_hip_model(16, 224)
    opt = optim.FusedClipAdamW(model, lr=cfg.SOLVER.BASE_LR, weight_decay=cfg.SOLVER.WEIGHT_DECAY,
                               clip_grad_l2norm=None)            # the fixture's step is unclipped
    assert opt.state_dict()["state"] == {}
    _grads_into(model)
    opt.step()
    torch.cuda.synchronize()
    path = checkpoint.save_checkpoint(str(tmp_path), model, opt, 4, cfg)
    assert os.path.basename(path) == layout["file_name"]
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert sorted(ck) == layout["top_level_keys"] and ck["epoch"] == layout["epoch"]
    assert type(ck["cfg"]).__name__ == layout["cfg_type"]
    assert list(ck["model_state"]) == layout["model_state_keys"]
    assert all(not v.is_cuda and v.is_contiguous() and v.untyped_storage().nbytes() == v.numel() * 4
               for v in ck["model_state"].values())            # independent tensors, not flat views
    osd = ck["optimizer_state"]
    assert len(osd["param_groups"]) == len(layout["param_groups"])
    for g, want in zip(osd["param_groups"], layout["param_groups"]):
        assert g["params"] == want["params"] and set(g) == set(want)
        for k in ("lr", "eps", "weight_decay", "amsgrad"):
            assert g[k] == want[k], k
        assert list(g["betas"]) == list(want["betas"])
    order = layout["optimizer_order"]
    assert sorted(osd["state"]) == list(range(len(order)))
    for j, n in enumerate(order):
        ent = osd["state"][j]
        assert sorted(ent) == layout["state_entry_keys"]
        assert tuple(ent["exp_avg"].shape) == tuple(layout["shapes"][n]) == tuple(ent["exp_avg_sq"].shape)
        assert ent["step"].dtype == torch.float32 and float(ent["step"]) == layout["state_step"]["value"]
    # numbers: parameters after the AdamW step and both moments, vs the reference's own step
    for n, d in layout["digests"].items():
        assert order[d["index"]] == n
        _close(ck["model_state"][n], d["param"])
        _close(osd["state"][d["index"]]["exp_avg"], d["exp_avg"])
        _close(osd["state"][d["index"]]["exp_avg_sq"], d["exp_avg_sq"], rtol=5e-4)
    # round trip into a fresh model + optimizer
    cfg2, model2, _, _ = S.build_hip_model(16, 224)
    model2.flat.data.zero_()
    opt2 = optim.construct_optimizer(model2, cfg2)
    assert checkpoint.load_checkpoint(path, model2, data_parallel=False, optimizer=opt2) == 4
    assert checkpoint.load_checkpoint.not_loaded == []
    assert torch.equal(model2.flat.data, model.flat.data)
    assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    assert opt2.step_count == 1


def test_resume_from_a_torch_adamw_checkpoint(tmp_path, golden_dir):
    """A checkpoint as the reference produces it -- torch.optim.AdamW over the reference's two
    groups -- resumed by the fused optimizer: the next step equals torch's next step."""
    from svit_amd import checkpoint, optim
    layout = json.load(open(os.path.join(golden_dir, "layout.json")))
    cfg, model, spec, sd = S.build_hip_model(4, 64)
    names = [n for n, _ in model.named_parameters()]
    cpu = {n: torch.nn.Parameter(sd[n].clone()) for n in names}
    dec = [n for n in names if not (cpu[n].dim() == 1 or n.endswith(".bias"))]
    rest = [n for n in names if n not in set(dec)]
    assert len(dec) == layout["param_groups"][0]["params"][-1] + 1
    ref = torch.optim.AdamW([{"params": [cpu[n] for n in dec], "weight_decay": 1e-4},
                             {"params": [cpu[n] for n in rest], "weight_decay": 0.0}],
                            lr=2e-4, eps=1e-8, weight_decay=1e-4)

    def grads(tag):
        return {n: 1e-3 * P.tensor("resume:%s:%s" % (tag, n), tuple(cpu[n].shape)) for n in names}
    for tag in ("a", "b"):
        for n, g in grads(tag).items():
            cpu[n].grad = g
        ref.step()
    path = str(tmp_path / "checkpoint_epoch_00002.pyth")
    torch.save({"epoch": 1, "model_state": {n: cpu[n].detach().clone() for n in names},
                "optimizer_state": ref.state_dict(), "cfg": cfg.dump(), "scaler_state": {}}, path)
    opt = optim.FusedClipAdamW(model, lr=1.0, weight_decay=1e-4, clip_grad_l2norm=None)
    assert checkpoint.load_checkpoint(path, model, data_parallel=False, optimizer=opt) == 1
    assert opt.step_count == 2 and opt.param_groups[0]["lr"] == 2e-4
    g3 = grads("c")
    for n in names:
        cpu[n].grad = g3[n]
        model.flat.g(n).copy_(g3[n])
    ref.step()
    opt.step()
    torch.cuda.synchronize()
    for n in names:
        d = float((model.flat.p(n).cpu() - cpu[n].detach()).abs().max())
        assert d <= 2e-6, (n, d)
    # a checkpoint of another clip length: temporal tables differ in shape and are skipped
    cfg8, model8, _, _ = S.build_hip_model(8, 64)
    assert checkpoint.load_checkpoint(path, model8, data_parallel=False, epoch_reset=True) == -1
    skipped = set(checkpoint.load_checkpoint.not_loaded)
    assert "pos_embed_temporal" in skipped and all("rel_pos_t" in k or k == "pos_embed_temporal" for k in skipped)
    saved = torch.load(path, map_location="cpu", weights_only=False)["model_state"]
    assert torch.equal(model8.flat.p("blocks.3.mlp.fc1.weight").cpu(), saved["blocks.3.mlp.fc1.weight"])
