"""Checkpoint I/O in the reference's `.pyth` format (SURVEY.md 8(f) rank 4;
slowfast/utils/checkpoint.py:37-55,124-156,198-385).

A file written here loads with the reference's `cu.load_checkpoint` and vice versa: a pickled
dict {"epoch", "model_state", "optimizer_state", "cfg", "scaler_state"} whose model_state has the
reference's 405 names in registration order and whose optimizer_state is the
torch.optim.AdamW.state_dict() of the reference's two param groups (svit_amd.optim.FusedClipAdamW
reads and writes that layout from its flat moment buffers).  Layout pinned by
tests/golden/layout.json, recorded from the reference's own save_checkpoint.
"""
import os
from collections import OrderedDict

import torch
import torch.distributed as dist


def get_checkpoint_dir(path_to_job):
    return os.path.join(path_to_job, "checkpoints")


def get_path_to_checkpoint(path_to_job, epoch):
    return os.path.join(get_checkpoint_dir(path_to_job), "checkpoint_epoch_{:05d}.pyth".format(epoch))


def get_last_checkpoint(path_to_job):
    d = get_checkpoint_dir(path_to_job)
    names = [f for f in (os.listdir(d) if os.path.isdir(d) else []) if "checkpoint" in f]
    assert len(names), "No checkpoints found in '{}'.".format(d)
    return os.path.join(d, sorted(names)[-1])


def has_checkpoint(path_to_job):
    d = get_checkpoint_dir(path_to_job)
    return os.path.isdir(d) and any("checkpoint" in f for f in os.listdir(d))


def is_checkpoint_epoch(cfg, cur_epoch, multigrid_schedule=None):
    """checkpoint.py:99-121 without the multigrid branch (not part of the SViT recipe)."""
    if getattr(cfg.TRAIN, "VAL_ONLY", False):
        return False
    if cur_epoch + 1 == cfg.SOLVER.MAX_EPOCH:
        return True
    if multigrid_schedule is not None:
        raise NotImplementedError("multigrid schedules are outside the SViT path")
    return (cur_epoch + 1) % cfg.TRAIN.CHECKPOINT_PERIOD == 0


def _is_master(cfg):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank() % max(1, cfg.NUM_GPUS * getattr(cfg, "NUM_SHARDS", 1)) == 0
    return True


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().to("cpu", copy=True)
    if isinstance(obj, dict):
        return type(obj)((k, _to_cpu(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


# torch.amp.GradScaler().state_dict() of a scaler that has not stepped yet
DEFAULT_SCALER_STATE = {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5,
                        "growth_interval": 2000, "_growth_tracker": 0}


def save_checkpoint(path_to_job, model, optimizer, epoch, cfg, scaler=None):
    """checkpoint.py:124-156.  Tensors are copied off the flat device buffers (independent CPU
    tensors, as the reference writes them)."""
    if not _is_master(cfg):
        return None
    os.makedirs(get_checkpoint_dir(path_to_job), exist_ok=True)
    ms = model.module if hasattr(model, "module") else model
    checkpoint = {
        "epoch": epoch,
        "model_state": OrderedDict((k, _to_cpu(v)) for k, v in ms.state_dict().items()),
        "optimizer_state": _to_cpu(optimizer.state_dict()),
        "cfg": cfg.dump(),
        # the reference's trainer hands its loader an ENABLED GradScaler (TRAIN.MIXED_PRECISION,
        # train_net.py:501-505; checkpoint.py:381-382), whose load_state_dict refuses an empty
        # dict: the bf16 path needs no loss scaling, so write a fresh scaler's state
        "scaler_state": scaler.state_dict() if scaler is not None else dict(DEFAULT_SCALER_STATE),
    }
    path = get_path_to_checkpoint(path_to_job, epoch + 1)
    with open(path, "wb") as f:
        torch.save(checkpoint, f)
    return path


def load_checkpoint(path_to_checkpoint, model, data_parallel=True, optimizer=None, scaler=None,
                    inflation=False, convert_from_caffe2=False, epoch_reset=False,
                    clear_name_pattern=(), replace_name_pattern=(), should_split_qkv=False):
    """checkpoint.py:198-385 for PyTorch checkpoints: optional renaming, then every tensor whose
    name AND shape match the model is loaded (the rest keeps its initialisation), then -- unless
    `epoch_reset` -- the optimizer moments.  Returns the checkpoint's epoch, or -1."""
    assert os.path.exists(path_to_checkpoint), "Checkpoint '{}' not found".format(path_to_checkpoint)
    if inflation or convert_from_caffe2 or should_split_qkv:
        raise NotImplementedError("2-D inflation / caffe2 / split-qkv conversions are outside the SViT path")
    ms = model.module if data_parallel else model
    with open(path_to_checkpoint, "rb") as f:
        checkpoint = torch.load(f, map_location="cpu", weights_only=False)
    pre = checkpoint["model_state"]
    for item in clear_name_pattern:
        pre = OrderedDict((k.replace(item, "") if item in k else k, v) for k, v in pre.items())
    if len(replace_name_pattern) > 0:
        new = {}
        for k, v in pre.items():
            for a, b in replace_name_pattern:
                if a in k:
                    k = k.replace(a, b)
            new[k] = v
        pre = new
    model_dict = ms.state_dict()
    match = {k: v for k, v in pre.items() if k in model_dict and v.size() == model_dict[k].size()}
    not_loaded = [k for k in model_dict.keys() if k not in match]
    ms.load_state_dict(match, strict=False)
    load_checkpoint.not_loaded = not_loaded       # what the reference logs ("... not loaded.")
    epoch = -1
    if "epoch" in checkpoint.keys() and not epoch_reset:
        epoch = checkpoint["epoch"]
        if optimizer:
            optimizer.load_state_dict(checkpoint["optimizer_state"])
        if scaler:
            scaler.load_state_dict(checkpoint["scaler_state"])
    return epoch
