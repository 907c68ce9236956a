"""Multi-view test ensemble on the device (svit_amd/evaluate.py) against the
reference's own TestMeter / topks_correct results (tests/golden/meter.npz) and the CPU oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import meter_ref
from oracle import procedural as P
from oracle import svit_ref as R
from tests import smoke_impl as S


def _stream(golden_dir, manifest):
    g = np.load(os.path.join(golden_dir, "meter.npz"))
    m = manifest["meter"]
    return g, m, m["ensemble_views"] * m["spatial_crops"]


@pytest.mark.parametrize("order", ["ordered", "shuffled"])
@pytest.mark.parametrize("method", ["sum", "max"])
def test_device_meter_bit_exact_vs_reference(order, method, golden_dir, manifest):
    from svit_amd.evaluate import TestMeter
    g, m, num_clips = _stream(golden_dir, manifest)
    probs, labels_v, perm = torch.from_numpy(g["probs"]), torch.from_numpy(g["labels_v"]), g["perm_" + order]
    meter = TestMeter(m["videos"], num_clips, m["classes"], 1, ensemble_method=method)
    for a in range(0, len(perm), m["batch"]):
        ids = torch.from_numpy(perm[a:a + m["batch"]])
        meter.update_stats(probs[ids].cuda(), labels_v[ids // num_clips].cuda(), ids.cuda())
    key = "%s_%s" % (order, method)
    assert np.array_equal(meter.video_preds.cpu().numpy(), g[key + "_video_preds"])     # bit-exact
    assert np.array_equal(meter.clip_count.cpu().numpy(), g[key + "_clip_count"])
    assert np.array_equal(meter.video_labels.cpu().numpy(), g[key + "_video_labels"])
    assert meter.topks_correct((1, 5)) == list(g[key + "_topk_correct"])
    stats = meter.finalize_metrics((1, 5))
    want = m["results"][key]
    assert stats["top1_acc"] == want["top1_acc"] and stats["top5_acc"] == want["top5_acc"]
    meter.reset()
    assert float(meter.video_preds.abs().max()) == 0 and int(meter.clip_count.sum()) == 0


def test_deduplicated_views_fold_like_the_full_stream():
    """30 listed clips per video = 3 unique crops x 10 identical views: folding the 3 unique ones
    cyclically 10 times must equal the reference's 30 sequential additions bit for bit."""
    from svit_amd.evaluate import TestMeter
    V, crops, views, C = 5, 3, 10, 174
    g = torch.Generator().manual_seed(2)
    uniq = torch.softmax(torch.randn(V * crops, C, generator=g), dim=1)
    labels_v = torch.randint(0, C, (V,), generator=g)
    ref = meter_ref.TestMeterRef(V, crops * views, C)
    table = meter_ref.test_views(V, views, crops)
    ids = np.arange(len(table))
    full = torch.stack([uniq[v * crops + s] for v, s in table])
    for a in range(0, len(ids), 8):                                     # the reference's stream
        sl = ids[a:a + 8]
        ref.update_stats(full[sl].numpy(), labels_v[sl // (crops * views)].numpy(), sl)
    meter = TestMeter(V, crops, C, 1)
    uid = torch.arange(V * crops)
    for a in range(0, V * crops, 6):                                    # two videos per batch
        sl = uid[a:a + 6]
        meter.update_stats(uniq[sl].cuda(), labels_v[sl // crops].cuda(), sl.cuda(), repeat=views)
    assert np.array_equal(meter.video_preds.cpu().numpy(), ref.video_preds)
    assert np.array_equal(meter.clip_count.cpu().numpy(), ref.clip_count)
    assert meter.topks_correct((1, 5)) == ref.finalize_metrics((1, 5))[1]


def test_meter_reports_what_the_reference_asserts():
    from svit_amd.evaluate import TestMeter
    meter = TestMeter(4, 3, 7, 1)
    p = torch.rand(3, 7).cuda()
    meter.update_stats(p, torch.tensor([2, 2, 3]).cuda(), torch.tensor([0, 1, 2]).cuda())  # label flips
    with pytest.raises(AssertionError):
        meter.finalize_metrics((1, 5))
    meter = TestMeter(4, 3, 7, 1)
    meter.update_stats(p, torch.tensor([2, 2, 2]).cuda(), torch.tensor([0, 1, 12]).cuda())  # video 4 of 4
    with pytest.raises(IndexError):
        meter.finalize_metrics((1, 5))
    with pytest.raises(NotImplementedError):
        TestMeter(4, 3, 7, 1, ensemble_method="mean")
    with pytest.raises(ValueError):
        TestMeter(4, 3, 7, 1).update_stats(torch.rand(3, 6).cuda(), torch.zeros(3).cuda(), torch.zeros(3).cuda())


def test_three_crop_ensemble_end_to_end():
    """perform_test on a tiny model: 3 videos x 3 spatial crops (dedupe: x10 views folded), eval
    probabilities of the HIP path vs the fp32 oracle through the oracle's meter."""
    from svit_amd import evaluate
    cfg, model, spec, sd = S.build_hip_model(4, 64, train=False)
    V, crops = 3, cfg.TEST.NUM_SPATIAL_CROPS
    wide = P.tensor("input:wide", (V, 3, 4, 64, 85), amp=1.0)
    labels_v = torch.tensor([5, 17, 101])
    clips = evaluate.spatial_crops(wide.cuda(), 64, crops)               # [9,3,4,64,64]
    for v in range(V):
        for s in range(3):
            y, x = meter_ref.uniform_crop_offsets(64, 85, 64, s)
            assert torch.equal(clips[v * 3 + s].cpu(), wide[v, :, :, y:y + 64, x:x + 64])
    ids = torch.arange(V * crops)
    loader = [([clips[a:a + 4]], labels_v[ids[a:a + 4] // crops].cuda(), ids[a:a + 4].cuda(), {})
              for a in range(0, V * crops, 4)]
    meter = evaluate.TestMeter(V, crops, cfg.MODEL.NUM_CLASSES, len(loader))
    evaluate.perform_test(loader, model, meter, cfg)
    ref = meter_ref.TestMeterRef(V, crops * cfg.TEST.NUM_ENSEMBLE_VIEWS, cfg.MODEL.NUM_CLASSES)
    with torch.no_grad():
        probs, _ = R.forward({k: v for k, v in sd.items()}, spec, clips.cpu(), training=False)
    for rep in range(cfg.TEST.NUM_ENSEMBLE_VIEWS):
        ref.update_stats(probs.numpy(), labels_v[ids // crops].numpy(),
                         (ids // crops * crops * cfg.TEST.NUM_ENSEMBLE_VIEWS + rep * crops + ids % crops).numpy())
    got = meter.video_preds.cpu().numpy()
    assert np.array_equal(meter.clip_count.cpu().numpy(), ref.clip_count)          # 30 per video
    np.testing.assert_allclose(got, ref.video_preds, atol=0.03)      # sums of 30 probabilities
    assert S.cosine(torch.from_numpy(got), torch.from_numpy(ref.video_preds)) > 0.999
    # the device top-k equals the oracle's counting rule applied to the same scores
    assert meter.topks_correct((1, 5)) == meter_ref.topks_correct(got, labels_v.numpy(), (1, 5))
    assert set(meter.stats) == {"split", "top1_acc", "top5_acc"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _merge_worker(rank, world, port, golden_dir, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from svit_amd.evaluate import TestMeter
    g = np.load(os.path.join(golden_dir, "meter.npz"))
    probs, labels_v, perm = torch.from_numpy(g["probs"]), torch.from_numpy(g["labels_v"]), g["perm_shuffled"]
    meter = TestMeter(12, 6, 29, 1)
    for i, a in enumerate(range(0, len(perm), 8)):
        if i % world != rank:                     # DistributedSampler-like split of the batches
            continue
        ids = torch.from_numpy(perm[a:a + 8])
        meter.update_stats(probs[ids].cuda(), labels_v[ids // 6].cuda(), ids.cuda())
    meter.all_reduce()
    stats = meter.finalize_metrics((1, 5))
    if rank == 0:
        torch.save({"preds": meter.video_preds.cpu(), "count": meter.clip_count.cpu(),
                    "labels": meter.video_labels.cpu(), "stats": stats}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_merge_with_one_allreduce(tmp_path, golden_dir, manifest):
    port, out = _free_port(), str(tmp_path / "m.pt")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_merge_worker, args=(r, 2, port, golden_dir, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = torch.load(out)
    g = np.load(os.path.join(golden_dir, "meter.npz"))
    np.testing.assert_allclose(res["preds"].numpy(), g["shuffled_sum_video_preds"], rtol=1e-6, atol=1e-7)
    assert np.array_equal(res["count"].numpy(), g["shuffled_sum_clip_count"])
    assert np.array_equal(res["labels"].numpy(), g["shuffled_sum_video_labels"])
    want = manifest["meter"]["results"]["shuffled_sum"]
    assert res["stats"]["top1_acc"] == want["top1_acc"] and res["stats"]["top5_acc"] == want["top5_acc"]


def _rccl_eval_worker(port, out, use_rccl):
    """perform_test of a tiny model on 3 videos x 3 crops; with `use_rccl` inside a ONE-rank process group of the production
    backend (nccl = RCCL) with the meter's merge forced."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    if use_rccl:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from svit_amd import evaluate
    cfg, model, spec, sd = S.build_hip_model(4, 64, train=False)
    V, crops = 3, cfg.TEST.NUM_SPATIAL_CROPS
    wide = P.tensor("input:wide", (V, 3, 4, 64, 85), amp=1.0)
    labels_v = torch.tensor([5, 17, 101])
    clips = evaluate.spatial_crops(wide.cuda(), 64, crops)
    ids = torch.arange(V * crops)
    loader = [([clips[a:a + 4]], labels_v[ids[a:a + 4] // crops].cuda(), ids[a:a + 4].cuda(), {})
              for a in range(0, V * crops, 4)]
    meter = evaluate.TestMeter(V, crops, cfg.MODEL.NUM_CLASSES, len(loader))
    evaluate.perform_test(loader, model, meter, cfg, force_collectives=use_rccl)
    info = {"preds": meter.video_preds.cpu(), "count": meter.clip_count.cpu(), "labels": meter.video_labels.cpu(),
            "stats": meter.stats, "backend": dist.get_backend() if use_rccl else "none"}
    if use_rccl:
        info["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        dist.barrier()
        dist.destroy_process_group()
    torch.save(info, out)


def test_eval_leg_on_the_production_backend_with_one_rank(tmp_path):
    """Round 6: the eval leg of BASELINE config C5 -- `perform_test`'s end-of-loop merge of the device-resident
    ensemble (tools/test_net.py:128-150 replaced by ONE set of all-reduces, TestMeter.all_reduce) -- executed on the production
    backend: init_process_group("nccl", world_size=1, device_id=...) in a child process, the four collectives (SUM of the
    per-video scores and clip counts, MAX of the labels, SUM of the error flags) forced on the one-rank group.  A merge over one
    rank is the identity: scores, counts, labels and the top-k strings must be BIT-equal to the same loop with no process
    group.  Says that RCCL accepts the meter's dtypes / ops (fp32 SUM, int64 SUM / MAX, int32 SUM) beside libsvit_hip.so;
    N > 1 over xGMI stays the driver's to run."""
    ctx = mp.get_context("spawn")
    outs = {}
    for use_rccl in (False, True):
        out = str(tmp_path / ("ev%d.pt" % use_rccl))
        p = ctx.Process(target=_rccl_eval_worker, args=(_free_port(), out, use_rccl))
        p.start()
        p.join(600)
        assert p.exitcode == 0, (use_rccl, p.exitcode)
        outs[use_rccl] = torch.load(out)
    a, b = outs[False], outs[True]
    assert b["backend"] == "nccl" and b["nccl_version"]
    assert torch.equal(a["preds"], b["preds"]) and torch.equal(a["count"], b["count"]) and torch.equal(a["labels"], b["labels"])
    assert a["stats"] == b["stats"] and int(b["count"].min()) == 30


def test_uint8_frames_equal_reference_normalised_clips():
    """model([U8Clips]) == model([fp32 crops normalised like
    datasets/utils.py:287-303]) -- identical patch-embed operand, hence identical probabilities --
    and the 3-crop ensemble runs from ONE uint8 copy of each video."""
    from svit_amd import evaluate
    from svit_amd.input import spatial_crops_u8
    cfg, model, spec, sd = S.build_hip_model(4, 64, train=False)
    g = torch.Generator().manual_seed(9)
    u8 = torch.randint(0, 256, (2, 4, 64, 85, 3), generator=g, dtype=torch.uint8)
    clips = spatial_crops_u8(u8.cuda(), 64, 3, mean=cfg.DATA.MEAN, std=cfg.DATA.STD)
    assert tuple(clips.shape) == (6, 3, 4, 64, 64)
    t = u8.float() / 255.0
    t = (t - torch.tensor(cfg.DATA.MEAN)) / torch.tensor(cfg.DATA.STD)
    wide = t.permute(0, 4, 1, 2, 3).contiguous().cuda()
    ref_clips = evaluate.spatial_crops(wide, 64, 3)
    with torch.no_grad():
        p_u8, e_u8 = model([clips], {})
        p_f32, e_f32 = model([ref_clips], {})
    assert torch.equal(p_u8, p_f32) and torch.equal(e_u8["pred_bboxes"], e_f32["pred_bboxes"])
    # and against the fp32 oracle on the reference-normalised crops
    with torch.no_grad():
        probs, _ = R.forward(dict(sd), spec, ref_clips.cpu(), training=False)
    np.testing.assert_allclose(p_u8.cpu().numpy(), probs.numpy(), atol=4e-3)
    # training step through the uint8 input: same gradients as through the fp32 clips
    model.train()
    y = torch.tensor([3, 50, 7, 9, 100, 20]).cuda()
    grads = []
    for inp in (clips, ref_clips):
        model.zero_grad(set_to_none=True)
        logits, _ = model([inp], {})
        torch.nn.functional.cross_entropy(logits, y).backward()
        torch.cuda.synchronize()
        grads.append(model.flat.grad.detach().clone())
    assert S.cosine(grads[0], grads[1]) > 0.99999
    assert float((grads[0] - grads[1]).abs().max()) <= 2e-3 * float(grads[1].abs().max())


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_device_meter_random_streams_bit_exact(seed):
    """seeded random streams: any batch size, several clips of one video inside a batch,
    interleaved videos, both ensemble methods, folded repeats -- bit-identical to the restatement
    of the reference's sequential loop."""
    from svit_amd.evaluate import TestMeter
    rng = np.random.default_rng(seed)
    V, num_clips, C = int(rng.integers(3, 40)), int(rng.integers(1, 7)), int(rng.integers(2, 300))
    method = "sum" if seed % 2 == 0 else "max"
    repeat = int(rng.integers(1, 4))
    labels_v = rng.integers(0, C, V)
    ids_all = rng.permutation(V * num_clips)
    meter = TestMeter(V, num_clips, C, 1, ensemble_method=method)
    ref = meter_ref.TestMeterRef(V, num_clips, C, method)
    a = 0
    while a < len(ids_all):
        n = int(rng.integers(1, 33))
        ids = ids_all[a:a + n]
        a += n
        preds = rng.random((len(ids), C), dtype=np.float32)
        lab = labels_v[ids // num_clips]
        meter.update_stats(torch.from_numpy(preds).cuda(), torch.from_numpy(lab).cuda(),
                           torch.from_numpy(ids).cuda(), repeat=repeat)
        for _ in range(repeat):          # cyclic fold == the batch replayed `repeat` times in order
            ref.update_stats(preds, lab, ids)
    assert np.array_equal(meter.video_preds.cpu().numpy(), ref.video_preds)
    assert np.array_equal(meter.clip_count.cpu().numpy(), ref.clip_count)
    assert np.array_equal(meter.video_labels.cpu().numpy(), ref.video_labels)
    ks = (1, min(5, C))
    assert meter.topks_correct(ks) == ref.finalize_metrics(ks)[1]
