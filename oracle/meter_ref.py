"""CPU restatement of the reference's multi-view test ensemble -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (svit_amd/evaluate.py -> svit_ensemble_update / svit_topk_correct) never does.

Restates, in numpy fp32 with the reference's sequential order of additions:
  * TestMeter.update_stats / finalize_metrics   slowfast/utils/meters.py:303-336,378-398
  * metrics.topks_correct                        slowfast/utils/metrics.py:9-50
  * transform.uniform_crop offsets               slowfast/datasets/transform.py:327-339
  * the test-view table of the ssv2 dataset      slowfast/datasets/ssv2.py:139-150,275-288
Pinned by tests/golden/meter.npz, produced by the reference's own TestMeter / topks_correct /
uniform_crop (oracle/gen_golden.py::run_meter_case).
"""
import math

import numpy as np


class TestMeterRef:
    __test__ = False        # not a pytest class

    def __init__(self, num_videos, num_clips, num_cls, ensemble_method="sum"):
        self.num_clips, self.method = num_clips, ensemble_method
        self.video_preds = np.zeros((num_videos, num_cls), np.float32)
        self.video_labels = np.zeros((num_videos,), np.int64)
        self.clip_count = np.zeros((num_videos,), np.int64)

    def update_stats(self, preds, labels, clip_ids):
        """meters.py:303-336: one clip at a time, in batch order."""
        preds = np.asarray(preds, np.float32)
        for i in range(preds.shape[0]):
            v = int(clip_ids[i]) // self.num_clips
            if self.video_labels[v] > 0:
                assert self.video_labels[v] == int(labels[i]), "clips of one video disagree on its label"
            self.video_labels[v] = int(labels[i])
            if self.method == "sum":
                self.video_preds[v] = self.video_preds[v] + preds[i]
            elif self.method == "max":
                self.video_preds[v] = np.maximum(self.video_preds[v], preds[i])
            else:
                raise NotImplementedError("Ensemble Method {} is not supported".format(self.method))
            self.clip_count[v] += 1

    def finalize_metrics(self, ks=(1, 5)):
        """meters.py:378-398 -> {"top1_acc": "12.34", ...} plus the raw counts."""
        correct = topks_correct(self.video_preds, self.video_labels, ks)
        stats = {"split": "test_final"}
        for k, c in zip(ks, correct):
            stats["top{}_acc".format(k)] = "{:.{prec}f}".format(c / self.video_preds.shape[0] * 100.0, prec=2)
        return stats, correct


def label_rank(row, label):
    """position of `label` in the descending order of `row`; equal scores: lower class first
    (the reference's torch.topk leaves ties unspecified; real probabilities do not tie)."""
    s = row[label]
    return int((row > s).sum() + (row[:label] == s).sum())


def topks_correct(preds, labels, ks):
    """metrics.py:9-50 for 2-d preds: how many rows have their label among the top-k scores."""
    preds = np.asarray(preds)
    ranks = np.array([label_rank(preds[i], int(labels[i])) for i in range(preds.shape[0])])
    return [int((ranks < k).sum()) for k in ks]


def uniform_crop_offsets(height, width, size, spatial_idx):
    """transform.py:327-339 -> (y_offset, x_offset)."""
    assert spatial_idx in (0, 1, 2)
    y = int(math.ceil((height - size) / 2))
    x = int(math.ceil((width - size) / 2))
    if height > width:
        y = 0 if spatial_idx == 0 else (height - size if spatial_idx == 2 else y)
    else:
        x = 0 if spatial_idx == 0 else (width - size if spatial_idx == 2 else x)
    return y, x


def test_views(num_videos, ensemble_views, spatial_crops):
    """ssv2.py:139-150,275-288: dataset index i -> (video, spatial crop).  The temporal sampling of
    a test clip is the segment midpoints (ssv2.py:225-230), i.e. independent of the view index."""
    n = ensemble_views * spatial_crops
    out = []
    for v in range(num_videos):
        for idx in range(n):
            out.append((v, 1 if spatial_crops == 1 else idx % spatial_crops))
    return out
test_views.__test__ = False
