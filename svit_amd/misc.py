"""Step-loop helpers of the reference's trainer that sit on the hot path's boundary
(slowfast/utils/misc.py:25-35, tools/train_net.py:131,153-160)."""
import math
from datetime import datetime

import torch


def check_nan_losses(loss, extra_msg=None):
    """misc.py:25-35, unchanged semantics: raises RuntimeError on a NaN loss.  `math.isnan` on a
    device tensor is a device->host sync, once per step in the reference's loop
    (tools/train_net.py:131)."""
    if math.isnan(loss):
        msg = "ERROR: Got NaN losses {}".format(datetime.now())
        if extra_msg is not None:
            msg = msg + f"extra_msg: {extra_msg}"
        raise RuntimeError(msg)


class NanWatch:
    """The same guarantee without the per-step sync: every step ORs `isnan(loss)` into a device
    flag (one tiny launch, capturable); the host reads it every `period` steps -- cfg.LOG_PERIOD,
    when the reference syncs anyway to log -- and raises then, naming the first bad step.  A
    replayed 15 ms step otherwise stalls behind a host round trip per iteration, which is one of
    the data-parallel scaling risks SURVEY.md 8(e) lists."""

    def __init__(self, device, period=10):
        self.period = max(1, int(period))
        self.first_bad = torch.full((), -1, dtype=torch.int64, device=device)
        self.step = 0

    def update(self, loss):
        bad = torch.isnan(loss.detach()).any()
        unset = self.first_bad < 0
        self.first_bad = torch.where(bad & unset, torch.full_like(self.first_bad, self.step), self.first_bad)
        self.step += 1
        if self.step % self.period == 0:
            self.check()

    def check(self):
        first = int(self.first_bad)          # the one sync per period
        if first >= 0:
            raise RuntimeError("ERROR: Got NaN losses {} (first at step {})".format(datetime.now(), first))
