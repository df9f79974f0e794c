"""Epoch-level LR schedules of the reference (core/scheduler.py:4-124); scalar host math.  The sequences are pinned against the
reference's own classes in tests/golden/schedulers.npz (tests/test_host_cpu.py::test_schedulers_match_the_reference)."""
import math


class _Sched:
    def __init__(self, optimizer):
        self.optimizer = optimizer
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_lrs = [g["initial_lr"] for g in optimizer.param_groups]
        # the reference applies lr(0) at construction and then puts last_epoch back to -1 (scheduler.py:19-20), so the first
        # `step()` of the epoch loop applies lr(0) AGAIN: epochs run at lr(0), lr(0), lr(1), lr(2), ...
        self.step(0)
        self.last_epoch = -1

    def get_lr(self):
        raise NotImplementedError

    def get_last_lr(self):
        return self.get_lr()

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
        for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
            g["lr"] = lr


class CosineSchedule(_Sched):
    """base_lr * cos(99*pi*epoch / (200*(K-1)))  (scheduler.py:47-62)"""

    def __init__(self, optimizer, K):
        self.K = K
        super().__init__(optimizer)

    def get_lr(self):
        k = 2 if self.K == 1 else self.K
        return [b * math.cos((99 * math.pi * self.last_epoch) / (200 * (k - 1))) for b in self.base_lrs]


class CosineAnnealingWarmUp(_Sched):
    def __init__(self, optimizer, warmup_length, T_max=0, last_epoch=-1):
        self.warmup_length, self.T_max = warmup_length, T_max
        super().__init__(optimizer)

    def get_lr(self):
        if self.last_epoch < self.warmup_length:
            return [b * (self.last_epoch + 1) / self.warmup_length for b in self.base_lrs]
        return [b * 0.5 * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) for b in self.base_lrs]


class PatienceSchedule:
    """divide lr by `factor` after `patience` epochs without loss improvement (scheduler.py:91-124)"""

    def __init__(self, optimizer, patience, factor):
        self.optimizer, self.patience, self.factor = optimizer, patience, factor
        self.best_loss, self.counter = float("inf"), 0

    def step(self, current_loss=None, **kwargs):
        if current_loss is None:
            return
        if current_loss < self.best_loss:
            self.best_loss, self.counter = current_loss, 0
        else:
            self.counter += 1
        if self.counter >= self.patience:
            for g in self.optimizer.param_groups:
                g["lr"] /= self.factor
            self.counter = 0

    def get_last_lr(self):
        return self.optimizer.param_groups[0]["lr"]
