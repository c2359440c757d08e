"""Learning-rate schedule of the training recipe: linear warm-up, cosine decay, warm restarts with
geometrically growing cycles — the schedule of the reference's ``CosineAnnealingWarmupRestarts``
(utils/train_utils.py:18-145; constructed at model/parq_lightning.py:169-192, stepped once per epoch).

Written as a pure function of the epoch counter (``lr_at``) plus a thin stateful wrapper with the reference's constructor
and ``step()`` so that it can be handed to Lightning or driven by hand; parity with the reference's sequences is pinned
by tests/golden/g13_lr_schedule.npz.
"""
from __future__ import annotations

import math


def cycle_position(epoch: int, first_cycle_steps: int, cycle_mult: float, warmup_steps: int):
    """(cycle index, step inside the cycle, length of that cycle) after `epoch` scheduler steps.  Cycle k+1 has
    int((len_k - warmup) * cycle_mult) + warmup steps (utils/train_utils.py:102-111)."""
    cycle, length, pos = 0, first_cycle_steps, epoch
    while pos >= length:
        pos -= length
        length = int((length - warmup_steps) * cycle_mult) + warmup_steps
        cycle += 1
    return cycle, pos, length


def lr_at(epoch: int, first_cycle_steps: int, cycle_mult: float, max_lr: float, min_lr: float, warmup_steps: int,
          gamma: float = 1.0) -> float:
    """Learning rate at scheduler epoch `epoch` (epoch 0 = at construction, epoch k = after k calls of step())."""
    cycle, pos, length = cycle_position(epoch, first_cycle_steps, cycle_mult, warmup_steps)
    peak = max_lr * gamma ** cycle
    if pos < warmup_steps:
        return (peak - min_lr) * pos / warmup_steps + min_lr
    return min_lr + (peak - min_lr) * (1.0 + math.cos(math.pi * (pos - warmup_steps) / (length - warmup_steps))) / 2.0


class CosineAnnealingWarmupRestarts:
    """Same constructor arguments and ``step()`` / ``state_dict()`` surface as the reference class.  Before the first
    ``step()`` the optimizer runs at ``min_lr`` (``max_lr`` when there is no warm-up), as in the reference
    (utils/train_utils.py:62-69)."""

    def __init__(self, optimizer, first_cycle_steps: int, cycle_mult: float = 1.0, max_lr: float = 0.1, min_lr: float = 0.001,
                 warmup_steps: int = 0, gamma: float = 1.0, last_epoch: int = -1):
        assert warmup_steps < first_cycle_steps
        self.optimizer = optimizer
        self.first_cycle_steps, self.cycle_mult = first_cycle_steps, cycle_mult
        self.max_lr, self.min_lr, self.warmup_steps, self.gamma = max_lr, min_lr, warmup_steps, gamma
        # torch's scheduler base class performs one step() inside its constructor, so the reference sits at epoch
        # last_epoch + 1 once constructed (and then overwrites the rate with min_lr / max_lr, which is lr_at(0) anyway)
        self.last_epoch = last_epoch + 1
        self._apply(self.get_last_lr()[0] if last_epoch >= 0 else (self.min_lr if warmup_steps != 0 else self.max_lr))

    def _apply(self, lr):
        for g in self.optimizer.param_groups:
            g["lr"] = lr

    def get_last_lr(self):
        lr = lr_at(self.last_epoch, self.first_cycle_steps, self.cycle_mult, self.max_lr, self.min_lr, self.warmup_steps, self.gamma)
        return [lr] * len(self.optimizer.param_groups)

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else int(math.floor(epoch))
        self._apply(self.get_last_lr()[0])

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, state):
        self.__dict__.update(state)
        self._apply(self.get_last_lr()[0])
