"""Linear learning-rate warm-up keyed on the optimizer's step count
(reference: src/dmme/lr_scheduler/warmup.py:4-19; stepped once per training step)."""

from torch.optim.lr_scheduler import LRScheduler


class WarmupLR(LRScheduler):
    def __init__(self, optimizer, warmup=0.0, last_epoch=-1):
        self.warmup_steps = warmup
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        # the reference keys on optimizer._step_count (+1); newer torch dropped that counter from
        # optimizers, where the scheduler's own call count minus the construction-time call is equal
        done = getattr(self.optimizer, "_step_count", self._step_count - 1) + 1
        scale = done / self.warmup_steps if done < self.warmup_steps else 1.0
        return [g["initial_lr"] * scale for g in self.optimizer.param_groups]
