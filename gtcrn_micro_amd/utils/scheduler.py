"""Learning-rate schedule of the reference's trainer, mirrored for drop-in use (same class name, constructor
arguments and ``compute_lr`` helper as ``gtcrn_micro/utils/scheduler.py:28-61``).

Shape of the schedule (conf/cfg_train_DNS3.yaml: 25 000 warm-up steps, decay until 250 000, 1e-3 -> 1e-6):

    step <  warmup_steps          lr rises linearly from 0:      max_lr * step / warmup_steps
    warmup_steps <= step < decay  half cosine from max_lr to min_lr
    step >= decay_until_step      min_lr
"""
from math import cos, pi

from torch.optim.lr_scheduler import _LRScheduler


def _piecewise(step, n_warm, n_decay, lr_hi, lr_lo):
    if step < n_warm:                       # linear ramp (lr is exactly 0 at step 0)
        return lr_hi * (step / n_warm)
    if step >= n_decay:                     # floor
        return lr_lo
    progress = (step - n_warm) / float(n_decay - n_warm)
    return lr_lo + (lr_hi - lr_lo) * 0.5 * (1.0 + cos(pi * progress))


class LinearWarmupCosineAnnealingLR(_LRScheduler):
    def __init__(self, optimizer, warmup_steps, decay_until_step, max_lr, min_lr, last_epoch=-1):
        self.warmup_steps = int(warmup_steps)
        self.decay_until_step = int(decay_until_step)
        self.max_lr = float(max_lr)
        self.min_lr = float(min_lr)
        super().__init__(optimizer, last_epoch)

    compute_lr = staticmethod(_piecewise)

    def get_lr(self):
        value = _piecewise(self.last_epoch, self.warmup_steps, self.decay_until_step, self.max_lr, self.min_lr)
        return [value] * len(self.optimizer.param_groups)
