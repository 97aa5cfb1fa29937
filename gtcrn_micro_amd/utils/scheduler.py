"""Mirror of the reference's ``utils/scheduler.py`` LinearWarmupCosineAnnealingLR (:28-61):
lr(step) = max_lr * step / warmup_steps during warm-up, then a half cosine from max_lr down to min_lr
at ``decay_until_step``, min_lr afterwards (conf/cfg_train_DNS3.yaml: 25 000 / 250 000 / 1e-3 / 1e-6)."""
import math

from torch.optim import lr_scheduler


class LinearWarmupCosineAnnealingLR(lr_scheduler._LRScheduler):
    def __init__(self, optimizer, warmup_steps, decay_until_step, max_lr, min_lr, last_epoch=-1):
        self.warmup_steps, self.decay_until_step = warmup_steps, decay_until_step
        self.max_lr, self.min_lr = max_lr, min_lr
        super().__init__(optimizer, last_epoch)

    @staticmethod
    def compute_lr(step, warmup_steps, decay_until_step, max_lr, min_lr):
        if step < warmup_steps:
            return max_lr * step / warmup_steps
        if step >= decay_until_step:
            return min_lr
        ratio = (step - warmup_steps) / (decay_until_step - warmup_steps)
        return min_lr + 0.5 * (1.0 + math.cos(math.pi * ratio)) * (max_lr - min_lr)

    def get_lr(self):
        lr = self.compute_lr(self.last_epoch, self.warmup_steps, self.decay_until_step, self.max_lr, self.min_lr)
        return [lr for _ in self.optimizer.param_groups]
