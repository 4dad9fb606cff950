"""Learning-rate and temperature schedules of the reference training loop (row C6 of SURVEY.md 8a) as plain
functions of the step: the fused AdamW takes the learning rate as a per-step argument
(``PriorTrainer.train_step(..., lr=...)`` / ``replay_step(lr=...)``), so no optimizer object is involved.

  * ``OneCycleLR``: ``torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr, total_steps = epochs * len(dl) * 5,
    final_div_factor = 1000, pct_start = 2 / epochs)`` with torch's defaults (cosine annealing, div_factor 25,
    two phases, ``cycle_momentum=True``: AdamW's beta1 runs 0.95 -> 0.85 while the rate rises and back to 0.95 while it
    falls) - train_diffusion_prior.py:343-357; ``lr_at(step)`` and ``momentum_at(step)`` are what the optimizer step
    number ``step`` runs with;
  * ``cosine_anneal`` - train_diffusion_prior.py:122-123 (soft-CLIP temperatures per epoch).
"""
import math


class OneCycleLR:
    def __init__(self, max_lr, total_steps, pct_start=0.3, div_factor=25.0, final_div_factor=1e4,
                 cycle_momentum=True, base_momentum=0.85, max_momentum=0.95):
        if total_steps <= 0:
            raise ValueError("total_steps must be positive")
        self.max_lr, self.total_steps = float(max_lr), int(total_steps)
        self.cycle_momentum = bool(cycle_momentum)
        self.base_momentum, self.max_momentum = float(base_momentum), float(max_momentum)
        self.initial_lr = self.max_lr / div_factor
        self.min_lr = self.initial_lr / final_div_factor
        self.up_end = float(pct_start * total_steps) - 1.0          # torch: phase 1 ends at pct_start * total - 1
        self.down_end = float(total_steps) - 1.0
        self.last_step = 0

    @staticmethod
    def _cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    def lr_at(self, step):
        """Learning rate of optimizer step number ``step`` (0-based: the rate the first step runs with is lr_at(0))."""
        if step >= self.total_steps:
            raise ValueError(f"step {step} beyond total_steps {self.total_steps}")
        if step <= self.up_end:
            return self._cos(self.initial_lr, self.max_lr, step / self.up_end)
        return self._cos(self.max_lr, self.min_lr, (step - self.up_end) / (self.down_end - self.up_end))

    def momentum_at(self, step):
        """AdamW beta1 of optimizer step ``step`` (torch's ``cycle_momentum``: inverse to the rate, same two cosine
        phases), or None when the scheduler leaves the optimizer's betas alone."""
        if not self.cycle_momentum:
            return None
        if step >= self.total_steps:
            raise ValueError(f"step {step} beyond total_steps {self.total_steps}")
        if step <= self.up_end:
            return self._cos(self.max_momentum, self.base_momentum, step / self.up_end)
        return self._cos(self.base_momentum, self.max_momentum, (step - self.up_end) / (self.down_end - self.up_end))

    def get_last_lr(self):
        return [self.lr_at(self.last_step)]

    def step(self):
        self.last_step += 1

    def state_dict(self):
        return dict(self.__dict__)

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


def reference_schedule(max_lr, num_epochs, steps_per_epoch):
    """The scheduler `trainer()` builds (train_diffusion_prior.py:341-357)."""
    return OneCycleLR(max_lr, int(num_epochs * steps_per_epoch) * 5, pct_start=2 / num_epochs, final_div_factor=1000)


def cosine_anneal(start, end, steps):
    """train_diffusion_prior.py:122-123 as a list of floats."""
    if steps == 1:
        return [float(start)]
    return [end + (start - end) / 2.0 * (1.0 + math.cos(math.pi * i / (steps - 1))) for i in range(steps)]
