class KLAdaptiveRL:
    """KL-adaptive learning-rate scheduler (interface only: step(kl))."""

    def __init__(self, optimizer, kl_threshold=0.008, min_lr=1e-6, max_lr=1e-2, kl_factor=2, lr_factor=1.5, **kwargs):
        self.optimizer, self.kl_threshold, self.min_lr, self.max_lr = optimizer, kl_threshold, min_lr, max_lr
        self.kl_factor, self.lr_factor = kl_factor, lr_factor

    def step(self, kl=None):
        if kl is None:
            return
        for g in self.optimizer.param_groups:
            if kl > self.kl_threshold * self.kl_factor:
                g["lr"] = max(g["lr"] / self.lr_factor, self.min_lr)
            elif kl < self.kl_threshold / self.kl_factor:
                g["lr"] = min(g["lr"] * self.lr_factor, self.max_lr)
