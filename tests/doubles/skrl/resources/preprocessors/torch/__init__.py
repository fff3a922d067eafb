import torch


class RunningStandardScaler(torch.nn.Module):
    def __init__(self, size, epsilon=1e-8, clip_threshold=5.0, device=None):
        super().__init__()
        n = int(size if isinstance(size, int) else torch.tensor(getattr(size, "shape", size)).prod())
        self.epsilon, self.clip = epsilon, clip_threshold
        self.register_buffer("mean", torch.zeros(n, dtype=torch.float64, device=device))
        self.register_buffer("var", torch.ones(n, dtype=torch.float64, device=device))
        self.register_buffer("count", torch.ones((), dtype=torch.float64, device=device))

    def forward(self, x, train=False, inverse=False, no_grad=True):
        if inverse:
            return torch.sqrt(self.var.float()) * torch.clamp(x, -self.clip, self.clip) + self.mean.float()
        if train:
            b = x.reshape(-1, x.shape[-1]).double()
            m, v, n = b.mean(0), b.var(0, unbiased=False), b.shape[0]
            d, tot = m - self.mean, self.count + n
            self.var = (self.var * self.count + v * n + d ** 2 * self.count * n / tot) / tot
            self.mean, self.count = self.mean + d * n / tot, tot
        return torch.clamp((x - self.mean.float()) / (torch.sqrt(self.var.float()) + self.epsilon), -self.clip, self.clip)
