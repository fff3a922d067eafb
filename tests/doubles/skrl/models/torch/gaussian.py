import torch
from torch.distributions import Normal


class GaussianMixin:
    def __init__(self, clip_actions=False, clip_log_std=True, min_log_std=-20, max_log_std=2, reduction="sum", role=""):
        self._clip_actions = clip_actions and hasattr(self.action_space, "low")
        if self._clip_actions:
            self._clip_min = torch.tensor(self.action_space.low, device=self.device, dtype=torch.float32)
            self._clip_max = torch.tensor(self.action_space.high, device=self.device, dtype=torch.float32)
        self._clip_log_std, self._min_log_std, self._max_log_std = clip_log_std, min_log_std, max_log_std
        self._reduction = {"sum": torch.sum, "mean": torch.mean, "prod": torch.prod, "none": None}[reduction]
        self._distribution = None

    def act(self, inputs, role=""):
        mean_actions, log_std, outputs = self.compute(inputs, role)
        if self._clip_log_std:
            log_std = torch.clamp(log_std, self._min_log_std, self._max_log_std)
        self._log_std = log_std
        self._distribution = Normal(mean_actions, log_std.exp())
        actions = self._distribution.rsample()
        if self._clip_actions:
            actions = torch.clamp(actions, min=self._clip_min, max=self._clip_max)
        log_prob = self._distribution.log_prob(inputs.get("taken_actions", actions))
        if self._reduction is not None:
            log_prob = self._reduction(log_prob, dim=-1)
        if log_prob.dim() != actions.dim():
            log_prob = log_prob.unsqueeze(-1)
        outputs["mean_actions"] = mean_actions
        return actions, log_prob, outputs

    def get_entropy(self, role=""):
        return self._distribution.entropy().to(self.device) if self._distribution is not None else torch.tensor(0.0)

    def get_log_std(self, role=""):
        return self._log_std.repeat(1, 1)

    def distribution(self, role=""):
        return self._distribution
