import torch


class DeterministicMixin:
    def __init__(self, clip_actions=False, role=""):
        self._d_clip_actions = clip_actions and hasattr(self.action_space, "low")
        if self._d_clip_actions:
            self._d_clip_min = torch.tensor(self.action_space.low, device=self.device, dtype=torch.float32)
            self._d_clip_max = torch.tensor(self.action_space.high, device=self.device, dtype=torch.float32)

    def act(self, inputs, role=""):
        actions, outputs = self.compute(inputs, role)
        if self._d_clip_actions:
            actions = torch.clamp(actions, min=self._d_clip_min, max=self._d_clip_max)
        return actions, None, outputs
