import numpy as np
import torch


def _size(space):
    if space is None:
        return None
    if isinstance(space, int):
        return space
    if hasattr(space, "shape") and space.shape is not None:
        return int(np.prod(space.shape))
    raise ValueError(f"unsupported space {space!r}")


class Model(torch.nn.Module):
    def __init__(self, observation_space, action_space, device=None):
        super().__init__()
        self.device = torch.device(device if device is not None else "cpu")
        self.observation_space, self.action_space = observation_space, action_space
        self.num_observations, self.num_actions = _size(observation_space), _size(action_space)
        self._random_distribution = None

    def set_mode(self, mode: str):
        self.train(mode == "train")

    def compute(self, inputs, role=""):
        raise NotImplementedError

    def forward(self, inputs, role=""):
        return self.act(inputs, role)

    def freeze_parameters(self, freeze=True):
        for p in self.parameters():
            p.requires_grad = not freeze

    def update_parameters(self, model, polyak=1.0):
        with torch.no_grad():
            for p, q in zip(self.parameters(), model.parameters()):
                p.data.mul_(1 - polyak).add_(polyak * q.data)
