"""skrl.agents.torch.Agent (1.1.0 call semantics): init / set_running_mode / pre_interaction / act / record_transition /
post_interaction / track_data.  TensorBoard, wandb and checkpoint files are replaced by in-memory records."""
import collections
import copy

import torch


class Agent:
    def __init__(self, models, memory=None, observation_space=None, action_space=None, device=None, cfg=None):
        self.models, self.memory = models, memory
        self.observation_space, self.action_space = observation_space, action_space
        self.cfg = cfg if cfg is not None else {}
        self.device = torch.device(device if device is not None else "cpu")
        for m in self.models.values():
            if m is not None:
                m.to(self.device)
        self.tracking_data = collections.defaultdict(list)
        self.written = []            # (timestep, {tag: mean}) -- what the TensorBoard writer would have received
        self.checkpoints = []        # timesteps at which a checkpoint would have been written
        exp = self.cfg.get("experiment", {})
        self.write_interval = exp.get("write_interval", 1000)
        self.checkpoint_interval = exp.get("checkpoint_interval", 1000)
        self.training = True
        self._cumulative_rewards = None
        self._initialised = 0

    def init(self, trainer_cfg=None):
        self._initialised += 1
        self.trainer_cfg = copy.copy(trainer_cfg) if trainer_cfg is not None else {}

    def load(self, path):
        """skrl 1.1.0: a checkpoint is {module name: state_dict or object}; every registered module present in it is restored"""
        modules = torch.load(path, map_location=self.device, weights_only=False)
        self.loaded = []
        for name, data in modules.items():
            module = getattr(self, "checkpoint_modules", {}).get(name)
            if module is None:
                continue
            if hasattr(module, "load_state_dict"):
                module.load_state_dict(data)
                if hasattr(module, "eval"):
                    module.eval()
            self.loaded.append(name)

    def track_data(self, tag, value):
        self.tracking_data[tag].append(value)

    def set_running_mode(self, mode):
        self.training = mode == "train"
        for m in self.models.values():
            if m is not None:
                m.set_mode(mode)

    set_mode = set_running_mode

    def pre_interaction(self, timestep, timesteps):
        pass

    def act(self, states, timestep, timesteps):
        raise NotImplementedError

    def record_transition(self, states, actions, rewards, next_states, terminated, truncated, infos, timestep, timesteps):
        if self._cumulative_rewards is None:
            self._cumulative_rewards = torch.zeros_like(rewards, dtype=torch.float32)
        self._cumulative_rewards.add_(rewards)
        done = (terminated + truncated).nonzero(as_tuple=False)
        if done.numel():
            self.tracking_data["Reward / Total reward (mean)"].append(float(self._cumulative_rewards[done[:, 0]].mean()))
            self._cumulative_rewards[done[:, 0]] = 0

    def post_interaction(self, timestep, timesteps):
        timestep += 1
        if timestep > 1 and self.checkpoint_interval > 0 and not timestep % self.checkpoint_interval:
            self.checkpoints.append(timestep)
        if timestep > 1 and self.write_interval > 0 and not timestep % self.write_interval:
            self.written.append((timestep, {k: float(sum(v) / len(v)) for k, v in self.tracking_data.items() if len(v)}))
            self.tracking_data.clear()
