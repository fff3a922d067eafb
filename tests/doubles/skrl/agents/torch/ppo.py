"""A small, real PPO (GAE, clipped surrogate / value losses, Adam, KL early stop) with skrl 1.1.0's interface."""
import copy
import itertools

import torch

from . import Agent

PPO_DEFAULT_CONFIG = {
    "rollouts": 16, "learning_epochs": 8, "mini_batches": 2, "discount_factor": 0.99, "lambda": 0.95,
    "learning_rate": 1e-3, "learning_rate_scheduler": None, "learning_rate_scheduler_kwargs": {},
    "state_preprocessor": None, "state_preprocessor_kwargs": {}, "value_preprocessor": None, "value_preprocessor_kwargs": {},
    "random_timesteps": 0, "learning_starts": 0, "grad_norm_clip": 0.5, "ratio_clip": 0.2, "value_clip": 0.2,
    "clip_predicted_values": False, "entropy_loss_scale": 0.0, "value_loss_scale": 1.0, "kl_threshold": 0,
    "rewards_shaper": None, "time_limit_bootstrap": False,
    "experiment": {"directory": "", "experiment_name": "", "write_interval": 250, "checkpoint_interval": 1000,
                   "store_separately": False, "wandb": False, "wandb_kwargs": {}},
}


class PPO(Agent):
    instances = []     # test hook: every agent ever built (the reference's train.py keeps its agent in a local variable)

    def __init__(self, models, memory=None, observation_space=None, action_space=None, device=None, cfg=None):
        PPO.instances.append(self)
        _cfg = copy.deepcopy(PPO_DEFAULT_CONFIG)
        _cfg.update(cfg if cfg is not None else {})
        super().__init__(models, memory, observation_space, action_space, device, _cfg)
        self.policy, self.value = self.models.get("policy"), self.models.get("value")
        c = self.cfg
        self._rollouts, self._rollout = c["rollouts"], 0
        self._epochs, self._mini_batches = c["learning_epochs"], c["mini_batches"]
        self._gamma, self._lambda = c["discount_factor"], c["lambda"]
        self._ratio_clip, self._value_clip = c["ratio_clip"], c["value_clip"]
        self._clip_predicted_values = c["clip_predicted_values"]
        self._entropy_scale, self._value_scale = c["entropy_loss_scale"], c["value_loss_scale"]
        self._grad_norm_clip, self._kl_threshold = c["grad_norm_clip"], c["kl_threshold"]
        self._learning_starts = c["learning_starts"]
        self._rewards_shaper = c["rewards_shaper"]
        params = self.policy.parameters() if self.policy is self.value else itertools.chain(self.policy.parameters(),
                                                                                             self.value.parameters())
        self.optimizer = torch.optim.Adam(params, lr=c["learning_rate"])
        sched = c["learning_rate_scheduler"]
        self.scheduler = sched(self.optimizer, **c["learning_rate_scheduler_kwargs"]) if sched is not None else None
        sp, vp = c["state_preprocessor"], c["value_preprocessor"]
        self._state_preprocessor = sp(**c["state_preprocessor_kwargs"]) if sp is not None else (lambda x, **kw: x)
        self._value_preprocessor = vp(**c["value_preprocessor_kwargs"]) if vp is not None else (lambda x, **kw: x)
        self.updates = 0
        self.checkpoint_modules = {"policy": self.policy, "value": self.value, "optimizer": self.optimizer}
        if sp is not None:
            self.checkpoint_modules["state_preprocessor"] = self._state_preprocessor
        if vp is not None:
            self.checkpoint_modules["value_preprocessor"] = self._value_preprocessor

    def init(self, trainer_cfg=None):
        super().init(trainer_cfg)
        self.set_mode("eval")
        if self.memory is not None and not self.memory.tensors:
            self.memory.create_tensor("states", self.observation_space)
            self.memory.create_tensor("actions", self.action_space)
            for name in ("rewards", "log_prob", "values", "returns", "advantages"):
                self.memory.create_tensor(name, 1)
            self.memory.create_tensor("terminated", 1, dtype=torch.bool)
        self._current_log_prob = self._current_next_states = None

    def act(self, states, timestep, timesteps):
        actions, log_prob, outputs = self.policy.act({"states": self._state_preprocessor(states)}, role="policy")
        self._current_log_prob = log_prob
        return actions, log_prob, outputs

    def record_transition(self, states, actions, rewards, next_states, terminated, truncated, infos, timestep, timesteps):
        super().record_transition(states, actions, rewards, next_states, terminated, truncated, infos, timestep, timesteps)
        if self.memory is None:
            return
        self._current_next_states = next_states
        if self._rewards_shaper is not None:
            rewards = self._rewards_shaper(rewards, timestep, timesteps)
        values, _, _ = self.value.act({"states": self._state_preprocessor(states)}, role="value")
        values = self._value_preprocessor(values, inverse=True)
        self.memory.add_samples(states=states, actions=actions, rewards=rewards, terminated=terminated,
                                log_prob=self._current_log_prob, values=values)

    def post_interaction(self, timestep, timesteps):
        self._rollout += 1
        if not self._rollout % self._rollouts and timestep >= self._learning_starts:
            self.set_mode("train")
            self._update(timestep, timesteps)
            self.set_mode("eval")
        super().post_interaction(timestep, timesteps)

    def _update(self, timestep, timesteps):
        mem = self.memory
        with torch.no_grad():
            last_values, _, _ = self.value.act({"states": self._state_preprocessor(self._current_next_states.float())}, role="value")
            last_values = self._value_preprocessor(last_values, inverse=True)
            values, rewards = mem.get_tensor_by_name("values"), mem.get_tensor_by_name("rewards")
            not_done = mem.get_tensor_by_name("terminated").logical_not().float()
            adv, advantages = 0, torch.zeros_like(rewards)
            for i in reversed(range(mem.memory_size)):
                nxt = values[i + 1] if i < mem.memory_size - 1 else last_values
                adv = rewards[i] - values[i] + self._gamma * not_done[i] * (nxt + self._lambda * adv)
                advantages[i] = adv
            returns = advantages + values
            advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
            mem.set_tensor_by_name("values", self._value_preprocessor(values, train=True))
            mem.set_tensor_by_name("returns", self._value_preprocessor(returns, train=True))
            mem.set_tensor_by_name("advantages", advantages)
        names = ["states", "actions", "log_prob", "values", "returns", "advantages"]
        for _ in range(self._epochs):
            kls = []
            for states, actions, old_logp, old_values, rets, advs in mem.sample_all(names, self._mini_batches):
                states = self._state_preprocessor(states, train=True)
                _, logp, _ = self.policy.act({"states": states, "taken_actions": actions}, role="policy")
                with torch.no_grad():
                    ratio_log = logp - old_logp
                    kls.append((((torch.exp(ratio_log) - 1) - ratio_log).mean()))
                if self._kl_threshold and kls[-1] > self._kl_threshold:
                    break
                ratio = torch.exp(logp - old_logp)
                surrogate = -torch.min(advs * ratio, advs * torch.clamp(ratio, 1 - self._ratio_clip, 1 + self._ratio_clip)).mean()
                entropy = -self._entropy_scale * self.policy.get_entropy(role="policy").mean() if self._entropy_scale else 0.0
                pred, _, _ = self.value.act({"states": states}, role="value")
                if self._clip_predicted_values:
                    pred = old_values + torch.clamp(pred - old_values, -self._value_clip, self._value_clip)
                value_loss = self._value_scale * torch.nn.functional.mse_loss(rets, pred)
                self.optimizer.zero_grad()
                (surrogate + entropy + value_loss).backward()
                if self._grad_norm_clip > 0:
                    torch.nn.utils.clip_grad_norm_(itertools.chain(self.policy.parameters(), self.value.parameters()),
                                                   self._grad_norm_clip)
                self.optimizer.step()
                self.track_data("Loss / Policy loss", float(surrogate.detach()))
                self.track_data("Loss / Value loss", float(value_loss.detach()))
            if self.scheduler is not None and kls:
                self.scheduler.step(float(torch.stack(kls).mean()))
        self.updates += 1
