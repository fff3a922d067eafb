"""skrl.trainers.torch.Trainer (1.1.0): stores env / agents / cfg; subclasses implement train() / eval()."""
import os


class Trainer:
    def __init__(self, env, agents, agents_scope=None, cfg=None):
        self.cfg = cfg if cfg is not None else {}
        self.env, self.agents = env, agents
        self.agents_scope = agents_scope if agents_scope is not None else []
        self.timesteps = self.cfg.get("timesteps", 0)
        cap = os.environ.get("SKRL_DOUBLE_MAX_TIMESTEPS")       # test hook: bound the reference's hard-coded 1 000 000 steps
        if cap is not None:
            self.timesteps = min(self.timesteps, int(cap))
        self.headless = self.cfg.get("headless", False)
        self.disable_progressbar = self.cfg.get("disable_progressbar", False)
        self.close_environment_at_exit = self.cfg.get("close_environment_at_exit", True)
        self.initial_timestep = 0
        self.num_simultaneous_agents = len(agents) if isinstance(agents, (list, tuple)) else 1

    def train(self):
        raise NotImplementedError

    def eval(self):
        raise NotImplementedError

    def single_agent_eval(self):
        raise NotImplementedError
