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
        """skrl 1.1.0 call order for one agent: reset once; per step act (no grad) -> env.step -> render unless headless ->
        record_transition -> the BASE agent's post_interaction (logging only, no learning) -> carry the next states over
        (vectorised envs reset themselves)."""
        import torch
        assert self.num_simultaneous_agents == 1
        base = type(self.agents).__mro__[1]
        states, infos = self.env.reset()
        for timestep in range(self.initial_timestep, self.timesteps):
            with torch.no_grad():
                actions = self.agents.act(states, timestep=timestep, timesteps=self.timesteps)[0]
                next_states, rewards, terminated, truncated, infos = self.env.step(actions)
                if not self.headless:
                    self.env.render()
                self.agents.record_transition(states=states, actions=actions, rewards=rewards, next_states=next_states,
                                              terminated=terminated, truncated=truncated, infos=infos, timestep=timestep,
                                              timesteps=self.timesteps)
                base.post_interaction(self.agents, timestep=timestep, timesteps=self.timesteps)
                if self.env.num_envs > 1:
                    states = next_states
                elif terminated.any() or truncated.any():
                    states, infos = self.env.reset()
                else:
                    states = next_states
