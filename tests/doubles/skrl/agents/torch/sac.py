"""Interface placeholder: only PPO is exercised through the test double."""
from . import Agent

SAC_DEFAULT_CONFIG = {"experiment": {}}


class SAC(Agent):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("the skrl test double implements PPO only")
