"""Interface placeholder: only PPO is exercised through the test double."""
from . import Agent

TRPO_DEFAULT_CONFIG = {"experiment": {}}


class TRPO(Agent):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("the skrl test double implements PPO only")
