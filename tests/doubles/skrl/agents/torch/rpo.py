"""Interface placeholder: only PPO is exercised through the test double."""
from . import Agent

RPO_DEFAULT_CONFIG = {"experiment": {}}


class RPO(Agent):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("the skrl test double implements PPO only")
