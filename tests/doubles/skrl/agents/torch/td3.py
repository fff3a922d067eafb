"""Interface placeholder: only PPO is exercised through the test double."""
from . import Agent

TD3_DEFAULT_CONFIG = {"experiment": {}}


class TD3(Agent):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("the skrl test double implements PPO only")
