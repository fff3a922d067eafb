from .rover_env import RoverEnv, RLTaskEnv  # noqa: F401
