from .lift_env import FrankaCubeLiftEnv, LiftEnvCfg  # noqa: F401
from .rover_env import RoverEnv, RLTaskEnv  # noqa: F401
