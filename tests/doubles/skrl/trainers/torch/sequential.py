from . import Trainer

SEQUENTIAL_TRAINER_DEFAULT_CONFIG = {"timesteps": 100000, "headless": False, "disable_progressbar": False,
                                     "close_environment_at_exit": True}


class SequentialTrainer(Trainer):
    pass
