"""TEST DOUBLE of skrl 1.1.0 (https://github.com/Toni-SM/skrl), which is not installable in this environment.

Only what the reference's trainer glue touches is provided -- ``rover_envs/utils/skrl_utils.py:15-41`` (SkrlVecEnvWrapper),
``:96-148`` (SkrlSequentialLogTrainer.train), ``rover_envs/learning/train/*`` (agent factories), ``rover_envs/envs/navigation/
learning/skrl/models.py`` (models), ``rover_envs/utils/config.py`` (YAML conversion) -- with the call semantics of skrl 1.1.0
restated from its documentation.  PPO is a small but real implementation (GAE, clipped surrogate, Adam),
so the reference's loop really acts, records and updates.  Lives under tests/: never imported by the product package.
"""
__version__ = "1.1.0+double"
