from .base import Model  # noqa: F401
from .deterministic import DeterministicMixin  # noqa: F401
from .gaussian import GaussianMixin  # noqa: F401
