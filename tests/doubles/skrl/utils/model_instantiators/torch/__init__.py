from .. import Shape  # noqa: F401
