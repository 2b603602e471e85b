from . import problems  # noqa: F401
