from . import video  # noqa: F401
