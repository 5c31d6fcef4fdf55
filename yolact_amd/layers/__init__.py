from .detection import Detect  # noqa: F401
