from .detection import Detect                                           # noqa: F401

__all__ = ['Detect']
