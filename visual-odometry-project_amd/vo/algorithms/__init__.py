from .ransac import RANSAC  # noqa: F401
