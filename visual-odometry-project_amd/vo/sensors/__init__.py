from .camera import Camera  # noqa: F401
