from .p3p import P3PPoseEstimator  # noqa: F401
