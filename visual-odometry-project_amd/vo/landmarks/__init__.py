from .triangulation import LandmarksTriangulator  # noqa: F401
