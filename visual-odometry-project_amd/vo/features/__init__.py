from .klt import KLTTracker  # noqa: F401
from .harris import HarrisCornerDetector  # noqa: F401
from .sift import SIFTDetector  # noqa: F401
from .tracker import Tracker  # noqa: F401
