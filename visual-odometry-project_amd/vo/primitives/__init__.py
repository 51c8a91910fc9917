from .features import Features  # noqa: F401
from .frame import Frame  # noqa: F401
from .matches import Matches  # noqa: F401
from .state import State  # noqa: F401
from .loader import Sequence  # noqa: F401
