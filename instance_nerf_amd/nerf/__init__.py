from .network import NeRFNetwork  # noqa: F401
from .renderer import NeRFRenderer  # noqa: F401
