"""guidemaker_amd -- MI355X-native neighbour search + CFD scoring behind GuideMaker's
TargetProcessor interface (see DESIGN.md / INTEGRATION.md).  Requires libgmhip.so (HIP, gfx950)."""
from . import _lib
from . import core
from .core import PamTarget, TargetProcessor, cfd_score, CONFIG_PATH
from ._lib import Index, GmHipError

__version__ = "0.1.0"
__all__ = ["PamTarget", "TargetProcessor", "cfd_score", "CONFIG_PATH", "Index", "GmHipError", "core"]
