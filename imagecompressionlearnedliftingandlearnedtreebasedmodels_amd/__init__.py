"""MI355X-native learned-lifting DWT + CNN entropy-model hot path (gfx950 / CDNA4).

Host side mirrors the reference's Python module API (agents/liftingDWT_agent.py, graphs/models/LiftingBasedDWT_net.py);
all device work goes through the C-ABI library ``liblldwt.so`` (include/lldwt.h) -- there is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
