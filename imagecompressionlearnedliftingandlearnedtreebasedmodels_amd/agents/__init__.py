"""``from agents import *`` exposes the agent classes so that main.py:30 (``globals()[config.agent]``) resolves them."""
from .base import BaseAgent
from .liftingDWT_agent import LiftingBasedDWTAgent

__all__ = ["BaseAgent", "LiftingBasedDWTAgent"]
