"""BASELINE.json names ``lamp.Attention.ScaledDotProductAttention``; in the reference that class
lives in lamp/SubLayers.py:16-43 (lamp/Attention.py holds Luong-style attentions nothing imports,
SURVEY.md G1).  Re-exported here so both import paths resolve."""
from .SubLayers import ScaledDotProductAttention  # noqa: F401
