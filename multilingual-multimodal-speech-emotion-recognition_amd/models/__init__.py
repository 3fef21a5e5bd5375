"""Mirror of the reference's `src/models/__init__.py` exports (plus the classes train.py imports by path)."""
from .audio_encoder import AudioEncoder
from .text_encoder import TextEncoder
from .fusion import FusionLayer
from .classifier import Classifier

GatedFusion = FusionLayer          # the name BASELINE.json's north star uses for the same module
from .cross_attention import CrossModalAttention as CrossAttention  # noqa: E402

__all__ = ["AudioEncoder", "TextEncoder", "FusionLayer", "Classifier", "GatedFusion", "CrossAttention"]
