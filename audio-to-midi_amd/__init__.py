"""MI355X-native hot path of kasper0406/audio-to-midi: raw audio -> CNN frontend -> RoPE
transformer -> per-frame event logits (+ BCE loss and gradients), hand-written HIP for gfx950
behind a C ABI (include/amt.h), with a Python host that mirrors the reference's model API."""
from . import lib, params, rope, synth  # noqa: F401
from .model import OutputSequenceGenerator, model_config, StepOutput  # noqa: F401
from .rope import RopeFreqs, precompute_frequencies  # noqa: F401
