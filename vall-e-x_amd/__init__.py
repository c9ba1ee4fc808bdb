"""vall-e-x_amd: MI355X-native drop-in for the VALL-E X inference hot path.

Mirrors the reference's Python surface for that path and nothing else:
    utils.generation.preload_models / generate_audio / generate_audio_from_long_text   (utils/generation.py:50,92,155)
    models.vallex.VALLE(...).inference(...)                                            (models/vallex.py:405,458)
over the C ABI in include/vallex_hip.h (libvallex_hip.so, hand-written gfx950 kernels).
Import as `import vallex_amd` (see /vallex_amd.py: the directory name is not a Python identifier).
"""
from . import macros  # noqa: F401
from ._capi import Batch, Engine, VallexHipError, load_library  # noqa: F401
from .macros import SAMPLE_RATE  # noqa: F401
