"""minorseq_amd — MI355X-native juliet call+phase hot path.

The compute lives in csrc/ (hand-written HIP for gfx950 behind the C ABI of include/juliet_hip.h);
this package is the ctypes view of that ABI plus layout helpers.  There is no CPU fallback.
"""
__version__ = "0.1.0"
