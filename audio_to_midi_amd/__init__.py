"""Importable alias of the ``audio-to-midi_amd/`` package directory.

The package directory carries the reference's name (with a hyphen), which Python cannot import
directly; this shim points ``audio_to_midi_amd``'s module search path at it and runs its
``__init__``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "audio-to-midi_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
