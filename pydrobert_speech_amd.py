"""Import shim: ``import pydrobert_speech_amd`` -> the package in ``pydrobert-speech_amd/``.

The package directory carries the reference's distribution name plus ``_amd`` (a hyphen
is not importable), so this module replaces itself in ``sys.modules`` with the real
package object, loaded from that directory with its sub-module search path set.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "pydrobert-speech_amd")
_spec = _ilu.spec_from_file_location(
    __name__,
    _os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir],
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
