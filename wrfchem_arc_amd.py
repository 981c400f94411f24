"""Import alias: the package directory is named ``wrfchem-arc-interactions_amd``
(a hyphen is not a valid identifier), so ``import wrfchem_arc_amd as wa`` loads
it through importlib and re-exports it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
pkg = importlib.import_module("wrfchem-arc-interactions_amd")
sys.modules[__name__] = pkg
