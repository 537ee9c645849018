"""Importable alias of the package directory `6dgs_amd` (a Python identifier cannot start with a digit)."""
import importlib as _importlib
import sys as _sys

_sys.modules[__name__] = _importlib.import_module("6dgs_amd")
