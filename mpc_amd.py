"""Import shim: `import mpc_amd` == the package directory `motion-planning-for-autonomous-driving-with-mpc_amd/`
(whose hyphenated name is not a Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("motion-planning-for-autonomous-driving-with-mpc_amd")
sys.modules[__name__] = _pkg
