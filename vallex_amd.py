"""Import alias: `import vallex_amd` loads the package that lives in ./vall-e-x_amd/ (not a valid identifier)."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vall-e-x_amd")
_spec = importlib.util.spec_from_file_location("vallex_amd", os.path.join(_d, "__init__.py"),
                                               submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vallex_amd"] = _mod
_spec.loader.exec_module(_mod)
