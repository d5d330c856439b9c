"""Import alias: `import dispu_amd` loads the package that lives in ./dis-pu_amd/ (the directory
name carries a hyphen, which is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dis-pu_amd")
_spec = importlib.util.spec_from_file_location("dispu_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dispu_amd"] = _mod
_spec.loader.exec_module(_mod)
