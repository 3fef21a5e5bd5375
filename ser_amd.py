"""Import shim: the package directory name required by the build contract
(`multilingual-multimodal-speech-emotion-recognition_amd/`) is not a valid Python
identifier, so `import ser_amd` loads that directory as the package `ser_amd`."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multilingual-multimodal-speech-emotion-recognition_amd")
_spec = importlib.util.spec_from_file_location("ser_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ser_amd"] = _mod
_spec.loader.exec_module(_mod)
