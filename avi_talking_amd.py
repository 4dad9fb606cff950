"""Import alias: the product package lives in ``avi-talking_amd/`` (a name Python
cannot import directly); ``import avi_talking_amd`` loads that directory as a package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "avi-talking_amd")
_spec = importlib.util.spec_from_file_location(
    "avi_talking_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["avi_talking_amd"] = _mod
_spec.loader.exec_module(_mod)
