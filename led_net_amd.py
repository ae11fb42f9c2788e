"""Import shim: the product package lives in the directory `led-net_amd/`
(a hyphen is not importable), so `import led_net_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'led-net_amd')
_spec = importlib.util.spec_from_file_location(
    'led_net_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['led_net_amd'] = _mod
_spec.loader.exec_module(_mod)
