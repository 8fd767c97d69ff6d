"""Import shim: ``import salt_amd`` loads the package that lives in the directory
``open-solution-salt-identification_amd/`` (the repo's literal name contains hyphens, which Python
cannot spell in an import statement).  Sub-modules import as ``salt_amd.<name>``."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'open-solution-salt-identification_amd')
_spec = importlib.util.spec_from_file_location('salt_amd', os.path.join(_DIR, '__init__.py'),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['salt_amd'] = _mod
_spec.loader.exec_module(_mod)
