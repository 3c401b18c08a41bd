"""Import alias: the package lives in ``rvdd-release_amd/`` (the layout the
project prescribes); a hyphen is not importable, so this stub points the
importable name ``rvdd_release_amd`` at that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "rvdd-release_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
