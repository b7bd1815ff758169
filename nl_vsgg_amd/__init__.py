"""Import alias.  The product lives in ``nl-vsgg_amd/`` (a hyphen is not a legal
Python identifier), so ``import nl_vsgg_amd`` resolves its sub-modules there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nl-vsgg_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
