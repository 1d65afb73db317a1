"""Import shim: the package directory is `sa-gnn_amd/` (the repo layout contract), a name the
`import` statement cannot spell. This module makes it importable as `sa_gnn_amd`:
`import sa_gnn_amd.model` resolves to `sa-gnn_amd/model.py`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "sa-gnn_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f
