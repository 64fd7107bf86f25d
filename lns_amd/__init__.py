"""Importable alias for the package directory `lns-latent-neural-pde-solver_amd/`.

The product package directory carries the repository's name, which is not a
valid Python identifier; this alias points `lns_amd.__path__` at it so that
`import lns_amd.modules.autoencoder2d` etc. resolve there.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "lns-latent-neural-pde-solver_amd")
__path__ = [_PKG_DIR]
with open(_os.path.join(_PKG_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))
