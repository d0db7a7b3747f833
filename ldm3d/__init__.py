"""Importable alias for the package directory ``3d-latent-diffusion-model_amd/`` (its name is not a
valid Python identifier).  ``import ldm3d.networks`` resolves to ``3d-latent-diffusion-model_amd/networks.py``."""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "3d-latent-diffusion-model_amd")
__path__ = [_PKG_DIR]
with open(_os.path.join(_PKG_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))
