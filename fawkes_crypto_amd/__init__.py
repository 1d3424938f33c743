"""Import shim: the product package lives in `fawkes-crypto_amd/` (the reference crate's name, which
is not a valid Python identifier).  `import fawkes_crypto_amd` executes that package under this name."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'fawkes-crypto_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
