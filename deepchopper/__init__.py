"""Import-name compatibility: `import deepchopper` served by the MI355X-native package.

A caller of the reference writes `import deepchopper`, `deepchopper.DeepChopper.from_pretrained(...)`,
`from deepchopper.deepchopper import majority_voting` (the PyO3 module, src/python.rs:879-958) or
`python -m deepchopper predict ...` (deepchopper/__main__.py).  This package maps exactly those names onto
`deepchopper_amd` -- the hot-path subset of the reference's surface, nothing else: training, evaluation
and the data modules of the reference are not part of this build and raise AttributeError / ImportError as usual.

It contains no logic of its own; install it only where the reference package is not installed.
"""
import sys as _sys
import types as _types

import deepchopper_amd as _impl
from deepchopper_amd import DeepChopper, __version__, cli, rs_api as _rs_api, ui
from deepchopper_amd.rs_api import *  # noqa: F401,F403 - what `from .deepchopper import *` gives in the reference

# `deepchopper.deepchopper`: the compiled module's name in the reference
deepchopper = _rs_api
_sys.modules[__name__ + ".deepchopper"] = _rs_api
_sys.modules[__name__ + ".cli"] = cli
_sys.modules[__name__ + ".ui"] = ui

# `deepchopper.models.DeepChopper` (deepchopper/models/__init__.py -> dc_hg.py:120-163)
models = _types.ModuleType(__name__ + ".models", "DeepChopper model factory (see deepchopper_amd.model)")
models.DeepChopper = DeepChopper
_sys.modules[models.__name__] = models

__all__ = ["DeepChopper", "__version__", "cli", "models", "ui", *_rs_api.__all__]
