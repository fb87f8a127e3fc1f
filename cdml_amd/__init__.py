"""Import alias for the package directory ``collaborative-deep-metric-learning_amd/``.

The directory name the project contract asks for is not a valid Python
identifier, so this two-line package puts it on its own ``__path__``: every
module is a single object reachable as ``cdml_amd.<module>``.
"""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                              "collaborative-deep-metric-learning_amd"))

from ._lib import CdmlError, lib_path, load_library  # noqa: E402,F401
