"""Import-surface shim (oracle only): sentence splitting for the fixed hint template."""
from . import tokenize  # noqa: F401
