"""Import-surface shim (oracle only). Nothing here computes anything on the pinned path:
PointNet++ arithmetic lives in un-vendored third-party wheels => parity unpinned there."""
from . import nn, transforms, data  # noqa: F401
