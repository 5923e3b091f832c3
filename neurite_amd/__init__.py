"""
neurite_amd -- MI355X-native (gfx950 / CDNA4) implementation of adalca/neurite's 3-D volume hot path:
utils.interpn / SpatialTransformer / Resize, metrics.Dice and label-weighted categorical
cross-entropy (more rows of SURVEY.md section 8 follow), behind the reference's own call signatures.

    import neurite_amd as ne
    warped = ne.layers.SpatialTransformer()([moving, flow])     # [B, X, Y, Z, C] ROCm tensors
    d = ne.metrics.Dice().dice(fixed, warped)                   # [B, L]

This is the `neurite.torch` backend that neurite/__init__.py:33-42 selects with
NEURITE_BACKEND=pytorch and that the reference does not ship.  PyTorch-ROCm provides device memory
and streams; every voxel-sized computation is a hand-written HIP kernel reached through the C ABI
in include/neurite_amd.h.  There is no CPU path.
"""

__version__ = '0.1.0'

from . import _lib  # noqa: F401
from . import errors  # noqa: F401
from . import deferred  # noqa: F401
from . import checked  # noqa: F401
from . import utils  # noqa: F401
from . import layers  # noqa: F401
from . import metrics  # noqa: F401
from . import losses  # noqa: F401
from . import models  # noqa: F401
from . import fused  # noqa: F401
from . import distributed  # noqa: F401
from . import augment  # noqa: F401
from . import synthesis  # noqa: F401
from . import synth  # noqa: F401

backend = 'pytorch'


def library_path():
    """Path of the HIP shared library this package drives."""
    return _lib.LIB_PATH
