"""Exception types of neurite_amd."""

from ._lib import NeuriteAmdError  # noqa: F401


class InvalidArgumentError(ValueError):
    """
    Stand-in for tf.errors.InvalidArgumentError, which the reference's tf.debugging asserts raise
    (neurite/tf/metrics.py:441-444 'value outside range', :509 'metric not finite').
    """
