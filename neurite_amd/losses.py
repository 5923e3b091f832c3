"""
neurite_amd.losses -- neurite/tf/losses.py:46-205 for the hot path: the metrics classes with
`loss` (negative Dice [B, L] / the CCE scalar) and `mean_loss` (negative mean Dice).
"""

import numpy as np

from . import metrics

__all__ = ['Dice', 'SoftDice', 'HardDice', 'CategoricalCrossentropy', 'WeightedCategoricalCrossentropy',
           'MeanSquaredErrorProb', 'multiple_losses_decorator']


class _DiceLossMixin:
    def loss(self, y_true, y_pred):
        """dice loss (negative Dice score), [batch_size, nb_labels] (neurite/tf/losses.py:68-80)."""
        return - self.dice(y_true, y_pred)

    def mean_loss(self, y_true, y_pred):
        """negative mean dice, scalar (neurite/tf/losses.py:82-95)."""
        return - self.mean_dice(y_true, y_pred)


class Dice(_DiceLossMixin, metrics.Dice):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)


class SoftDice(_DiceLossMixin, metrics.SoftDice):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)


class HardDice(_DiceLossMixin, metrics.HardDice):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)


class CategoricalCrossentropy(metrics.CategoricalCrossentropy):
    """neurite/tf/losses.py:193-205."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    def loss(self, *args, **kwargs):
        return self.cce(*args, **kwargs)


WeightedCategoricalCrossentropy = CategoricalCrossentropy


class MeanSquaredErrorProb(metrics.MeanSquaredErrorProb):
    """neurite/tf/losses.py:208-220."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    def loss(self, *args, **kwargs):
        return self.mse(*args, **kwargs)


def multiple_losses_decorator(losses, weights=None):
    """neurite/tf/losses.py:225-246."""
    if weights is None:
        weights = np.ones(len(losses))

    def loss(y_true, y_pred):
        total_val = 0
        for idx, los in enumerate(losses):
            total_val += weights[idx] * los(y_true, y_pred)
        return total_val

    return loss
