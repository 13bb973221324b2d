"""initializers.py:23-39 of the reference."""
import math

import numpy as np


class PriorProbability(object):
    """Bias initialiser: -log((1 - p) / p) so that the initial foreground probability is p."""

    def __init__(self, probability=0.01):
        self.probability = probability

    def get_config(self):
        return {"probability": self.probability}

    def __call__(self, shape, dtype=None):
        return np.ones(shape, dtype=dtype or np.float32) * -math.log((1 - self.probability) / self.probability)
