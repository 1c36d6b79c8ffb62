# Mirrors deep-learning/methods/__init__.py:1 for the RLVI plug-in only.
from .train_rlvi import *  # noqa: F401,F403
from .train_rlvi import update_sample_weights, false_negative_criterion  # noqa: F401
