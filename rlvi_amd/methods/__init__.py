# Mirrors deep-learning/methods/__init__.py for the RLVI plug-in and the two small-loss baselines
# that share its "per-sample CE -> select/weight -> mean" shape.
from .train_rlvi import *  # noqa: F401,F403
from .train_rlvi import update_sample_weights, false_negative_criterion  # noqa: F401
from .train_usdnl import *  # noqa: F401,F403
from .train_coteaching import *  # noqa: F401,F403
