from .first_step import FirstStepTrainer, FlipViews, LossWeights  # noqa: F401
from .data_parallel import GradientAllReducer  # noqa: F401
