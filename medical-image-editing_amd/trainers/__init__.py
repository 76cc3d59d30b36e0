from .first_step import FirstStepTrainer, FlipViews, RandomTransformViews, LossWeights  # noqa: F401
from .data_parallel import GradientAllReducer  # noqa: F401
from .second_step import SecondStepTrainer, GanLossWeights  # noqa: F401
