from .first_step import FirstStepTrainer, FlipViews, RandomTransformViews, LossWeights  # noqa: F401
from .data_parallel import GradientAllReducer  # noqa: F401
from .second_step import SecondStepTrainer, GanLossWeights  # noqa: F401
from .config import (build_first_step_trainer, configure_models, configure_optimizers, configure_losses,  # noqa: F401
                     loss_weights, set_transform)
