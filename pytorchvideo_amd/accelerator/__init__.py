from .efficient_blocks import EfficientBlockBase, NoOpConvertBlock  # noqa
from .model_transmuter import EFFICIENT_BLOCK_TRANSMUTER_REGISTRY, transmute_model  # noqa
from . import mi355x  # noqa  (registers the "mi355x" target)
from .mi355x.conversion import convert_to_deployable_form  # noqa
