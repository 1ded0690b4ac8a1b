"""Target device "mi355x": registers its transmuters on import, like the reference's
mobile_cpu package does (accelerator/deployment/mobile_cpu/transmuter/__init__.py:1-10)."""
from ..model_transmuter import EFFICIENT_BLOCK_TRANSMUTER_REGISTRY
from .blocks import EFFICIENT_BLOCK_TRANSMUTER_MI355X, Mi355xBlock  # noqa
from .conversion import convert_to_deployable_form  # noqa
from .session import Session  # noqa

EFFICIENT_BLOCK_TRANSMUTER_REGISTRY["mi355x"] = EFFICIENT_BLOCK_TRANSMUTER_MI355X
