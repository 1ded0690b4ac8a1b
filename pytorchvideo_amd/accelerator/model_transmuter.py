"""Device-plugin registry and the recursive transmuter (reference:
pytorchvideo/accelerator/deployment/common/model_transmuter.py:16-86).

`EFFICIENT_BLOCK_TRANSMUTER_REGISTRY[device]` is a list of callables
`nn.Module -> Optional[nn.Module]`; `transmute_model` walks the children of `model`, asks
every transmuter of the target device, installs the first non-None answer in place and
recurses into children nobody claimed.  A transmuter declines by returning None.

Inside a PyTorchVideo installation the registry below IS the reference's dict: importing
`pytorchvideo_amd.accelerator` registers the "mi355x" target there, and the reference's own
`transmute_model(model, target_device="mi355x")` works unchanged.
"""
import logging

import torch.nn as nn

try:
    from pytorchvideo.accelerator.deployment.common.model_transmuter import EFFICIENT_BLOCK_TRANSMUTER_REGISTRY
except ImportError:
    EFFICIENT_BLOCK_TRANSMUTER_REGISTRY = {}


def _first_match(module, transmuters, where):
    hits = [m for m in (t(module) for t in transmuters) if m is not None]
    if len(hits) > 1:
        logging.warning("%s has multiple matches: %s; using %s (highest priority)", where,
                        [type(h).__name__ for h in hits], type(hits[0]).__name__)
    return hits[0] if hits else None


def transmute_model(model: nn.Module, target_device: str = "mi355x", prefix: str = ""):
    assert target_device in EFFICIENT_BLOCK_TRANSMUTER_REGISTRY, (
        f"{target_device} not registered in EFFICIENT_BLOCK_TRANSMUTER_REGISTRY!")
    transmuters = EFFICIENT_BLOCK_TRANSMUTER_REGISTRY[target_device]
    for name, child in model.named_children():
        where = f"{prefix}.{name}"
        replacement = _first_match(child, transmuters, where)
        if replacement is not None:
            model._modules[name] = replacement
            logging.info("Replacing %s (%s) with %s", where, type(child).__name__, type(replacement).__name__)
        else:
            transmute_model(child, target_device=target_device, prefix=where)
