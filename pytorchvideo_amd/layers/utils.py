"""Small helpers shared by the model builders (reference: pytorchvideo/layers/utils.py)."""
import math


def set_attributes(obj, params=None):
    """Copy every entry of `params` (usually `locals()`) except `self` onto `obj`
    (reference: layers/utils.py:7-16).  Assigning nn.Modules this way registers them as
    sub-modules under the argument's name, which is what fixes the state_dict keys."""
    for name, value in (params or {}).items():
        if name != "self":
            setattr(obj, name, value)


def round_width(width, multiplier, min_width=8, divisor=8, ceil=False):
    """Scale a channel count and snap it to a multiple of `divisor`
    (reference: layers/utils.py:19-39; X3D width expansion)."""
    if not multiplier:
        return width
    scaled = width * multiplier
    floor_width = min_width or divisor
    if ceil:
        snapped = int(math.ceil(scaled / divisor)) * divisor
    else:
        snapped = int(scaled + divisor / 2) // divisor * divisor
    snapped = max(floor_width, snapped)
    if snapped < 0.9 * scaled:  # never shrink by more than 10 %
        snapped += divisor
    return int(snapped)


def round_repeats(repeats, multiplier):
    """Scale a block count, rounding up (reference: layers/utils.py:42-49)."""
    return repeats if not multiplier else int(math.ceil(multiplier * repeats))
