from .convolutions import Conv2plus1d, ConvReduce3D, create_conv_2plus1d  # noqa
from .drop_path import DropPath  # noqa
from .squeeze_excitation import SqueezeExcitation  # noqa
from .swish import Swish  # noqa
from .roi_align import RoIAlign, roi_align  # noqa
from .utils import round_repeats, round_width, set_attributes  # noqa
from .attention import Mlp, MultiScaleAttention, MultiScaleBlock  # noqa
from .positional_encoding import SpatioTemporalClsPositionalEncoding  # noqa
