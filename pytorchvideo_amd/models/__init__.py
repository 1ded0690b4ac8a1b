from .head import create_res_basic_head, create_res_roi_pooling_head, create_vit_basic_head, ResNetBasicHead, ResNetRoIHead  # noqa
from .net import DetectionBBoxNetwork, MultiPathWayWithFuse, Net  # noqa
from .resnet import BottleneckBlock, create_bottleneck_block, create_resnet, create_resnet_with_roi_head  # noqa
from .stem import create_conv_patch_embed, create_res_basic_stem, ResNetBasicStem  # noqa
from .weight_init import init_net_weights  # noqa
from .x3d import create_x3d  # noqa
from .csn import create_csn  # noqa
from .r2plus1d import create_r2plus1d  # noqa
from .slowfast import create_slowfast, create_slowfast_with_roi_head  # noqa
from .vision_transformers import create_multiscale_vision_transformers  # noqa
