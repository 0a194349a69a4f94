from . import boxes
from .boxes import nms, batched_nms
from oracle.deform_conv import deform_conv2d
