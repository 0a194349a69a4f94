"""Stand-in for `torchvision` exposing only `ops.deform_conv2d`, `ops.nms`, `ops.boxes.batched_nms`,
each forwarding to OUR restatement under /root/repo/oracle (torchvision 0.12.0 semantics)."""
from . import ops
