from oracle.nms import nms, batched_nms
