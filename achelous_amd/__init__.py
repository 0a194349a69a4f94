"""achelous_amd — MI355X-native (gfx950) forward path of the Achelous vision-radar perception network.

`Achelous` mirrors the reference `nets.Achelous.Achelous` (constructor, forward signature, outputs, state_dict
keys); the arithmetic runs in hand-written HIP kernels behind the C ABI of include/achelous.h."""
from .nets import Achelous, Achelous3T
from .postprocess import decode_outputs, non_max_suppression

__all__ = ['Achelous', 'Achelous3T', 'decode_outputs', 'non_max_suppression']
