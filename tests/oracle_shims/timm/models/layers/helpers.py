from . import to_2tuple
