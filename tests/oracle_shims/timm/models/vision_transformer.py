from .layers import trunc_normal_
def _cfg(url='', **kwargs):
    return dict(url=url, **kwargs)
