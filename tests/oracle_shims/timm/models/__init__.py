from .registry import register_model
