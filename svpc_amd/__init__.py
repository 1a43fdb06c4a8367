"""svpc_amd — MI355X-native hot path of awkrail/svpc (recurrent / state-aware transformer forward+backward)."""
from .model import RecursiveTransformer, StateAwareRecursiveTransformer  # noqa: F401
from .synthetic import ModelConfig, make_batch, make_config  # noqa: F401

__all__ = ["StateAwareRecursiveTransformer", "RecursiveTransformer", "ModelConfig", "make_config", "make_batch"]
