"""svpc_amd — MI355X-native hot path of awkrail/svpc (recurrent / state-aware transformer forward+backward)."""
from .model import RecursiveTransformer, StateAwareRecursiveTransformer  # noqa: F401
from .synthetic import ModelConfig, make_batch, make_config  # noqa: F401



def keep_host_copy(device_tensor, host_tensor):
    """Attach the host copy a loader already holds to the tensor it uploaded (``ingr_sep_masks``: the model derives the per-batch
    ingredient spans from it on the host).  With it the training step reads nothing back from the device; without it the model
    falls back to a synchronous ``.cpu()``.  Returns the device tensor."""
    device_tensor._svpc_host = host_tensor
    return device_tensor


__all__ = ["StateAwareRecursiveTransformer", "RecursiveTransformer", "ModelConfig", "make_config", "make_batch", "keep_host_copy"]
