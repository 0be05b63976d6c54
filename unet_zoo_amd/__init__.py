"""unet_zoo_amd — MI355X-native (gfx950) engine for the unet_zoo encoder/decoder hot path.

Drop-in surface: ``create_model`` / ``list_models`` as in ``unet_zoo`` (reference
unet_zoo/__init__.py:1).  The arithmetic runs in hand-written HIP kernels behind the C ABI of
``libunetzoo_hip.so`` (include/unetzoo_hip.h); there is no CPU or PyTorch fallback.
"""
from .models import create_model, list_models, get_model_config, hip_models
from .graph import set_default_dtype, get_default_dtype
from .step import GraphedStep
from .config import Config, load_config

__version__ = "0.1.0"
__all__ = ["create_model", "list_models", "get_model_config", "hip_models", "set_default_dtype",
           "get_default_dtype", "GraphedStep", "Config", "load_config"]
