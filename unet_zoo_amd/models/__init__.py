"""Model registry with the reference's public surface (unet_zoo/models/__init__.py:27-238):
``create_model(name, pretrained=False, **kwargs)``, ``list_models()``, ``get_model_config(name)``.

Same 24 names, same case-insensitive lookup, same kwarg adaptation and error types
(``ValueError`` for an unknown name or a missing ``image_size``, ``TypeError`` for stray
kwargs, raised by the constructor).  Names whose graph is not yet on the HIP engine raise
``NotImplementedError`` — there is deliberately no PyTorch fallback backend in the product path
(SURVEY.md §8 scope: the hot path only).
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional

import torch.nn as nn

from .unet import UNet
from .attention_unet import AttentionUNet
from .u2net import U2NET, U2NETP
from .swin_unet_v2 import SwinTransformerSys
from .nested_unet import NestedUNet
from .resunet import ResUnet
from .missformer import MISSFormer
from .transatt_unet import TransAttUNet
from .unet_transformer import U_Transformer
from .multiresunet import MultiResUnet
from .uctransnet import UCTransNet, get_uctransnet_config

# every name the reference registers (models/__init__.py:27-52); value = constructor or None
_model_entries: Dict[str, Optional[Callable[..., nn.Module]]] = {
    'unet': UNet,
    'attention_unet': AttentionUNet,
    'transatt_unet': TransAttUNet,
    'raunet': None,
    'da_transformer': None,
    'unet_transformer': U_Transformer,
    'uctransnet': UCTransNet,
    'multiresunet': MultiResUnet,
    'nested_unet': NestedUNet,
    'missformer': MISSFormer,
    'vnet': None,
    'u2net': U2NET,
    'u2netp': U2NETP,
    'swin_unet_v2': SwinTransformerSys,
    'resunet': ResUnet,
    'wranet': None,
    'egeunet': None,
    'unext': None,
    'unext_s': None,
    'mmunet': None,
    'axialunet': None,
    'gated': None,
    'medt': None,
    'logo': None,
}

_config_functions: Dict[str, Callable[..., Dict[str, Any]]] = {'uctransnet': get_uctransnet_config}


def list_models() -> List[str]:
    """All registered model names, sorted (reference: models/__init__.py:59-61)."""
    return sorted(_model_entries.keys())


def hip_models() -> List[str]:
    """The subset of :func:`list_models` that runs on the HIP engine today."""
    return sorted(k for k, v in _model_entries.items() if v is not None)


def get_model_config(model_name: str, **kwargs) -> Dict[str, Any]:
    """Default configuration of a model, ``{}`` if it has none (models/__init__.py:63-76)."""
    if model_name in _config_functions:
        return _config_functions[model_name](**kwargs)
    return {}


def create_model(model_name: str, pretrained: bool = False, **kwargs) -> nn.Module:
    """Instantiate a UNet variant by name (reference: models/__init__.py:78-238)."""
    name = model_name.lower()
    if name not in _model_entries:
        raise ValueError(f"Unknown model: '{model_name}'. Available models: {list_models()}")
    factory = _model_entries[name]

    in_channels = kwargs.pop('in_channels', 3)
    num_classes = kwargs.pop('num_classes', 1)
    image_size = kwargs.pop('image_size', None)
    depth = kwargs.pop('depth', 5)

    if name in ('swin_unet_v2', 'uctransnet') and image_size is None:
        raise ValueError(f"Model '{model_name}' requires 'image_size' parameter in config.")

    if factory is None:
        raise NotImplementedError(
            f"'{name}' is registered (the reference's name list is kept) but its graph is not yet "
            f"implemented on the MI355X HIP engine; available today: {hip_models()}")

    args: Dict[str, Any] = {}
    if name == 'unet':
        # `depth` is accepted and dropped, as in the reference (models/__init__.py:97-99)
        args.update(in_channels=in_channels, num_classes=num_classes)
    elif name == 'attention_unet':
        args.update(in_channels=in_channels, num_classes=num_classes, depth=depth)
    elif name in ('u2net', 'u2netp'):
        args.update(in_ch=in_channels, out_ch=num_classes)
    elif name == 'swin_unet_v2':
        args.update(img_size=image_size, in_chans=in_channels, num_classes=num_classes)
    elif name == 'resunet':
        # models/__init__.py:167-170
        args.update(in_channels=in_channels, num_classes=num_classes, filters=kwargs.pop('filters', [64, 128, 256, 512]))
    elif name == 'missformer':
        # models/__init__.py:145-148: `image_size` is popped above and NOT forwarded, so the reference always builds
        # MISSFormer for its default 512x512 (missformer.py:868); mirrored.  `depth` is absorbed by **kwargs there.
        # Build the class directly (`MISSFormer(image_size=...)`) for another input size.
        args.update(in_channels=in_channels, num_classes=num_classes, depth=depth)
    elif name == 'uctransnet':
        # models/__init__.py:125-132, :225-226: the built-in config, img_size from image_size, `vis` from the kwargs
        args.update(config=get_uctransnet_config(), in_channels=in_channels, num_classes=num_classes,
                    img_size=image_size, vis=kwargs.pop('vis', False))
    elif name == 'multiresunet':
        # models/__init__.py:134-137: `depth` travels to the constructor (absorbed by **kwargs there)
        args.update(in_channels=in_channels, num_classes=num_classes, depth=depth)
    elif name == 'transatt_unet':
        # models/__init__.py:104-107: `depth` travels to the constructor (absorbed by **kwargs there)
        args.update(in_channels=in_channels, num_classes=num_classes, depth=depth)
    elif name == 'nested_unet':
        # models/__init__.py:139-143: depth travels to the constructor (absorbed by **kwargs there)
        args.update(in_channels=in_channels, num_classes=num_classes, depth=depth,
                    deep_supervision=kwargs.pop('deep_supervision', False))
    else:
        args.update(in_channels=in_channels, num_classes=num_classes)
    args.update(kwargs)  # leftovers reach the constructor: unknown ones raise TypeError there

    model = factory(**args)
    if pretrained:
        print(f"Warning: Pre-trained weights for {model_name} are not yet implemented.")
    return model


__all__ = ['UNet', 'AttentionUNet', 'U2NET', 'U2NETP', 'SwinTransformerSys', 'NestedUNet', 'ResUnet', 'MISSFormer', 'TransAttUNet', 'U_Transformer', 'MultiResUnet', 'UCTransNet', 'list_models', 'hip_models', 'get_model_config', 'create_model']
