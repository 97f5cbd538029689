"""ssl4polyp_amd -- MI355X-native hot path (MAE pre-train + ViT-B/16 fine-tune) for irconde/SSL4POLYP.

Public surface mirrors the reference's model factories (src/ssl4polyp/utils/__init__.py:29-67,
src/ssl4polyp/models/mae/models_mae.py:223-250).
"""
from .models import (MaskedAutoencoderViT, ViT_from_MAE, VisionTransformer_from_Any, get_ImageNet_or_random_ViT,
                     get_MAE_backbone, mae_vit_base_patch16, mae_vit_huge_patch14, mae_vit_large_patch16, supervised_loss)
from .engine import reserve_streams
from .optim import FusedAdamW, LossScaler

__all__ = ["MaskedAutoencoderViT", "ViT_from_MAE", "VisionTransformer_from_Any", "get_MAE_backbone",
           "get_ImageNet_or_random_ViT", "mae_vit_base_patch16", "mae_vit_large_patch16", "mae_vit_huge_patch14",
           "supervised_loss", "reserve_streams", "FusedAdamW", "LossScaler"]
