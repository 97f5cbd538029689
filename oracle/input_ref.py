"""CPU oracle for the device-side tail of the input pipeline (TEST INFRASTRUCTURE ONLY -- never imported by the product).

Restates what the reference's transform applies after PIL decoding / resizing (classification/data/transforms.py:225-253):
`T.RandomHorizontalFlip`, `T.RandomVerticalFlip` (as explicit per-sample flags), `T.ToTensor()`, `T.Normalize(mean, std)`
with IMAGENET_MEAN / IMAGENET_STD (transforms.py:16-17).

PARITY UNPINNED for this row: torchvision is not installed in the build container and the reference holds no golden
vectors for its transforms, so this file follows the published torchvision 0.10 semantics --
  functional.to_tensor : img.permute(2, 0, 1).contiguous().to(float32).div(255)
  functional.normalize : tensor.sub_(mean[:, None, None]).div_(std[:, None, None])
-- with the same torch CPU float32 operations in the same order.
"""
import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # transforms.py:16
IMAGENET_STD = (0.229, 0.224, 0.225)    # transforms.py:17


def to_tensor_normalize(frames_u8: torch.Tensor, flips=None, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
    """frames_u8: uint8 [B, H, W, 3] (HWC, RGB); flips: optional uint8 [B], bit 0 = horizontal, bit 1 = vertical.
    Returns float32 [B, 3, H, W]."""
    assert frames_u8.dtype == torch.uint8 and frames_u8.ndim == 4 and frames_u8.shape[-1] == 3
    x = frames_u8.cpu()
    if flips is not None:
        out = []
        for b in range(x.shape[0]):
            f = int(flips[b])
            img = x[b]
            if f & 1:
                img = img.flip(1)  # PIL FLIP_LEFT_RIGHT == reverse the W axis
            if f & 2:
                img = img.flip(0)  # FLIP_TOP_BOTTOM == reverse the H axis
            out.append(img)
        x = torch.stack(out)
    t = x.permute(0, 3, 1, 2).contiguous().to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32)[None, :, None, None]
    s = torch.tensor(std, dtype=torch.float32)[None, :, None, None]
    return t.sub_(m).div_(s)
