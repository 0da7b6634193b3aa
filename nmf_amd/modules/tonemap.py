"""sRGB tonemap (reference: modules/tonemap.py:34-55)."""
import torch


class SRGBTonemap(torch.nn.Module):
    def forward(self, img, noclip=False):
        limit = 0.0031308
        out = torch.where(img > limit, 1.055 * (img.clip(min=limit) ** (1.0 / 2.4)) - 0.055, 12.92 * img)
        return out if noclip else out.clip(0, 1)

    def inverse(self, img):
        limit = 0.04045
        return torch.where(img > limit, torch.pow((img + 0.055) / 1.055, 2.4), img / 12.92)
