"""Synthetic COCO-shaped batches resident on the device: the reference's DALI / OpenCV
loaders (``ssds/dataset/*``) are out of scope; only their tensor contract matters --
images [B,3,H,W] float, targets [B,maxG,5] = (x, y, w, h, label) in absolute pixels, rows padded with -1
(dali_dataiterator.py:153-186, dataset_factory.py:29-35)."""
import torch


class SyntheticDetectionLoader(object):
    def __init__(self, batch_size, image_size, num_classes, steps, device, max_gt=32, seed=1234, dtype=torch.float32):
        self.batch_size, self.image_size, self.num_classes = batch_size, tuple(image_size), num_classes
        self.steps, self.device, self.max_gt, self.dtype = steps, device, max_gt, dtype
        self.gen = torch.Generator(device=device).manual_seed(seed)

    def __len__(self):
        return self.steps

    def batch(self):
        B, (H, W), G = self.batch_size, self.image_size, self.max_gt
        dev, g = self.device, self.gen
        images = torch.rand((B, 3, H, W), device=dev, generator=g).to(self.dtype)
        n = torch.randint(1, G + 1, (B, 1), device=dev, generator=g)
        x = torch.rand((B, G), device=dev, generator=g) * 0.8 * W
        y = torch.rand((B, G), device=dev, generator=g) * 0.8 * H
        w = (0.02 + 0.38 * torch.rand((B, G), device=dev, generator=g)) * W
        h = (0.02 + 0.38 * torch.rand((B, G), device=dev, generator=g)) * H
        w = torch.minimum(w, W - x)
        h = torch.minimum(h, H - y)
        label = torch.randint(0, self.num_classes, (B, G), device=dev, generator=g).float()
        t = torch.stack([x.floor(), y.floor(), w.ceil(), h.ceil(), label], 2)
        pad = torch.arange(G, device=dev).view(1, G) >= n
        t[pad] = -1
        return images, t

    def __iter__(self):
        for _ in range(self.steps):
            yield self.batch()
