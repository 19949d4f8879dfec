"""Training ray sampler: the two ways run_nerf.py picks a batch of rays and target colours per iteration
(run_nerf.py:685-768; the same code in run_multi_nerf.py).

  use_batching=True   all rays of all training images are built once on the host with get_rays_np, shuffled with
                      np.random.shuffle, moved to the device and consumed N_rand at a time; after an epoch the bank
                      is re-shuffled with torch.randperm.
  use_batching=False  each iteration draws one training image (np.random.choice), builds its rays on the device,
                      optionally restricts them to a centre crop for the first precrop_iters iterations, and picks
                      N_rand pixels without replacement (np.random.choice).

The random draws are the reference's own numpy / torch calls in the reference's order, so a run seeded like the
reference selects the same rays.  `next(i)` returns (batch_rays [2, N_rand, 3], target_s [N_rand, 3]).
"""
import numpy as np
import torch

from .helpers import get_rays, get_rays_np


class RayBank:
    def __init__(self, images, poses, H, W, K, i_train, N_rand, use_batching=True, precrop_iters=0, precrop_frac=0.5,
                 device="cuda"):
        self.H, self.W, self.K = int(H), int(W), K
        self.N_rand = N_rand
        self.use_batching = bool(use_batching)
        self.precrop_iters, self.precrop_frac = precrop_iters, precrop_frac
        self.i_train = np.asarray(i_train)
        self.device = torch.device(device)
        poses = np.asarray(poses, dtype=np.float32)
        if self.use_batching:
            rays = np.stack([get_rays_np(self.H, self.W, K, p) for p in poses[:, :3, :4]], 0)   # [N, ro+rd, H, W, 3]
            rays_rgb = np.concatenate([rays, np.asarray(images)[:, None]], 1)                    # [N, ro+rd+rgb, H, W, 3]
            rays_rgb = np.transpose(rays_rgb, [0, 2, 3, 1, 4])
            rays_rgb = np.stack([rays_rgb[i] for i in self.i_train], 0)                          # train images only
            rays_rgb = np.reshape(rays_rgb, [-1, 3, 3]).astype(np.float32)
            np.random.shuffle(rays_rgb)
            self.rays_rgb = torch.from_numpy(rays_rgb).to(self.device)
            self.i_batch = 0
        else:
            self.images = images       # stays on the host; one image per iteration is uploaded
        self.poses = torch.from_numpy(poses).to(self.device)

    def __len__(self):
        return self.rays_rgb.shape[0] if self.use_batching else len(self.i_train) * self.H * self.W

    def next(self, i=0):
        """i: the global iteration number (only the centre-crop schedule depends on it)."""
        if self.use_batching:
            batch = self.rays_rgb[self.i_batch:self.i_batch + self.N_rand]
            batch = torch.transpose(batch, 0, 1)
            batch_rays, target_s = batch[:2], batch[2]
            self.i_batch += self.N_rand
            if self.i_batch >= self.rays_rgb.shape[0]:
                rand_idx = torch.randperm(self.rays_rgb.shape[0])
                self.rays_rgb = self.rays_rgb[rand_idx.to(self.device)]
                self.i_batch = 0
            return batch_rays, target_s
        H, W = self.H, self.W
        img_i = np.random.choice(self.i_train)
        target = torch.as_tensor(np.asarray(self.images[img_i]), dtype=torch.float32).to(self.device)
        rays_o, rays_d = get_rays(H, W, self.K, self.poses[img_i, :3, :4])
        if self.N_rand is None:
            return torch.stack([rays_o, rays_d], 0), target
        if i < self.precrop_iters:
            dH = int(H // 2 * self.precrop_frac)
            dW = int(W // 2 * self.precrop_frac)
            rows = torch.arange(H // 2 - dH, H // 2 + dH)
            cols = torch.arange(W // 2 - dW, W // 2 + dW)
        else:
            rows, cols = torch.arange(H), torch.arange(W)
        n = rows.numel() * cols.numel()
        select_inds = torch.from_numpy(np.random.choice(n, size=[self.N_rand], replace=False))
        r = rows[select_inds // cols.numel()].to(self.device)
        c = cols[select_inds % cols.numel()].to(self.device)
        return torch.stack([rays_o[r, c], rays_d[r, c]], 0), target[r, c]
