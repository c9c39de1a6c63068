"""Seeded synthetic inputs of the SuRF hot path: ring cameras, procedural images and the
analytic sphere pyramid used by bench.py, __graft_entry__.smoke() and the full-size property tests.
Everything is built with torch on the given device; nothing here is part of the product path."""
import math

import torch

AZIMUTHS = {3: [0.0, 0.25, -0.25], 5: [0.0, 0.25, -0.25, 0.5, -0.5], 7: [0.0, 0.2, -0.2, 0.4, -0.4, 0.6, -0.6]}


def ring_cameras(nv, H, W, radius=2.5):
    """nv pinhole cameras on a ring looking at the origin, fx = fy = 1.6 W, principal point at the centre;
    near/far as datasets/dtu.py:358-362: 0.95 (|o| - 1), 1.05 (|o| + 1)."""
    c2ws, intrs = [], []
    for a in AZIMUTHS[nv]:
        o = torch.tensor([radius * math.sin(a), 0.0, -radius * math.cos(a)], dtype=torch.float32)
        z = -o / o.norm()
        x = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0]), z)
        x = x / x.norm()
        y = torch.linalg.cross(z, x)
        c2w = torch.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = x, y, z, o
        K = torch.eye(4)
        K[0, 0] = K[1, 1] = 1.6 * W
        K[0, 2], K[1, 2] = (W - 1) / 2, (H - 1) / 2
        c2ws.append(c2w)
        intrs.append(K)
    c2ws, intrs = torch.stack(c2ws), torch.stack(intrs)
    dist = c2ws[:, :3, 3].norm(dim=1)
    near_fars = torch.stack([0.95 * (dist - 1), 1.05 * (dist + 1)], dim=1)
    return intrs, c2ws, near_fars


def procedural_images(nv, H, W, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    imgs = torch.stack([torch.stack([0.5 + 0.45 * torch.sin(0.031 * (c + 1) * xx + 0.017 * (v + 1) * yy + c)
                                     for c in range(3)]) for v in range(nv)])
    imgs = (imgs + 0.05 * torch.rand(nv, 3, H, W, generator=g)).clamp(0, 0.999)
    return imgs.to(device)


def pixel_rays(intr, c2w, H, W, step=1, device="cpu"):
    """All pixel rays of one view, row-major (datasets/dtu.py:428-433 convention)."""
    ys, xs = torch.meshgrid(torch.arange(0, H, step, dtype=torch.float32), torch.arange(0, W, step, dtype=torch.float32),
                            indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(xs.numel())], dim=-1)
    d = pix @ torch.inverse(intr)[:3, :3].t()
    d = d / d.norm(dim=-1, keepdim=True)
    rays_d = (d @ c2w[:3, :3].t()).contiguous()
    rays_o = c2w[:3, 3][None].expand_as(rays_d).contiguous()
    return rays_o.to(device), rays_d.to(device)


def sphere_pyramid(base_dim, device, bands=(float("inf"), 0.92, 0.23, 0.023), seed=0, slab=32):
    """Analytic stand-in for the 4-stage sparse pyramid around the sphere r = 0.5:
    stage s keeps voxels of the D_s = base * 2^s grid with | |x| - 0.5 | < bands[s]; features N(0, 0.1^2);
    matching logits -20 | |x| - 0.5 | on the finest dense grid.
    Returns (volumes[(N_s,8)], tables[(D,D,D) int32], matching_volume (D3,D3,D3)), coarse -> fine."""
    vols, tabs = [], []
    D = base_dim
    mvol = None
    for s, band in enumerate(bands):
        ax = (torch.arange(D, dtype=torch.float32, device=device) * (2.0 / (D - 1)) - 1.0)
        table = torch.full((D, D, D), -1, dtype=torch.int32, device=device)
        last = s == len(bands) - 1
        if last:
            mvol = torch.empty(D, D, D, dtype=torch.float32, device=device)
        count = 0
        yz2 = ax[None, :, None] ** 2 + ax[None, None, :] ** 2
        for x0 in range(0, D, slab):
            r = torch.sqrt(ax[x0:x0 + slab, None, None] ** 2 + yz2)
            dist = (r - 0.5).abs()
            if last:
                mvol[x0:x0 + slab] = -20.0 * dist
            keep = dist < band
            n = int(keep.sum())
            sub = table[x0:x0 + slab]
            sub[keep] = torch.arange(count, count + n, dtype=torch.int32, device=device)
            count += n
        g = torch.Generator().manual_seed(seed * 16 + s)
        # features are generated in chunks on the host generator for reproducibility across devices
        f = torch.zeros(count, 8, dtype=torch.float32, device=device)
        for c0 in range(0, count, 1 << 22):
            n = min(1 << 22, count - c0)
            f[c0:c0 + n, :7] = (torch.randn(n, 7, generator=g) * 0.1).to(device)
        vols.append(f)
        tabs.append(table)
        D *= 2
    return vols, tabs, mvol


def feature_pyramid(nv, H, W, seed=0, device="cpu"):
    """Random 4-level, 4-channel image-feature pyramid, fine -> coarse (stand-in for the FPN output)."""
    g = torch.Generator().manual_seed(1000 + seed)
    return [torch.randn(nv, 4, H >> l, W >> l, generator=g).to(device) for l in range(4)]


def sphere_logit(coords, D):
    """Analytic matching logit of the synthetic scene, -20 | |x| - 0.5 | at the voxel centres of a D^3 lattice: what the
    (untrained) sparse U-Net's first output channel is replaced by in the full-size volume-build bench / tests."""
    world = coords.float() * (2.0 / (D - 1)) - 1.0
    return -20.0 * (world.norm(dim=1) - 0.5).abs()
