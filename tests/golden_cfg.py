"""Configuration and helpers shared by the golden-fixture tests (mirrors make_golden.py's MODEL_CONF)."""
import torch

CFG = {
    "range_ratios": [1.0, 0.4, 0.1, 0.01],
    "base_volume_dim": 8,
    "n_samples_depths": [128, 64, 32, 16],
    "depth_res_levels": [4, 2, 2, 1],
    "n_samples": [64, 32, 16, 16],
    "sample_ranges": [1.0, 0.4, 0.1, 0.01],
    "n_depth": 256,
}


def stub_regnet(feats, coords, D, stage):
    """Same deterministic stand-in regulariser as tests/golden/make_golden.py (not a reference function)."""
    g = torch.Generator().manual_seed(100 + stage)
    A = torch.randn(feats.shape[1], 8, generator=g) * 0.5
    B = torch.randn(feats.shape[1], 8, generator=g) * 0.5
    world = coords * (2.0 / (D - 1)) - 1.0
    out = torch.tanh(feats @ A)
    out[:, 0] = -20.0 * (world.norm(dim=1) - 0.5).abs() + 0.5 * out[:, 0]
    return out, torch.tanh(feats @ B)


def pipeline_views(gp):
    """Per-stage sparse volumes / index tables / masks (fine -> coarse, as surf.py:159 passes them)
    and the final dense matching volume, rebuilt from the pipeline fixture."""
    vols, tabs, masks = [], [], []
    for s in range(4):
        out = gp[f"s{s}_reg_out"]
        table = gp[f"s{s}_table"].long()
        vols.append(out[:, 1:].contiguous())
        tabs.append(table)
        masks.append((table >= 0).float())
    return vols[::-1], tabs[::-1], masks[::-1], gp["s3_mvol"]
