"""Input pipeline at the config-2 shape: (32, 1500, 80) raw frames -> (32, 500, 240)."""
import os, sys, random, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import time_kernel_events
from oracle import ref_model as R
from asr_amd.utils.features import draw_spec_aug, featurize
dev = torch.device("cuda:0")
hp = R.make_hp(num_mask_F=1, num_mask_T=1, max_width_F=27, max_width_T=100, granularity=1)
B, Tm, F = 32, 1500, 80
lens = torch.linspace(1500, 900, B).long()
raw = torch.randn(B, Tm, F, device=dev)
mean, var = torch.randn(F), torch.rand(F) + 0.5
rng = random.Random(0)
masks = [draw_spec_aug(int(n), hp, rng) for n in lens]
fm, tm = [m[0] for m in masks], [m[1] for m in masks]
lens_d = lens.to(dev)
mean_d, var_d = mean.to(dev), var.to(dev)
from asr_amd.utils.features import masks_to_device
dt_all = time_kernel_events(lambda: featurize(raw, lens_d, 3, mean_d, var_d, fm, tm), 20)
fmd, tmd = masks_to_device(fm, tm, dev)
dt = time_kernel_events(lambda: featurize(raw, lens_d, 3, mean_d, var_d, fmd, tmd), 20)
nbytes = 2 * int(lens.sum()) * F * 4 + B * Tm * F * 4      # valid frames read twice, stacked batch written once
print(json.dumps({"featurize_us_with_mask_upload": dt_all * 1e6, "featurize_us_masks_resident": dt * 1e6,
                  "algorithmic_GBps_masks_resident": nbytes / dt / 1e9}))
