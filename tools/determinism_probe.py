#!/usr/bin/env python3
"""Is the training forward deterministic?  Runs the 12-step optimisation of the GPU test and, at every step, renders
the batch TWICE from the same random-generator state, with and without autograd, comparing every output bit by bit.
usage: python tools/determinism_probe.py [--rounds 6]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from nerf_loc_amd import synth, train, render  # noqa: E402
import test_gpu_train as tg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda:0")
rays = tg.T(synth.random_rays(512, seed=4), dev)
target = tg.T(np.random.RandomState(4).uniform(0, 1, size=(512, 3)).astype(np.float32), dev)
bad = 0
for rnd in range(a.rounds):
    nc, nf = tg.load_net(synth.make_state_dict(7, "A"), dev), tg.load_net(synth.make_state_dict(8, "A"), dev)
    opt = train.FusedAdam(list(nc.parameters()) + list(nf.parameters()), lr=5e-4, betas=(0.9, 0.999))
    kw = dict(network_query_fn=tg._query(dev), perturb=1.0, N_importance=128, network_fine=nf, N_samples=64, network_fn=nc,
              white_bkgd=True, raw_noise_std=0.0, retraw=True)
    for step in range(12):
        state = torch.cuda.get_rng_state(dev)
        outs = []
        for rep in range(3):
            torch.cuda.set_rng_state(state, dev)
            with torch.set_grad_enabled(rep < 2):
                ret = render.batchify_rays(rays, 1024 * 32, **kw)
            outs.append({k: v.detach().clone() for k, v in ret.items()})
        for rep, tag in ((1, "autograd again"), (2, "no autograd")):
            for k in outs[0]:
                x, y = outs[0][k], outs[rep][k]
                same = torch.equal(torch.nan_to_num(x), torch.nan_to_num(y))
                if not same and (rep == 1 or k not in ()):
                    d = (x.float() - y.float()).abs()
                    rows = (d.reshape(d.shape[0], -1).amax(1) > (0 if rep == 1 else 1e-4)).nonzero().flatten()
                    if len(rows):
                        bad += 1
                        print("round %d step %d: %s differs (%s): max %.3g, %d rays, first %s" %
                              (rnd, step, k, tag, float(d.max()), len(rows), rows[:12].tolist()), flush=True)
        torch.cuda.set_rng_state(state, dev)
        loss, _ = train.train_step(rays, target, {k: v for k, v in kw.items() if k != "retraw"}, opt, step + 1, 5e-4, 250)
    print("round %d done, last loss %.4f" % (rnd, float(loss)), flush=True)
print("differences:", bad)
