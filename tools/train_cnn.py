#!/usr/bin/env python3
"""Train the stone classifier (NNManager.create_net architecture, nn_manager.py:277-298) on the
synthetic renderer and write tools/out/cnn_weights.npz in Keras-1 'tf' layout
(tools/make_h5_fixtures.py turns it into camkifu_amd/data/keras.h5 with the real h5py).

The reference's trained keras.h5 cannot be fetched here (cvconf.py:58, no network), so labels
produced with seeded random weights are meaningless.  This script makes them meaningful:
boards are rendered directly in the canonical 380x380 frame (corners jittered by a few pixels,
as a detected transform would be), cut into the 100 patches of NNManager._get_x and labelled
with NNManager.compute_label.  CPU only (torch), a few minutes on 8 cores.
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from camkifu_amd import synth  # noqa: E402

ORIG = [0, 40, 80, 120, 160, 200, 240, 280, 320, 340]


def make_boards(n, seed):
    rng = np.random.default_rng(seed)
    X, Y = [], []
    for b in range(n):
        dens = rng.uniform(0.0, 0.65)
        stones = synth.random_stones(rng, density=dens, keep_first_line_empty=(b % 3 == 0))
        corners = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32) + rng.uniform(-3, 3, (4, 2)).astype(np.float32)
        img = synth.render(380, 380, stones, corners, seed=seed * 100000 + b, noise=rng.uniform(1.5, 4.5)).numpy()
        for i in range(10):
            rs = 17 if i == 9 else 2 * i
            for j in range(10):
                cs = 17 if j == 9 else 2 * j
                X.append(img[ORIG[i]:ORIG[i] + 40, ORIG[j]:ORIG[j] + 40])
                s = stones[rs:rs + 2, cs:cs + 2].reshape(-1)
                Y.append(int(s[0] + 3 * s[1] + 9 * s[2] + 27 * s[3]))
    return np.stack(X), np.array(Y, np.int64)


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(3, 32, 5)
        self.c2 = nn.Conv2d(32, 32, 5)
        self.c3 = nn.Conv2d(32, 90, 3)
        self.c4 = nn.Conv2d(90, 90, 3)
        self.d1 = nn.Linear(3240, 160)
        self.d2 = nn.Linear(160, 81)

    def forward(self, x, train=False):
        x = F.relu(self.c1(x))
        x = F.max_pool2d(F.relu(self.c2(x)), 2)
        if train:
            x = F.dropout(x, 0.25)
        x = F.relu(self.c3(x))
        x = F.max_pool2d(F.relu(self.c4(x)), 2)
        if train:
            x = F.dropout(x, 0.25)
        x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)        # Keras Flatten on channels-last
        x = F.relu(self.d1(x))
        if train:
            x = F.dropout(x, 0.5)
        return self.d2(x)


def export(net, path):
    def conv(w):     # torch correlation [o,c,i,j] -> Keras/Theano true-convolution kernel [kh,kw,c,o]
        return np.ascontiguousarray(w.detach().numpy()[:, :, ::-1, ::-1].transpose(2, 3, 1, 0)).astype(np.float32)
    W = dict(c1w=conv(net.c1.weight) / np.float32(255.0), c1b=net.c1.bias.detach().numpy(),
             c2w=conv(net.c2.weight), c2b=net.c2.bias.detach().numpy(),
             c3w=conv(net.c3.weight), c3b=net.c3.bias.detach().numpy(),
             c4w=conv(net.c4.weight), c4b=net.c4.bias.detach().numpy(),
             d1w=np.ascontiguousarray(net.d1.weight.detach().numpy().T), d1b=net.d1.bias.detach().numpy(),
             d2w=np.ascontiguousarray(net.d2.weight.detach().numpy().T), d2b=net.d2.bias.detach().numpy())
    np.savez_compressed(path, **{k: np.ascontiguousarray(v, np.float32) for k, v in W.items()})


def main():
    nboards = int(sys.argv[1]) if len(sys.argv) > 1 else 260
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    torch.manual_seed(synth.SEED)
    torch.set_num_threads(int(os.environ.get("TRAIN_THREADS", "6")))
    t0 = time.time()
    X, Y = make_boards(nboards, 1)
    Xv, Yv = make_boards(20, 2)
    print("data", X.shape, "%.0fs" % (time.time() - t0), flush=True)
    Xt = torch.from_numpy(X).permute(0, 3, 1, 2).float() / 255.0
    Yt = torch.from_numpy(Y)
    Xvt = torch.from_numpy(Xv).permute(0, 3, 1, 2).float() / 255.0
    net = Net()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    bs = 128
    for ep in range(epochs):
        perm = torch.randperm(len(Xt))
        tot = 0.0
        for k in range(0, len(perm) - bs + 1, bs):
            idx = perm[k:k + bs]
            loss = F.cross_entropy(net(Xt[idx], train=True), Yt[idx])
            opt.zero_grad()
            loss.backward()
            opt.step()
            tot += float(loss)
        with torch.no_grad():
            pred = torch.cat([net(Xvt[k:k + 500]).argmax(1) for k in range(0, len(Xvt), 500)]).numpy()
        acc = float((pred == Yv).mean())
        print("epoch %d loss %.4f val-acc %.4f  %.0fs" % (ep, tot / (len(perm) // bs), acc, time.time() - t0), flush=True)
        if ep == epochs // 2:
            for g in opt.param_groups:
                g["lr"] = 3e-4
    os.makedirs(os.path.join(ROOT, "tools", "out"), exist_ok=True)
    out = os.path.join(ROOT, "tools", "out", "cnn_weights.npz")
    export(net, out)
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
