#!/usr/bin/env python
"""Independent float64 cross-check of the oracle's network arithmetic.  The oracle (oracle/orc_network.c) restates tiny-cuda-nn
from recall and nothing in the reference tree pins it; this script is a SECOND, separately written statement of the same model, from the
Instant-NGP paper (Müller et al. 2022, §3: multiresolution hash encoding with trilinear interpolation and the spatial hash
x ^ y * 2654435761 ^ z * 805459861; small ReLU MLPs; spherical harmonics of degree 4 for the view direction) and Kingma & Ba's Adam:

  * forward in float64 with torch, gradients by AUTOGRAD (the oracle's backward is hand-derived),
  * the SH basis from scipy.special.sph_harm_y (real combinations), not from a table of constants,
  * Adam + weight decay on matrix parameters + bias-corrected EMA written from the update rules.

It writes tests/golden/network_f64.npz (inputs + float64 results); tests/test_oracle_f64_cpu.py holds orc_nerf_inference,
orc_nerf_forward_backward, orc_sh4 and orc_adam_ema_step to it within the fp16 storage tolerance.  What this cannot pin (and DESIGN.md §2 says so):
tcnn's level geometry (scale_l = N_min b^l - 1, 8-entry alignment), its parameter order and its initialisation — those enter here as inputs.

usage: python tests/golden/make_network_f64.py      (needs torch + scipy; CPU only)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd"), os.path.join(ROOT, "tests")]
PRIMES = (1, 2654435761, 805459861)


def level_index(res, size, x, y, z):
    """dense while the level fits its table, else the paper's spatial hash (mod table size)"""
    if res ** 3 <= size:
        return x + y * res + z * res * res
    return ((x * PRIMES[0]) ^ (y * PRIMES[1]) ^ (z * PRIMES[2])) % size


def encode(levels, table, pos):
    """pos: (n, 3) float64 in [0, 1]; table: (entries, 2) float64 torch tensor; -> (n, 32)"""
    feats = []
    for (scale, res, offset, size) in levels:
        p = pos * float(np.float32(scale)) + 0.5
        g = torch.floor(p)
        w = p - g
        gi = g.detach().to(torch.int64).numpy()
        acc = 0
        for c in range(8):
            bit = [(c >> d) & 1 for d in range(3)]
            idx = level_index(res, size, *(gi[:, d] + bit[d] for d in range(3)))
            weight = 1.0
            for d in range(3):
                weight = weight * (w[:, d] if bit[d] else 1.0 - w[:, d])
            acc = acc + weight[:, None] * table[torch.from_numpy(idx + offset)]
        feats.append(acc)
    return torch.cat(feats, dim=1)


def real_sh4(dirs):
    """16 real spherical harmonics (l <= 3) of directions, (n, 3) float64 -> (n, 16), ordered l^2 + l + m: sqrt(2) Re / Im of scipy's complex
    Y_l^|m| (which carries the Condon-Shortley phase, so Y_1,1 ~ -x, Y_1,-1 ~ -y as in the usual graphics tables)"""
    from scipy.special import sph_harm_y
    d = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    theta = np.arccos(np.clip(d[:, 2], -1, 1))          # polar
    phi = np.arctan2(d[:, 1], d[:, 0])                   # azimuth
    out = np.zeros((len(d), 16))
    for l in range(4):
        for m in range(-l, l + 1):
            y = sph_harm_y(l, abs(m), theta, phi)
            if m == 0:
                v = y.real
            elif m > 0:
                v = np.sqrt(2.0) * y.real
            else:
                v = np.sqrt(2.0) * y.imag
            out[:, l * l + l + m] = v
    return out


def network(params, levels, coords, sh_fn):
    """(r, g, b, sigma) raw network outputs; params: dict of float64 tensors"""
    pos, dirs = coords[:, 0:3], coords[:, 4:7]
    x = encode(levels, params["grid"], pos)   # (floor() has zero gradient: the weights carry d/dpos)
    h1 = torch.relu(x @ params["W1"].T)
    dens = h1 @ params["W2"].T                                  # 16 outputs, [0] = density
    sh = sh_fn(dirs)
    h2 = torch.relu(torch.cat([dens, sh], dim=1) @ params["W3"].T)
    h3 = torch.relu(h2 @ params["W4"].T)
    rgb = h3 @ params["W5"].T
    return torch.cat([rgb[:, 0:3], dens[:, 0:1]], dim=1)


def main():
    import capi
    import helpers as H
    ngp = capi.load_ngp_hip()          # host-only call: the level table of the descriptor (an INPUT of this check)
    desc = H.make_desc(ngp, log2_hashmap_size=11)
    levels = [(float(l["scale"]), int(l["resolution"]), int(l["offset"]), int(l["size"])) for l in desc["levels"][0]]
    n_entries = int(desc["n_grid_entries"][0])
    rs = np.random.RandomState(7)
    n = 384
    coords = np.zeros((n, 7), np.float32)
    coords[:, 0:3] = rs.rand(n, 3)
    d = rs.randn(n, 3); d /= np.linalg.norm(d, axis=1, keepdims=True)
    coords[:, 4:7] = (d + 1) * 0.5                               # directions arrive warped to [0, 1] (testbed_nerf.cu:308-316)
    sizes = [("W1", 64, 32), ("W2", 16, 64), ("W3", 64, 32), ("W4", 64, 64), ("W5", 16, 64)]
    p16 = np.concatenate([(rs.randn(o * i) * np.sqrt(2.0 / i)).astype(np.float16) for _, o, i in sizes] + [(rs.randn(n_entries * 2) * 0.5).astype(np.float16)])
    dl = (rs.randn(n, 4) * 0.05).astype(np.float16)

    def tensors():
        out, k = {}, 0
        for name, o, i in sizes:
            out[name] = torch.tensor(p16[k:k + o * i].astype(np.float64).reshape(o, i), requires_grad=True); k += o * i
        out["grid"] = torch.tensor(p16[k:].astype(np.float64).reshape(n_entries, 2), requires_grad=True)
        return out

    P = tensors()
    c64 = torch.tensor(coords.astype(np.float64), requires_grad=True)
    sh_np = real_sh4(coords[:, 4:7].astype(np.float64) * 2.0 - 1.0)
    out = network(P, levels, c64, lambda dirs: torch.tensor(sh_np))
    loss = (out * torch.tensor(dl.astype(np.float64))).sum()
    loss.backward()
    grads = np.concatenate([P[name].grad.numpy().ravel() for name, _, _ in sizes] + [P["grid"].grad.numpy().ravel()])
    dpos = c64.grad.numpy()[:, 0:3].copy()      # d loss / d position (trilinear interpolation is piecewise linear in the position)
    # d loss / d direction: the SH basis came from scipy (no autograd); central differences of the whole model in float64 instead
    ddir = np.zeros((n, 3))
    h = 1e-6
    for k in range(3):
        res = []
        for sgn in (+1, -1):
            cc = coords.astype(np.float64).copy(); cc[:, 4 + k] += sgn * h
            with torch.no_grad():
                Pn = {key: v.detach() for key, v in P.items()}
                o = network(Pn, levels, torch.tensor(cc), lambda dirs: torch.tensor(real_sh4(cc[:, 4:7] * 2.0 - 1.0)))
            res.append((o * torch.tensor(dl.astype(np.float64))).sum(dim=1).numpy())
        ddir[:, k] = (res[0] - res[1]) / (2 * h)

    # Adam (Kingma & Ba) with L2 on the matrix weights, zero-gradient skip for encoding entries, bias-corrected EMA of the fp16 weights
    m_n = 4000
    g16 = (rs.randn(m_n) * 3.0).astype(np.float16); g16[rs.rand(m_n) < 0.3] = 0
    n_matrix = 1500
    master = rs.randn(m_n) * 0.1
    m1, m2, ema = rs.randn(m_n) * 1e-3, np.abs(rs.randn(m_n)) * 1e-4, rs.randn(m_n) * 0.1
    step, lr, b1, b2, eps, l2, scale, decay = 7, 1e-2, 0.9, 0.99, 1e-15, 1e-6, 128.0, 0.95
    A = dict(master=master.astype(np.float32), m1=m1.astype(np.float32), m2=m2.astype(np.float32), ema=ema.astype(np.float32))
    w, a1, a2 = A["master"].astype(np.float64), A["m1"].astype(np.float64), A["m2"].astype(np.float64)
    g = g16.astype(np.float64) / scale
    is_matrix = np.arange(m_n) < n_matrix
    g = np.where(is_matrix, g + l2 * w, g)
    active = is_matrix | (g16.astype(np.float64) != 0)
    n1 = np.where(active, b1 * a1 + (1 - b1) * g, a1)
    n2 = np.where(active, b2 * a2 + (1 - b2) * g * g, a2)
    lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    nw = np.where(active, w - lr_t * n1 / (np.sqrt(n2) + eps), w)
    p16_prev = A["master"].astype(np.float16)                     # the fp16 copy the skipped entries keep
    w16 = np.where(active, nw.astype(np.float32).astype(np.float16).astype(np.float64), p16_prev.astype(np.float64))
    new_ema = (A["ema"].astype(np.float64) * decay * (1 - decay ** (step - 1)) + w16 * (1 - decay)) / (1 - decay ** step)

    out_path = os.path.join(ROOT, "tests", "golden", "network_f64.npz")
    np.savez_compressed(out_path, dL_dpos=dpos, dL_ddir=ddir, desc=desc.view(np.uint8), coords=coords, params16=p16.view(np.uint16), dL_dout=dl.view(np.uint16), out=out.detach().numpy(), grads=grads, sh=sh_np,
                        adam_grads16=g16.view(np.uint16), adam_master=A["master"], adam_m1=A["m1"], adam_m2=A["m2"], adam_ema=A["ema"], adam_params16=p16_prev.view(np.uint16),
                        adam_hyper=np.array([step, lr, b1, b2, eps, l2, scale, decay, n_matrix], np.float64),
                        adam_new_master=nw, adam_new_m1=n1, adam_new_m2=n2, adam_new_ema=new_ema)
    print("wrote", out_path, os.path.getsize(out_path), "bytes; |out| max", float(out.abs().max()), "grad norms", float(np.linalg.norm(grads[:10240])), float(np.linalg.norm(grads[10240:])))


if __name__ == "__main__":
    main()
