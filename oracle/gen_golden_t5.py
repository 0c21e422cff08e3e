#!/usr/bin/env python3
"""Generate tests/golden/t5_*.npz by running the REFERENCE's WanT5EncoderModel (build container only).

    python oracle/gen_golden_t5.py

Weights come from the integer-hash fill in ``videocof_amd/weights.py``; the fixtures hold inputs and the
reference's fp32 outputs only.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_import import load_reference                       # noqa: E402
from videocof_amd.weights import deterministic_t5_state_dict        # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TINY = dict(vocab=97, dim=256, dim_attn=128, dim_ffn=320, num_heads=2, num_layers=2, num_buckets=32)


@torch.no_grad()
def main():
    torch.manual_seed(0)
    te = load_reference().load_text_encoder()
    sd = deterministic_t5_state_dict(**TINY)
    model = te.WanT5EncoderModel(shared_pos=False, dropout=0.0, **TINY).eval()
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected

    # (a) bucket function over the full +-511 range the 512-token encoder uses, and a wider one
    rel = torch.arange(-700, 701)
    buckets = model.blocks[0].pos_embedding._relative_position_bucket(rel.clone())
    # (b) position bias tensor of block 0 at L = 20
    bias = model.blocks[0].pos_embedding(20, 20)
    # (c) full forward, batch of 2 padded to 72 tokens with 37 and 11 valid ones, and a full-length row
    g = torch.Generator().manual_seed(3)
    L = 72
    ids = torch.randint(1, TINY["vocab"], (3, L), generator=g)
    lens = [37, 11, 72]
    mask = torch.zeros(3, L, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        ids[b, n:] = 0
    out = model(ids, mask)[0]
    out_nomask = model(ids[:1], None)[0]
    # (d) one block in isolation
    x = torch.from_numpy(np.random.RandomState(5).uniform(-1.5, 1.5, (1, 40, TINY["dim"])).astype(np.float32))
    m = torch.ones(1, 40, dtype=torch.long)
    m[0, 29:] = 0
    y = model.blocks[1](x, m, pos_bias=None)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "t5_g13_encoder.npz")
    np.savez_compressed(path, rel=rel.numpy(), buckets=buckets.numpy(), bias20=bias.numpy(),
                        ids=ids.numpy(), mask=mask.numpy(), out=out.numpy(), out_nomask=out_nomask.numpy(),
                        blk_x=x.numpy(), blk_mask=m.numpy(), blk_y=y.numpy())
    print(f"t5_g13_encoder: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
