#!/usr/bin/env python3
"""Golden vectors for constructor variants the HSI configs never use but the reference's classes accept (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_variants.py

ResnetBlock(use_affine_level=True) (model/sr3_modules/unet.py:34-50) and SelfAttention(n_head=4) (:114-143): outputs only; inputs and
parameters come from synth.py.
"""
import json
import os

import numpy as np
import torch

from make_golden import HERE, T, _import_reference, fill
from synth import synth_tensor


def main():
    unet, _, _ = _import_reference()
    out, shapes = {}, {}
    with torch.no_grad():
        for tag, cin, cout in (("aff_same", 64, 64), ("aff_proj", 32, 64)):
            m = unet.ResnetBlock(cin, cout, noise_level_emb_dim=32, use_affine_level=True, norm_groups=32)
            shapes.update(fill(m, tag + "."))
            x = synth_tensor(tag + ".x", (2, cin, 8, 8))
            t = synth_tensor(tag + ".t", (2, 1, 32))
            out[tag + ".y"] = m(T(x), T(t)).numpy()
        m = unet.SelfAttention(64, n_head=4, norm_groups=32)
        shapes.update(fill(m, "attn4."))
        out["attn4.y"] = m(T(synth_tensor("attn4.x", (2, 64, 8, 8)))).numpy()
    out["shapes_json"] = np.array(json.dumps(shapes))
    np.savez_compressed(os.path.join(HERE, "variants.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
