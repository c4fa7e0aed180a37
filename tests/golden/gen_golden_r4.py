#!/usr/bin/env python3
"""Round-4 golden vectors (tests/golden/golden_ref3.npz), again produced by RUNNING THE REFERENCE ITSELF on CPU.

    python tests/golden/gen_golden_r4.py          (build container only: needs /root/reference)

Same method and stand-ins as gen_golden.py (imported from there); a third file so that the first two stay byte-identical.
  g16_temb_nf{16,64}   the time embedding of NCSNpp.forward in isolation (models/ncsnpp.py:324-343): the reference backbone's
                       own modules all_modules[0..2] (GaussianFourierProjection, Linear, Linear) applied as its forward applies
                       them, for t = (1.0, 0.53, 0.2, 0.03) — SURVEY.md section 8 row a14
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
from gen_golden import load_synth_weights, model_config  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    G._install_stubs()
    sys.path.insert(0, G.REF)
    import pl_model as ref_pl  # reference

    out = {}
    t = torch.tensor([1.0, 0.53, 0.2, 0.03])
    for nf in (16, 64):
        model = ref_pl.DiffSepModel(model_config(nf, 2))
        bb = model.score_model.backbone
        load_synth_weights(bb, 7)
        bb.eval()
        m = bb.all_modules
        temb = m[0](torch.log(t))            # ncsnpp.py:327-329 (embedding_type == "fourier": the input is log t)
        temb = m[1](temb)                    # ncsnpp.py:341
        temb = m[2](bb.act(temb))            # ncsnpp.py:342
        out[f"g16_temb_nf{nf}"] = temb.numpy()
    np.savez_compressed(os.path.join(HERE, "golden_ref3.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
