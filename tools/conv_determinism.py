"""Run each UNet conv layer twice on the same input and compare the outputs and BN statistics bit for bit."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_zoo_amd import _lib as L, ops

DEV, dt, B = "cuda", torch.bfloat16, 16
LAYERS = [("e1b", 256, 64, 64), ("e2a", 128, 64, 128), ("e2b", 128, 128, 128), ("e3a", 64, 128, 256),
          ("e3b", 64, 256, 256), ("e4a", 32, 256, 512), ("e4b", 32, 512, 512), ("bna", 16, 512, 1024),
          ("bnb", 16, 1024, 1024), ("d1a", 32, 1024, 512), ("d2a", 64, 512, 256), ("d3a", 128, 256, 128),
          ("d4a", 256, 128, 64), ("x", 128, 192, 128), ("y", 64, 64, 64)]
for name, hw, cin, cout in LAYERS:
    x = ops.new_act(B, hw, hw, cin, dt, DEV); x.buf.normal_()
    w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
    wp = ops.pack_weights(w, L.PACK_CONV_FWD, dt)
    outs = []
    for rep in range(6):
        y = ops.new_act(B, hw, hw, cout, dt, DEV)
        st = ops.conv_igemm(x, wp, None, y, ntaps=9, want_stats=True)
        torch.cuda.synchronize()
        outs.append((y.buf.clone(), st.clone() if st is not None else None))
    bad = sum(1 for o in outs[1:] if not torch.equal(o[0], outs[0][0]))
    bads = sum(1 for o in outs[1:] if o[1] is not None and not torch.equal(o[1], outs[0][1]))
    nd = [(o[0].float() - outs[0][0].float()).abs().max().item() for o in outs[1:]]
    print(f"{name} {hw} {cin}->{cout}: output mismatches {bad}/5 (max diff {max(nd):.3g}), stats mismatches {bads}/5")
