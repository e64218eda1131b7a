"""What storing the LayerNorm output gradients as bf16 (the executor's default in mixed precision; ortk_tuning.ln_fuse & 8 keeps fp32) does to
the parameter gradients: full-size configuration (BASELINE configs[0]: 4 images x 5 captions), gradients of the mixed-precision step in
both settings against the fp32 parity mode's — relative error and cosine over the whole arena and the worst single parameter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import common as C
import helpers as H
import sparse_image_captioning_amd as P
from sparse_image_captioning_amd.utils.config import Config
from sparse_image_captioning_amd.training import NativeTrainer
L = P._lib
state = H.torch_state(H.dense_param_shapes(C.FULL_CFG), C.G2_SEED)          # (the weights of golden G2)
b = {k: v.cuda() for k, v in H.torch_batch(C.make_inputs(**C.G2_INPUTS)).items()}
def grads(prec, fuse):
    L.set_tuning(ln_fuse=fuse)
    m = P.get_model("relation_transformer")(Config(**C.FULL_CFG), precision=prec)
    m.load_state_dict(state, strict=False); m = m.cuda().eval()
    tr = NativeTrainer(m, noamopt_factor=0.0, noamopt_warmup=10, keep_grads=True)
    for _ in range(2): tr.xe_step(b, train=False)
    ent = [e for e in m.named_weight_entries()]
    return tr.grads.clone(), ent
g32, ent = grads(0, 0)
for name, fuse in (("fp32 LayerNorm output gradients (ln_fuse 8)", 8), ("bf16 LayerNorm output gradients (default)", 0)):
    g16, _ = grads("bf16", fuse)
    rel = float((g16 - g32).norm() / g32.norm()); cos = float(torch.dot(g16, g32) / (g16.norm() * g32.norm()))
    worst = (0.0, "")
    for e in ent:
        a, c_ = g32[e["offset"]:e["offset"] + e["numel"]], g16[e["offset"]:e["offset"] + e["numel"]]
        if float(a.norm()) < 1e-7 or ".WGs." in e["name"] or e["name"].endswith("attn.linears.1.bias"): continue
        r = float((a - c_).norm() / a.norm())
        if r > worst[0]: worst = (r, e["name"])
    print(f"{name}: whole arena rel. error {rel:.4f}, cosine {cos:.6f}; worst parameter {worst[1]} rel. error {worst[0]:.4f}", flush=True)
L.set_tuning(ln_fuse=0)
