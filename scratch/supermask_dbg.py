import sys
sys.path[:0]=["/root/repo","/root/repo/tests","/root/repo/tests/golden"]
import torch, common as C, helpers as H
import sparse_image_captioning_amd as P
from sparse_image_captioning_amd.training import NativeTrainer
from sparse_image_captioning_amd.utils.config import Config
torch.manual_seed(8888)
cfg = Config(**dict(C.TINY_CFG, prune_type="supermask", prune_mask_freeze_scope="model.generator.", prune_supermask_init=5.0, drop_prob_src=0.1))
m = P.get_model("relation_transformer_prune")(cfg)
shapes = {k: v for k, v in H.prune_param_shapes(C.TINY_CFG).items() if not k.endswith("_pruning_mask")}
m.load_state_dict(H.torch_state(shapes, C.G1_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS), strict=False)
m = m.cuda(); b = {k: v.cuda() for k, v in H.g1_batch().items()}
m.train()
IT=int(sys.argv[1]) if len(sys.argv)>1 else 60
tr = NativeTrainer(m, noamopt_factor=0.1, noamopt_warmup=10, prune_supermask_lr=10.0, mask_eps=1e-8, sparsity_target=0.8, sparsity_weight=120.0, max_train_step=IT)
for i in range(IT):
    loss = tr.xe_step(b)
    if i % 5 == 0 or i == IT-1:
        print(i, float(loss), float(m.active_mask_sparsities[0]), float(m.all_mask_sparsities[0]), m.sparsity_loss if hasattr(m,'sparsity_loss') else None, flush=True)
