cd /root/repo
timeout 900 python -m pytest tests/test_gpu_group_modules.py tests/test_gpu_models.py tests/test_gpu_train_full.py -m gpu -q -x 2>&1 | tail -2
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from patchaugnet_amd import configs, patch_aug_net, train_ops, losses
from patchaugnet_amd.train import DEFAULTS, run_model
from patchaugnet_amd.weights import seeded_state_dict
n = 1024
m = patch_aug_net.Network(param=configs.scaled_config(configs.patch_aug_net_config(), n), use_a2a_recon=True, use_l2_norm=True)
m.load_state_dict(seeded_state_dict(m.state_dict())); m = m.cuda().train()
g = torch.Generator().manual_seed(5)
batch = tuple((torch.rand(1, k, n, 3, generator=g) * 2 - 1).cuda() for k in (1, 2, 4, 1))
args = dict(DEFAULTS, TRAIN_NEGATIVES_PER_QUERY=4)
with train_ops.zero_arena(torch.device("cuda", 0)):
    out = run_model(m, *batch, {(0, 1): None, (0, 2): None}, n, True, args=args, input_grad=True)
    r = out["patch_recon"]
    loss = losses.patch_chamfer_loss(r["origin_patches"], r["reconstructed_patches"]) + sum(d.sum() for d in out["global_desc"]) * 1e-3
    loss.backward()
print("input_grad=True step ran; loss", float(loss))
PY
