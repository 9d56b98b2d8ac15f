"""One training step around the descriptor path (BASELINE.json configs[3]).

Mirrors ``run_model`` (place_recognition/train_place_recognition.py:142-164) and the loss assembly of ``train_one_epoch``
(:255-392): the tuple (query, positives, negatives, other negative) is concatenated to one (T, 1, N, 3) batch, pushed through
the model (module path: autograd over the HIP point ops' backward kernels), descriptors are split back into the four
groups for the quadruplet loss, and the patch-reconstruction branch feeds the HIP Chamfer loss.  The contrastive
patch-feature term (:308-385) needs the reference's precomputed overlap protobufs and is out of scope here.
"""
import numpy as np
import torch

from . import losses

DEFAULTS = {  # configs/patch_aug_net.yaml:55-75 (training section)
    "TRAIN_POSITIVES_PER_QUERY": 2, "TRAIN_NEGATIVES_PER_QUERY": 14, "MARGIN_1": 0.5, "MARGIN_2": 0.2,
    "TRIPLET_USE_BEST_POSITIVES": False, "LOSS_LAZY": True, "LOSS_IGNORE_ZERO_BATCH": False, "FEATURE_OUTPUT_DIM": 256,
}


def _as_tensor(a):
    return torch.from_numpy(a).float() if isinstance(a, np.ndarray) else a.float()


def run_model(model, queries, positives, negatives, other_neg, nn_dict=None, num_points=4096, require_grad=True, device=None, args=DEFAULTS):
    """train_place_recognition.py:142-164.  queries (bs,1,N,3), positives (bs,P,N,3), negatives (bs,Nn,N,3), other_neg (bs,1,N,3);
    returns {'global_desc': (q, pos, neg, other) split along dim 1, 'patch_recon': dict or None}."""
    device = device or next(model.parameters()).device
    q = _as_tensor(queries)
    # the four groups go to the device first and are concatenated there (a host-side cat spins the ATen thread pool, hostcpu.py)
    feed = torch.cat([_as_tensor(t).to(device, non_blocking=True) for t in (q, positives, negatives, other_neg)], 1)
    feed = feed.view((-1, 1, num_points, 3)).requires_grad_(require_grad)
    with torch.set_grad_enabled(require_grad):
        out = model(feed, nn_dict, return_feat=False) if nn_dict is not None else model(feed, return_feat=False)
    desc, recon = out if nn_dict is not None else (out, None)
    desc = desc.view(q.shape[0], -1, args["FEATURE_OUTPUT_DIM"])
    split = torch.split(desc, [1, args["TRAIN_POSITIVES_PER_QUERY"], args["TRAIN_NEGATIVES_PER_QUERY"], 1], dim=1)
    return {"global_desc": split, "patch_recon": recon}


def training_step(model, optimizer, queries, positives, negatives, other_neg, nn_dict=None, num_points=4096, args=DEFAULTS,
                  loss_alpha=None, place_loss="quadruplet", recon_loss="patch_chamfer"):
    """train_one_epoch's body for one batch (:255-392 without the overlap-pair term): returns the dict of weighted losses."""
    loss_alpha = loss_alpha or {"place_recognition": 1.0, "patch_recon_a2a": 1.0}
    model.train()
    optimizer.zero_grad(set_to_none=True)
    out = run_model(model, queries, positives, negatives, other_neg, nn_dict, num_points, True, args=args)
    oq, op, on, oo = out["global_desc"]
    cur = {"place_recognition": losses.get_loss_func(place_loss)(oq, op, on, oo, args["MARGIN_1"], args["MARGIN_2"],
                                                                 use_min=args["TRIPLET_USE_BEST_POSITIVES"], lazy=args["LOSS_LAZY"],
                                                                 ignore_zero_loss=args["LOSS_IGNORE_ZERO_BATCH"])}
    recon = out["patch_recon"]
    if recon is not None and getattr(model, "use_a2a_recon", False):
        cur["patch_recon_a2a"] = losses.get_loss_func(recon_loss)(recon["origin_patches"], recon["reconstructed_patches"])
    total = 0.0
    for k in cur:
        cur[k] = cur[k] * loss_alpha.get(k, 1.0)
        total = total + cur[k]
    if float(total.detach()) > 1e-10:                             # :390-392
        total.backward()
        optimizer.step()
    cur["total"] = total
    return {k: float(v.detach()) for k, v in cur.items()}
