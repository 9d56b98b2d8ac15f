"""PatchAugNet: pyramid backbone + pyramid NetVLAD with adaptive feature aggregation + patch decoder.

Model API of ``place_recognition/patch_aug_net/models/patch_aug_net.py:22-107`` and the constructor call of
``place_recognition/evaluate.py:102-104``:  ``Network(param=cfg, use_a2a_recon=True, use_l2_norm=True)``;
``forward(x: (B,1,N,3) fp32) -> (desc (B,256), fp_features [(B,256,N_i,1)], center_idx [(B,m_i) int32])``,
or with ``nn_dict`` the training tuple ``((desc, patch_recon_data), fp_features, center_idx)``.
State-dict keys equal the reference's (tests/golden/patch_aug_net_state_dict_keys.json).

In ``eval()`` mode under ``torch.no_grad()`` the forward pass runs the fused HIP engine
(patchaugnet_amd/engine.py); otherwise -- train(), or eval() with autograd -- the autograd-capable module path
(patchaugnet_amd/backbone.py), whose dense layers run on the HIP GEMM kernels too (patchaugnet_amd/train_ops.py):
no forward of a model on the MI355X reaches torch.matmul / rocBLAS / MIOpen.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import loupe as lp
from . import pointops, train_ops
from .backbone import PyramidBackbone

__all__ = ["Network", "PointNetDecoder"]


class PointNetDecoder(nn.Module):
    """pointnet_autoencoder.py:85-111 -- FC 256 -> 1024 -> 1024 -> num_points*3, BN + ReLU, tanh."""

    def __init__(self, embedding_size, output_channels=3, num_points=1024):
        super().__init__()
        self.num_points, self.output_channels = num_points, output_channels
        self.fc1 = nn.Linear(embedding_size, 1024)
        self.fc2 = nn.Linear(1024, 1024)
        self.bn1 = nn.BatchNorm1d(1024)
        self.bn2 = nn.BatchNorm1d(1024)
        self.fc3 = nn.Linear(1024, num_points * output_channels)

    def forward_cm(self, xt):
        """train() mode on the MI355X, channel-major: xt (R, C, P) = R related clouds, the P patch features of each as columns ->
        (R, P, num_points, 3).  Every cloud is its own BatchNorm batch, as if the decoder were called once per cloud in order
        (patch_aug_net.py:83-98).  Both hidden layers are one chain on the MFMA GEMM kernels (csrc/train_gemm.hip; BatchNorm statistics
        over the P columns fused into the GEMM epilogues, BatchNorm + ReLU applied by the next GEMM's loader), the output layer a GEMM
        with bias + tanh epilogue."""
        layers = [train_ops.BNLayer(self.fc1.weight, self.bn1, bias=self.fc1.bias), train_ops.BNLayer(self.fc2.weight, self.bn2, bias=self.fc2.bias)]
        h = train_ops.chain_train(xt, layers, groups=True, training=self.training)
        y = train_ops.linear_cm(h, self.fc3.weight, self.fc3.bias, act=1)             # (R, num_points*3, P)
        return y.transpose(1, 2).contiguous().view(xt.shape[0], xt.shape[2], self.num_points, self.output_channels)

    def forward(self, x):
        if train_ops.on_device(x):
            return self.forward_cm(x.t().contiguous().unsqueeze(0))[0]
        x = F.relu(self.bn1(self.fc1(x)))
        x = F.relu(self.bn2(self.fc2(x)))
        return torch.tanh(self.fc3(x)).view(x.shape[0], self.num_points, self.output_channels).contiguous()


class Network(nn.Module):
    def __init__(self, param=None, use_a2a_recon=False, use_l2_norm=False):
        super().__init__()
        fs, c = param["FEATURE_SIZE"], 3
        c_fp = c if param["USE_ORIGIN_PC_IN_FP"] else 0
        self.backbone = PyramidBackbone(
            sampling=param["SAMPLING"], knn=param["KNN"], knn_dilation=param["KNN_DILATION"], gp=param["GROUP"],
            sa_mlps=[[c, 32, 32, 64], [64, 64, 64, 256], [256, 256, 256, 512]],
            fp_mlps=[[fs[1] + c_fp, 256, 256, fs[0]], [fs[2] + 64, 256, fs[1]], [512 + 256, 256, fs[2]]],
            use_origin_pc_in_fp=param["USE_ORIGIN_PC_IN_FP"])
        if param["AGGREGATION"] != "spvlad":
            raise ValueError("No aggregation algorithm: %r" % (param["AGGREGATION"],))
        self.aggregation = lp.SpatialPyramidNetVLAD(
            feature_size=param["FEATURE_SIZE"], max_samples=param["MAX_SAMPLES"], cluster_size=param["CLUSTER_SIZE"],
            output_dim=param["OUTPUT_DIM"], gating=param["GATING"], aggregation_type=param["AGGREGATION_TYPE"], add_batch_norm=True)
        self.use_l2_norm, self.use_a2a_recon = use_l2_norm, use_a2a_recon
        if use_a2a_recon:
            self.decoder = PointNetDecoder(embedding_size=256, num_points=param["KNN"][0])
        self.param = dict(param)
        self._engine = None
        self.fused_eval = True       # eval()+no_grad() forwards run the fused HIP engine; set False for the module path

    # ---- fused inference engine (eval + no_grad) ----------------------------------------------------------------
    def _fused(self, x, views=True):
        from .engine import engine_for
        eng = engine_for(self, x.device)
        if getattr(self, "geo_overlap", None) is not None:       # latency mode: see PatchAugNetEngine.geo_overlap
            eng.geo_overlap = bool(self.geo_overlap)
        return eng.forward(x, views=views)

    def prepare(self, device=None):
        """Build the fused engine (BatchNorm folding, weight packing) NOW, on the caller's current stream of `device`.  Callers that
        issue forwards on several streams (extract.StreamPipeline, GraphedExtractor) do this first, so no pack kernel is ever
        enqueued from inside a pipelined forward."""
        from .engine import engine_for
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        return engine_for(self, dev)

    def invalidate_engine(self):
        """Drop the folded / packed weight copies (they are rebuilt at the next eval forward)."""
        self._engine = None

    def train(self, mode=True):
        self._engine = None            # parameters may change: rebuild folded weights at the next eval forward
        return super().train(mode)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def __getstate__(self):
        """The engine holds ctypes pointer arrays into device buffers: never copied or pickled (copy.deepcopy / torch.save of the module)."""
        st = self.__dict__.copy()
        st["_engine"] = None
        st.pop("_related_cache", None)
        st.pop("_graphed_extractors", None)      # captured hipGraphs of distributed.extract_dataset: bound to THIS module's engine
        return st

    def related_index(self, device, related):
        """Device index tensor of the related-cloud list, made once per (device, key set): no host -> device copy per step.  Entries are
        kept per key set -- a forward with ANOTHER key set (validation, a second trainer) never replaces or frees one that a captured
        hipGraph has baked the address of; train.GraphedTrainer additionally holds its own reference for the graph's lifetime, so the
        bounded eviction below cannot free it either."""
        cache = self.__dict__.setdefault("_related_cache", {})
        key = (str(device), tuple(related))
        t = cache.get(key)
        if t is None:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("patch_aug_net.Network: nn_dict key set %r first seen while a hipGraph is being captured; run one eager "
                                   "forward with it first (GraphedTrainer's warm-up does)" % (tuple(related),))
            while len(cache) >= 64:
                cache.pop(next(iter(cache)))
            t = cache[key] = torch.tensor(list(related), dtype=torch.long, device=device)
        return t

    def forward(self, x, nn_dict=None, return_feat=True, use_engine=None, geometry=None):
        """x: (B, 1, N, 3).  geometry: the result of ``self.backbone.geometry(x.squeeze(1))`` for this x (module path only): the sampling /
        grouping / 3-NN launches are then skipped (train.GraphedTrainer computes it for the next batch while the current one trains)."""
        if use_engine is None:
            use_engine = self.fused_eval and not self.training and not torch.is_grad_enabled() and nn_dict is None
        fused = use_engine
        if fused:
            desc, (fp_features, center_idx) = self._fused(x, views=return_feat)
            return (desc, fp_features, center_idx) if return_feat else desc
        xyz = x.squeeze(1)
        res = self.backbone(xyz, geometry=geometry)
        center_idx, sample_idx, fp_features = res["center_idx_origin"], res["sample_idx_origin"], res["fp_features"]
        out = self.aggregation(fp_features)
        if nn_dict is not None:                                   # patch reconstruction branch (:68-104)
            related = sorted({i for pair in nn_dict for i in pair})
            origin = res.get("origin_patches_cm")                                           # (B, 3, m0, k): with the precomputed geometry
            if origin is None:
                origin = pointops.grouping(xyz.transpose(1, 2).contiguous(), sample_idx[0])
            data = {"cloud_indices": related, "center_indices": [], "origin_patches": [], "patch_features": [],
                    "reconstructed_patches": []}
            hip_dec = self.use_a2a_recon and train_ops.on_device(x)
            feats_cm = fp_features[1].squeeze(-1).index_select(0, self.related_index(x.device, related))   # (R, 256, m0): one patch feature per column
            if self.use_l2_norm:
                feats_cm = train_ops.l2_normalize(feats_cm) if train_ops.on_device(x) else F.normalize(feats_cm, dim=1)
            recon = self.decoder.forward_cm(feats_cm.contiguous()) if hip_dec else None      # all related clouds in one set of launches
            for r, ci in enumerate(related):
                feats = feats_cm[r].transpose(1, 0)                                          # (m0, 256)
                data["center_indices"].append(center_idx[0][ci:ci + 1])
                data["origin_patches"].append(origin[ci].permute(1, 2, 0))                   # (m0, k, 3)
                data["patch_features"].append(feats)
                if hip_dec:
                    data["reconstructed_patches"].append(recon[r])
                elif self.use_a2a_recon:
                    data["reconstructed_patches"].append(self.decoder(feats))
            out = out, data
        return (out, fp_features, center_idx) if return_feat else out
