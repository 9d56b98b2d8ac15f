"""PPT-Net: pyramid backbone with grouped self-attention + four-scale pyramid NetVLAD.

Model API of ``place_recognition/pptnet_origin/models/pptnet.py:24-62`` as constructed by
``place_recognition/evaluate.py:91-96``: ``Network(param=cfg, use_normalize=False|True)``;
``forward(x: (B,1,N,3)) -> (desc (B,256), fp_features, center_idx)``.  State-dict keys equal the reference's
(tests/golden/pptnet_state_dict_keys.json), including the doubly-saved tied q/k weights.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import loupe as lp
from .backbone import PyramidBackbone, SALayer

SA_Layer = SALayer          # the reference's class name (pptnet.py:246)

__all__ = ["Network", "SA_Layer"]


class Network(nn.Module):
    def __init__(self, param=None, use_normalize=True):
        super().__init__()
        fs, c = param["FEATURE_SIZE"], 3
        self.backbone = PyramidBackbone(
            sampling=param["SAMPLING"], knn=param["KNN"], gp=param["GROUP"], attention=True,
            sa_mlps=[[c, 32, 32, 64], [64, 64, 64, 128], [128, 128, 128, 256], [256, 256, 256, 512]],
            fp_mlps=[[fs[1] + c, 256, 256, fs[0]], [fs[2] + 64, 256, fs[1]], [fs[3] + 128, 256, fs[2]], [512 + 256, 256, fs[3]]])
        if param["AGGREGATION"] != "spvlad":
            raise ValueError("No aggregation algorithm: %r" % (param["AGGREGATION"],))
        self.aggregation = lp.SpatialPyramidNetVLAD4(
            feature_size=param["FEATURE_SIZE"], max_samples=param["MAX_SAMPLES"], cluster_size=param["CLUSTER_SIZE"],
            output_dim=param["OUTPUT_DIM"], gating=param["GATING"], add_batch_norm=True)
        self.use_normalize = use_normalize
        self.param = dict(param)
        self._engine = None
        self.fused_eval = True       # eval()+no_grad() forwards run the fused HIP engine; set False for the module path

    def prepare(self, device=None):
        """Build the fused engine now, on the caller's current stream (see patch_aug_net.Network.prepare)."""
        from .engine import engine_for
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        return engine_for(self, dev)

    def invalidate_engine(self):
        self._engine = None

    def train(self, mode=True):
        self._engine = None
        return super().train(mode)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_engine"] = None               # ctypes pointer arrays: never copied or pickled
        st.pop("_graphed_extractors", None)      # captured hipGraphs of distributed.extract_dataset: bound to THIS module's engine
        return st

    def forward(self, x, return_feat=True, use_engine=None):
        if use_engine is None:
            use_engine = self.fused_eval and not self.training and not torch.is_grad_enabled()
        if use_engine:
            from .engine import engine_for
            eng = engine_for(self, x.device)
            if getattr(self, "geo_overlap", None) is not None:       # latency mode: see PatchAugNetEngine.geo_overlap
                eng.geo_overlap = bool(self.geo_overlap)
            d, (fp, cidx) = eng.forward(x, views=return_feat)
            return (d, fp, cidx) if return_feat else d
        res = self.backbone(x.squeeze(1))
        fp = res["fp_features"]
        d = self.aggregation(fp[0], fp[1], fp[2], fp[3])
        if self.use_normalize:
            d = F.normalize(d)
        return (d, fp, res["center_idx_origin"]) if return_feat else d
