"""MI355X-native implementation of PatchAugNet's descriptor-extraction hot path.

Layout:
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/patchaugnet_hip.h)
  _lib.py          ctypes binding of libpatchaugnet_hip.so (no fallback: raises if missing)
  pointops.py      host mirror of the reference's libs/pointops/functions/pointops.py
  pointops_cuda.py the reference's pybind module name, re-exported on top of the C ABI
  pt_util.py, loupe.py, backbone.py, patch_aug_net.py, pptnet.py
                   model classes with the reference's API and state-dict keys
  engine.py        fused inference engine (eval mode)
  weights.py, configs.py  deterministic weights / synthetic submaps, hot-path hyper-parameters
"""
__version__ = "0.1.0"
