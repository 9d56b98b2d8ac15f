"""ctypes binding of libpatchaugnet_hip.so (C ABI: include/patchaugnet_hip.h).

The HIP library is the product path; there is NO fallback.  If the shared object
is missing or a tensor is not on a HIP device the call raises immediately.
"""
import ctypes
import os
import subprocess

import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("PA_LIB_PATH", os.path.join(_CSRC, "libpatchaugnet_hip.so"))   # override: A/B of two builds in one session
_lib = None

_I, _F, _P = ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# name -> argument type string: i = int, f = float, p = device pointer; a trailing stream pointer is implied
_SIGS = {
    "pa_furthestsampling": "iiippp",
    "pa_gathering_forward": "iiiippp",
    "pa_gathering_backward": "iiiippp",
    "pa_knnquery": "iiiipppp",
    "pa_knnquery_window": "iiiiiipppp",
    "pa_cloud_cellsort": "iipp",
    "pa_knnquery_presorted": "iiiippppp",
    "pa_furthestsampling_range": "iiiiipppp",
    "pa_grouping_forward": "iiiiippp",
    "pa_grouping_backward": "iiiiippp",
    "pa_grouping_int_forward": "iiiiippp",
    "pa_group_edge_forward": "iiiiippppp",
    "pa_group_edge_backward": "iiiiipppp",
    "pa_group_xyz": "iiiiippppp",
    "pa_compose_indices": "iilppp",
    "pa_nearestneighbor": "iiipppp",
    "pa_interpolation_forward": "iiiipppp",
    "pa_interpolation_backward": "iiiipppp",
    "pa_interpolation_backward_gather": "iiiiplpppp",
    "pa_interpolation_backward_lists": "iiippp",
    "pa_ballquery": "iiifippp",
    "pa_featuredistribute": "iiippp",
    "pa_featuregather_forward": "iiiippp",
    "pa_featuregather_backward": "iiiippp",
    "pa_labelstat_and_ballquery": "iiifiippppp",
    "pa_labelstat_ballrange": "iiifipppp",
    "pa_labelstat_idx": "iiiiippp",
    "pa_chamfer_forward": "iiipppppp",
    "pa_chamfer_backward": "iiipppppppp",
    "pa_chamfer_l1_forward": "iiipppppppp",
    "pa_chamfer_l1_backward": "iiippppppppp",
    "pa_knn_generic": "pipiiipp",
    "pa_knn_candidates": "pipipiip",
    "pa_emd_forward": "iiippppppppppfi",
    "pa_emd_backward": "iippppp",
    "pa_mlp_chain": "iiipppplipippppiiiippppiiiipi",
    "pa_rowgroup_max": "liipp",
    "pa_linear": "liipipppipipi",
    "pa_pack_weights": "iipp",
    "pa_pack_weights_f16": "iipp",
    "pa_mlp_chain_f16": "iiippppplipippppiiiippppiiiipi",
    "pa_linear_f16": "liipipppipipi",
    "pa_fp_chain_premul_f16": "ippppplppppiiiippppi",
    "pa_fp_chain_premul": "ippppplppppiiiippppi",
    "pa_mlp_chain_packed": "iiippppplipippppiiiippppiiiipi",
    "pa_sa_attention": "iiipppp",
    "pa_sa_attention_trans": "iiipppppp",
    "pa_sa_attention_f16": "iiiippppp",
    "pa_sa_attention_trans_f16": "iiiippppppp",
    "pa_sa_attention_f16_pack_trans": "ipp",
    "pa_netvlad": "iiiippppppii",
    "pa_furthestsampling_gather": "iiippp",
    "pa_three_nn_weights": "iiipppp",
    "pa_afa": "iiiippppppipp",
    "pa_netvlad_rows": "iiiipppppppii",
    "pa_netvlad_pack_weights": "iipp",
    "pa_afa_rows": "iiiippppppppipp",
    "pa_fc": "iiipppppippp",
    "pa_netvlad_pyramid": "iipppppppppi",
    "pa_netvlad_pyramid_f16": "iippppppppppi",
    "pa_netvlad_pyramid_f16h": "iippppppppppii",
    "pa_afa_fused": "iiiippppppipp",
    "pa_vlad_maxpool": "iiipip",
    "pa_tgemm_nn": "iiiipliipliipppliippipi",
    "pa_tgemm_kk": "iiilpliipppliippliii",
    "pa_tgemm_kk_rep": "iiilpliipppliippipi",
    "pa_adam_tick": "p",
    "pa_adam_step": "ipppppppffff",
    "pa_bn_finalize": "iidpppffpppp",
    "pa_bn_bwd_reduce": "iilpppipi",
    "pa_tgemm_nn_bnred": "iiiipiipliippplippip",
    "pa_bn_eval_params": "iippppfp",
    "pa_quadruplet_loss": "iiiipffiiiipp",
    "pa_softmax_cols": "iiippp",
    "pa_softmax_cols_backward": "iiipppp",
    "pa_vlad_residual_normalize": "iiiipppppp",
    "pa_vlad_residual_normalize_backward": "iiipppppppp",
    "pa_l2_normalize": "iiippp",
    "pa_l2_normalize_backward": "iiipppp",
    "pa_bn_rows_train": "iipppffpppppp",
    "pa_bn_rows_backward": "iipppppppp",
    "pa_afa_attention": "iiippppp",
    "pa_afa_attention_backward": "iiipppppp",
    "pa_attn_softmax_renorm": "iippp",
    "pa_attn_softmax_renorm_backward": "iipppp",
    "pa_bn_bwd_finalize": "iidpppp",
    "pa_bn_apply": "iiliippppi",
    "pa_maxpool_bwd": "ilippp",
    "pa_maxpool_bwd_bnred": "iilipppppip",
    "pa_knn_mfma_select": "pliiiippppiippp",
    "pa_patch_pairs_count": "ipppppiipppp",
    "pa_patch_pairs_fill": "ipppppiipqpppp",
    "pa_fp_premul_g16": "lpipp",
    "pa_fp_fold_forward": "iiiiipppppipp",
    "pa_fp_fold_backward": "iiiiipppippppi",
    "pa_fp_chain_premul_x3": "ippplppppiiiipppi",
    "pa_linear_x3": "lpipfpi",
    "pa_fp_chain_premul_g16": "ipppplppppiiiipppi",
    "pa_fp_chain_premul_g16h": "ipppplppppiiiippp",
}
_T = {"i": _I, "f": _F, "p": _P, "l": ctypes.c_long, "d": ctypes.c_double, "q": ctypes.c_ulonglong}


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 with hipcc (csrc/Makefile).  Cross-compiles without a GPU."""
    cmd = ["make", "-C", _CSRC, "-j8"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("hipcc build of libpatchaugnet_hip.so failed:\n" + res.stdout[-4000:])
    return LIB_PATH


def register(sigs):
    """Declare further entry points (used by the fused-kernel modules)."""
    _SIGS.update(sigs)
    if _lib is not None:
        _declare(_lib, sigs)


def _declare(lib, sigs):
    for name, sig in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = [_T[ch] for ch in sig] + [_P]
        fn.restype = _I


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C patchaugnet_amd/csrc`). There is no CPU or PyTorch fallback for these ops.")
    l = ctypes.CDLL(path)
    l.pa_last_error.restype = ctypes.c_char_p
    l.pa_abi_version.restype = _I
    for name, nargs in (("pa_interpolation_backward_scratch_ints", 3), ("pa_netvlad_scratch_floats", 3), ("pa_afa_scratch_floats", 4), ("pa_fc_scratch_floats", 3), ("pa_afa_rows_scratch_floats", 4), ("pa_pack_weights_f16_halfs", 2),
                        ("pa_afa_fused_scratch_floats", 3), ("pa_attn_train_scratch_floats", 2), ("pa_sa_attention_f16_scratch_halfs", 4), ("pa_cloud_cellsort_floats", 2)):
        getattr(l, name).argtypes = [_I] * nargs
        getattr(l, name).restype = ctypes.c_long
    l.pa_sa_group_window.argtypes, l.pa_sa_group_window.restype = [_I, _I], _I      # thread-local state setter: no stream argument
    _declare(l, _SIGS)
    return l


def lib():
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def has(name):
    """Does the library export `name`?"""
    return hasattr(_lib or lib(), name)


def check_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("patchaugnet_amd ops run on the MI355X only: got a tensor on %s "
                               "(no CPU fallback; the CPU checker lives in oracle/ and is test infrastructure)" % t.device)
        if not t.is_contiguous():
            raise RuntimeError("patchaugnet_amd ops need contiguous tensors")


def ptr(t):
    return _P(t.data_ptr()) if t is not None else _P(0)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() costs ~8 us of Python per call (device
    index resolution); the raw accessors are two C calls -- with ~45 launches per step that is a third of the host's issue time."""
    if _raw_stream is not None and _cur_device is not None:
        return _P(_raw_stream(_cur_device()))
    return _P(torch.cuda.current_stream().cuda_stream)


def call(name, *args, stream=None):
    """Invoke an entry point on the current torch stream; raise on a non-zero return."""
    l = _lib or lib()
    rc = getattr(l, name)(*args, stream if stream is not None else stream_ptr())
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {l.pa_last_error().decode()}")
