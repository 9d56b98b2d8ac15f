"""Host-side checks that need no GPU: the C-ABI library loads, exports every symbol include/patchaugnet_hip.h declares,
the product package refuses CPU tensors (no silent fallback) and never imports the oracle, state-dict key parity."""
import ctypes
import json
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from patchaugnet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.LIB_PATH


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "patchaugnet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long|void|const char \*)\s*\*?\s*([a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    names = declared_symbols()
    assert len(names) >= 40, names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.pa_abi_version.restype = ctypes.c_int
    assert lib.pa_abi_version() >= 1


def test_no_packed_fp32_instruction_carries_an_operand_modifier(libpath, tmp_path):
    """The gfx950 fault recorded in csrc/pa_common.h (pa_pk_plain) and DESIGN.md section 5: v_pk_{add,mul,fma}_f32 with op_sel / op_sel_hi / neg_lo / neg_hi
    return wrong values while a neighbouring wave issues 16x16x32 MFMAs (every kernel of the "f16" mode).  The build keeps the form out
    (-fno-slp-vectorize -fno-vectorize, pinned pairs in the hand-written pair arithmetic); this disassembles every code object of the library and checks it."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump in this image")
    copy = tmp_path / "lib.so"
    copy.write_bytes(open(libpath, "rb").read())
    subprocess.run([objdump, "--offloading", str(copy)], check=True, capture_output=True, cwd=tmp_path)      # writes one file per bundle next to the copy
    objects = sorted(f for f in os.listdir(tmp_path) if f.endswith("gfx950"))
    assert len(objects) >= 30, objects
    packed, offenders = 0, []
    for f in objects:
        dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        kernel = "?"
        for line in dis.splitlines():
            m = re.match(r"[0-9a-f]+ <(\w+)>:", line)
            if m:
                kernel = m.group(1)
            elif re.search(r"\bv_pk_\w+_f32\b", line):
                packed += 1
                if re.search(r"op_sel|neg_lo|neg_hi", line):
                    offenders.append((kernel, line.split("//")[0].strip()))
    assert packed > 100, packed                       # the check is looking at real code: the sampling and chain kernels do use the plain forms
    assert not offenders, offenders[:10]


def test_bindings_cover_the_pa_entry_points(libpath):
    from patchaugnet_amd import _lib
    pa = [n for n in declared_symbols() if n.startswith("pa_") and n not in ("pa_abi_version", "pa_last_error", "pa_sa_group_window")
          and not n.endswith("_scratch_floats") and not n.endswith("_scratch_halfs") and not n.endswith("_cellsort_floats") and n not in ("pa_pack_weights_f16_halfs", "pa_interpolation_backward_scratch_ints")]
    assert sorted(pa) == sorted(_lib._SIGS), set(pa) ^ set(_lib._SIGS)


def internal_symbols():
    """Test / profiling hooks declared by the private header csrc/pa_internal.h."""
    src = open(os.path.join(ROOT, "patchaugnet_amd", "csrc", "pa_internal.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long|void)\s*\*?\s*([a-z_0-9]+)\s*\(", src)))


def test_public_header_is_the_boundary_only():
    """include/patchaugnet_hip.h declares the reference's launcher names and the pa_* ops: no A/B switch, debug hook or experimental entry."""
    names = declared_symbols()
    assert not [n for n in names if n.endswith("_enable") or "debug" in n or n in ("pa_fpx256", "pa_fp_chain_premul_tap")]
    text = open(os.path.join(ROOT, "include", "patchaugnet_hip.h")).read()
    assert "getenv" not in text and not re.search(r"\bPA_[A-Z]+_[A-Z_]+=", text)


def test_library_exports_its_hooks_and_no_deleted_variant(libpath):
    """One library: the hooks of csrc/pa_internal.h are exported; the kernel variants that lost their A/B (lane-per-query kNN, register-resident
    FPX chain, wave-private training GEMM, FPS without the LDS copy, the resident / tail EMD forms) are gone from the sources, not parked in a
    second library."""
    hooks = internal_symbols()
    assert len(hooks) >= 5
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    assert not [n for n in exported if any(t in n for t in ("knn_lane", "fpx256", "tgemm_nnw", "tgemm_wave", "reg_xyz", "premul_tap"))]
    assert all(n in exported for n in hooks), [n for n in hooks if n not in exported]
    csrc = os.path.join(ROOT, "patchaugnet_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert not [l for l in mk.splitlines() if l.startswith(("exp:", "exp ", "libpatchaugnet_hip_exp.so:"))], "the Makefile builds a second (experimental) library again"
    assert "PA_EXPERIMENTAL" not in "".join(open(os.path.join(csrc, f)).read() for f in os.listdir(csrc) if f.endswith((".hip", ".h")))


def test_argument_validation_without_gpu(libpath):
    """Bad sizes / null pointers are rejected before any launch, so this runs on a CPU-only box."""
    from patchaugnet_amd import _lib
    l = _lib.lib()
    rc = l.pa_knnquery(1, 0, 4, 4, None, None, None, None, None)
    assert rc == -1 and b"pa_knnquery" in l.pa_last_error()
    rc = l.pa_furthestsampling(1, 16, 4, None, None, None, None)
    assert rc == -1


def test_ops_refuse_cpu_tensors():
    from patchaugnet_amd import pointops
    x = torch.rand(1, 32, 3)
    for fn in (lambda: pointops.furthestsampling(x, 4), lambda: pointops.knnquery(4, x, x),
               lambda: pointops.nearestneighbor(x, x), lambda: pointops.grouping(x.transpose(1, 2).contiguous(), torch.zeros(1, 2, 2, dtype=torch.int32))):
        with pytest.raises(RuntimeError, match="MI355X"):
            fn()


def test_product_package_never_imports_the_oracle():
    code = ("import sys; import patchaugnet_amd, patchaugnet_amd.pointops, patchaugnet_amd.patch_aug_net, patchaugnet_amd.pptnet, "
            "patchaugnet_amd.loupe, patchaugnet_amd.profiling; "
            "bad=[m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]; assert not bad, bad")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "patchaugnet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


@pytest.mark.parametrize("name", ["patch_aug_net", "pptnet"])
def test_state_dict_keys_match_the_reference(name):
    from patchaugnet_amd import configs, patch_aug_net, pptnet
    if name == "patch_aug_net":
        m = patch_aug_net.Network(param=configs.patch_aug_net_config(), use_a2a_recon=True, use_l2_norm=True)
    else:
        m = pptnet.Network(param=configs.pptnet_config(), use_normalize=False)
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", f"{name}_state_dict_keys.json")))
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, (shape, dt) in ref.items():
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == dt, k


@pytest.mark.skipif(not os.path.isdir("/root/reference/libs/pointops"), reason="reference tree only exists in the build container")
def test_reference_op_layer_imports_against_the_mirrors():
    """INTEGRATION.md route 1: the reference's own pointops.py / chamfer_dist / emd_module import cleanly when this repo's
    mirrors are registered under the native module names, and every native function they call exists in the mirror."""
    import re
    import subprocess
    import sys
    code = (
        "import sys; sys.dont_write_bytecode = True; sys.path.insert(0, %r); sys.path.insert(1, '/root/reference')\n"
        "import patchaugnet_amd.pointops_cuda as pc, patchaugnet_amd.chamfer_dist as ch, patchaugnet_amd.emd_module as emd\n"
        "sys.modules['pointops_cuda'] = pc; sys.modules['chamfer'] = ch; sys.modules['emd'] = emd\n"
        "from libs.pointops.functions import pointops\n"
        "import libs.chamfer_dist as cd\n"
        "from libs.emd_module.emd_module import emdModule\n"
        "print('OK', pointops.furthestsampling is not None, cd.ChamferDistanceL1 is not None, emdModule is not None)\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and "OK True True True" in out.stdout, out.stderr[-2000:]
    from patchaugnet_amd import pointops_cuda
    used = set(re.findall(r"pointops_cuda\.(\w+)\(", open("/root/reference/libs/pointops/functions/pointops.py").read()))
    assert used and not (used - set(dir(pointops_cuda)))


def test_bench_refuses_gpu_counts_it_cannot_honour():
    """`python bench.py --gpus 2` on a box without two GPUs must fail loudly, never print a 1-GPU line (N-rank self-launch of bench.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(torch.cuda.device_count(), 1) + 1)],
                         capture_output=True, text=True, env=env)
    assert out.returncode != 0 and "refusing" in out.stderr and '"n_gpus"' not in out.stdout
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert out.returncode != 0 and "does not match WORLD_SIZE" in out.stderr


def test_bench_kernel_regexes_match_the_built_kernels():
    """bench.py finds the dominant kernel and the graded gather in rocprofv3's output by (demangled) name: a template signature change must not
    silently turn roofline.traffic into null.  Checked against the kernel symbols of the built library."""
    import re
    import shutil
    import subprocess
    import bench
    from patchaugnet_amd import _lib
    filt = shutil.which("llvm-cxxfilt") or shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if not os.path.exists(filt) or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("no demangler / library")
    data = open(_lib.LIB_PATH, "rb").read()
    mangled = sorted({m.decode() for m in re.findall(rb"_ZN[0-9A-Za-z_]+(?:chain_kernel|group_lds_kernel|tgemm_cm_kernel)[0-9A-Za-z_]+", data)})
    assert mangled, "no kernel symbols found in the library"
    names = subprocess.run([filt], input="\n".join(mangled), capture_output=True, text=True, check=True).stdout.splitlines()
    assert any(re.search(bench.DOMINANT_KERNEL_RE, n) for n in names), [n for n in names if "chain_kernel<1, 16, 3" in n]
    assert any(re.search(bench.GROUPING_KERNEL_RE, n) for n in names)
    assert any(re.search(bench.TRAIN_DOMINANT_KERNEL_RE, n) for n in names), [n for n in names if "tgemm_cm_kernel<8, 1, 2" in n]
