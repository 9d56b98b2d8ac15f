"""Build the REFERENCE's own torch binding layer against libpatchaugnet_hip.so (build container only; test infrastructure).

INTEGRATION.md route 2: libs/pointops/src/pointops_api.cpp and the eight */*_cuda.cpp files of the reference are compiled from where
they lie under /root/reference -- WITHOUT its .cu files -- by torch's own CUDAExtension flow (which hipifies host sources on ROCm, as it
would for the reference's setup.py) and linked against this repo's library, whose group-2 symbols (include/patchaugnet_hip.h) are the
launcher names those files call.  Sources are copied to a scratch directory under /tmp for the build; only the resulting module goes
to oracle/_ref/ (git-ignored, travels to the GPU box), where tests/test_gpu_boundary.py imports it as `pointops_cuda`, calls every
function with the reference's pre-allocated-output convention and compares with the oracle.  Nothing in the product imports it.

Usage: python -m oracle.build_ref        (no-op when /root/reference is absent or the module is up to date)
"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

REF_SRC = "/root/reference/libs/pointops/src"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "oracle", "_ref")

SETUP = '''
import glob
from setuptools import setup
from torch.utils.cpp_extension import BuildExtension, CUDAExtension
lib = {lib!r}
setup(name="pointops_cuda", ext_modules=[CUDAExtension("pointops_cuda", ["src/pointops_api.cpp"] + sorted(glob.glob("src/*/*_cuda.cpp")),
      library_dirs=[lib], libraries=["patchaugnet_hip"], extra_link_args=["-Wl,-rpath,$ORIGIN/../../patchaugnet_amd/csrc"])],
      cmdclass={{"build_ext": BuildExtension}})
'''


def up_to_date():
    outs = glob.glob(os.path.join(OUT, "pointops_cuda*.so"))
    if not outs:
        return False
    srcs = glob.glob(os.path.join(REF_SRC, "*.cpp")) + glob.glob(os.path.join(REF_SRC, "*", "*_cuda.cpp")) + glob.glob(os.path.join(REF_SRC, "*", "*.h"))
    return os.path.getmtime(outs[0]) >= max(os.path.getmtime(s) for s in srcs)


def build(verbose=False):
    if not os.path.isdir(REF_SRC):
        return None
    if up_to_date():
        return glob.glob(os.path.join(OUT, "pointops_cuda*.so"))[0]
    tmp = tempfile.mkdtemp(prefix="pa_route2_", dir="/tmp")
    try:
        for pat in ("pointops_api.cpp", "cuda_utils.h", "*/*_cuda.cpp", "*/*_cuda_kernel.h"):
            for f in glob.glob(os.path.join(REF_SRC, pat)):
                dst = os.path.join(tmp, "src", os.path.relpath(f, REF_SRC))
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                shutil.copy(f, dst)
        with open(os.path.join(tmp, "setup.py"), "w") as fh:
            fh.write(SETUP.format(lib=os.path.join(ROOT, "patchaugnet_amd", "csrc")))
        res = subprocess.run([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=tmp, text=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             env=dict(os.environ, PYTORCH_ROCM_ARCH="gfx950", MAX_JOBS="8"))
        if verbose or res.returncode:
            print(res.stdout[-3000:])
        if res.returncode:
            raise RuntimeError("building the reference binding layer against libpatchaugnet_hip.so failed")
        os.makedirs(OUT, exist_ok=True)
        so = glob.glob(os.path.join(tmp, "pointops_cuda*.so"))[0]
        dst = os.path.join(OUT, os.path.basename(so))
        shutil.copy(so, dst)
        subprocess.run(["strip", "--strip-debug", dst], check=False)       # -g objects: 18 MB -> ~1 MB to carry to the GPU box
        return dst
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
