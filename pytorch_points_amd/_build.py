"""Builds the two native artefacts in-tree:

  libpp_hip.so   the C-ABI library: hand-written HIP for gfx950, compiled with hipcc.  -ffp-contract=off
                 pins the fp32 rounding sequence to the source (DESIGN.md "Arithmetic contract").
  _pp_torch.so   the host-side bridge (csrc/torch_bridge.cpp): the Chamfer operators as C++
                 torch::autograd::Function nodes that call the C ABI; plain C++ compiled with g++ against the
                 installed torch (no device code in it).

Both are git-ignored but travel to the GPU box with the repo snapshot.
"""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libpp_hip.so")
BRIDGE = os.path.join(HERE, "_pp_torch.so")
BRIDGE_SRC = os.path.join(CSRC, "torch_bridge.cpp")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
               "-shared", "-std=c++17", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile csrc/*.hip -> libpp_hip.so and csrc/torch_bridge.cpp -> _pp_torch.so.  No-op when up to date."""
    if force or is_stale():
        _build_library(force, verbose)
    build_bridge(force=force, verbose=verbose)
    return LIB


def _build_library(force, verbose):
    """one hipcc -c per source, in parallel, into build/ (objects are reused while their source and the headers
    are older), then one link"""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    newest_header = max(os.path.getmtime(h) for h in headers)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            jobs.append([hipcc, *flags, "-I" + INCLUDE, "-I" + CSRC, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, (os.cpu_count() or 2)))) as pool:
        list(pool.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])


BRIDGE_META = BRIDGE + ".meta"


def _bridge_tag():
    """what the bridge was compiled against: it links libtorch and the CPython ABI, so a module built for another
    torch or Python (it travels with a snapshot of the tree) must be rebuilt, not imported"""
    import sys
    import sysconfig

    import torch
    return "torch=%s python=%s abi=%s" % (torch.__version__, sys.version.split()[0], sysconfig.get_config_var("SOABI"))


def bridge_is_stale():
    if not os.path.exists(BRIDGE):
        return True
    t = os.path.getmtime(BRIDGE)
    if any(os.path.getmtime(d) > t for d in (BRIDGE_SRC, os.path.join(INCLUDE, "pp_hip.h"))):
        return True
    return bridge_abi_mismatch()


def bridge_abi_mismatch():
    """the module on disk was built against another torch / Python (or carries no tag): importing it would fail with
    undefined symbols at best"""
    try:
        return open(BRIDGE_META).read().strip() != _bridge_tag()
    except OSError:
        return True


def build_bridge(force=False, verbose=False):
    """g++ torch_bridge.cpp against the installed (ROCm) torch; links libpp_hip.so by $ORIGIN.  ~1 minute."""
    if not force and not bridge_is_stale():
        return BRIDGE
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    tlib = ce.library_paths()[0]
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", BRIDGE_SRC, "-o", BRIDGE,
           "-DTORCH_EXTENSION_NAME=_pp_torch", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-I" + INCLUDE,
           *["-I" + p for p in ce.include_paths()], "-I" + os.path.join(rocm, "include"),
           "-I" + sysconfig.get_paths()["include"], "-L" + tlib, "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10",
           "-lc10_hip", "-ltorch_hip", "-l:librccl.so", "-L" + HERE, "-l:libpp_hip.so", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(BRIDGE_META, "w") as fh:
        fh.write(_bridge_tag() + "\n")
    return BRIDGE


if __name__ == "__main__":
    build(force=True, verbose=True)
