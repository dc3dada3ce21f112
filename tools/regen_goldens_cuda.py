#!/usr/bin/env python3
"""Regenerate the cross-check fixture from the REAL reference on an NVIDIA box.

Parity of this repository's oracle is "unpinned": the reference (yifita/pytorch_points) ships no tests or golden
vectors and cannot be built in the MI355X image (needs nvcc, the CUDA runtime, THC headers, cuSOLVER).  What
`tests/golden/ref_xcheck.npz` holds today was produced by the reference's kernel bodies run through a CPU stand-in
for the CUDA execution model (oracle/xcheck/ref_xcheck.py) -- good hygiene, not a pin.  This script closes the
gap for anyone who has an NVIDIA GPU:

    # on a CUDA machine, in an environment where the reference is BUILT AND INSTALLED (its own setup.py):
    #     git clone https://github.com/yifita/pytorch_points && cd pytorch_points && pip install .
    python /path/to/this/repo/tools/regen_goldens_cuda.py --out ref_cuda.npz

It imports the reference's own extension modules (`pytorch_points._ext.losses`, `pytorch_points._ext.sampling`),
feeds them the inputs of tests/golden/*.npz (data files of this repository) and writes the SAME schema as
ref_xcheck.npz under the tag `cuda/` (`cuda/<fixture>/<array>`), plus a `provenance` entry (JSON: GPU, CUDA
runtime, torch, the reference package's file hash and -- if it is a git checkout -- its commit).  Copy the result
to tests/golden/ref_cuda.npz; tests/test_oracle.py::test_oracle_matches_real_reference picks it up when present
(indices equal; distances within 2 ulp: the FMA contraction of `a*a+b*b+c*c` is nvcc's choice).

Nothing of the reference is copied or stored by this script: it calls an installed package and records outputs.
It never runs on the MI355X box (no CUDA there) and is not imported by the product or the tests.
"""
import argparse
import glob
import hashlib
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _load_synthetic():
    """tests' counter-based generators, loaded by path: importing the package would put this repository's
    `pytorch_points` alias in front of the real reference"""
    spec = importlib.util.spec_from_file_location("pp_synthetic", os.path.join(ROOT, "pytorch_points_amd", "synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _import_reference():
    # the repository root must not shadow the installed reference (it holds a `pytorch_points` alias package)
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != ROOT]
    import torch
    if not torch.cuda.is_available() or torch.version.cuda is None:
        raise SystemExit("regen_goldens_cuda.py needs an NVIDIA GPU and a CUDA build of torch")
    import pytorch_points
    if os.path.abspath(os.path.dirname(pytorch_points.__file__)).startswith(ROOT):
        raise SystemExit("`import pytorch_points` resolved to this repository's alias, not to the reference")
    from pytorch_points._ext import losses, sampling
    return torch, pytorch_points, losses, sampling


def _provenance(torch, pkg):
    pdir = os.path.dirname(pkg.__file__)
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(pdir, "_ext", "*"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    commit = None
    try:
        commit = subprocess.run(["git", "-C", pdir, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        pass
    nvcc = None
    try:
        nvcc = subprocess.run(["nvcc", "--version"], capture_output=True, text=True, timeout=10).stdout.strip().split("\n")[-1]
    except Exception:
        pass
    return {"gpu": torch.cuda.get_device_name(0), "capability": list(torch.cuda.get_device_capability(0)),
            "cuda_runtime": torch.version.cuda, "torch": torch.__version__, "nvcc": nvcc,
            "nvcc_flags": "the reference's setup.py defaults (CUDAExtension, -O2; --fmad left at nvcc's default true)",
            "reference_package_dir": pdir, "reference_commit": commit, "reference_ext_sha256": h.hexdigest(),
            "schema": "cuda/<fixture>/<array>, arrays as in tests/golden/ref_xcheck.npz"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="ref_cuda.npz")
    args = ap.parse_args()
    S = _load_synthetic()
    torch, pkg, losses, sampling = _import_reference()
    dev = torch.device("cuda:0")
    f32, i32 = np.float32, np.int32
    out = {}
    tag = "cuda"

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def H(t):
        return t.detach().cpu().numpy()

    # Chamfer forward / backward: _ext/nmdistance.cpp:30-34 (outputs allocated by the caller, as model_loss.py:412-433)
    for path in sorted(glob.glob(os.path.join(GOLDEN, "chamfer_*.npz"))):
        name = os.path.basename(path)[:-4]
        g = np.load(path)
        x1, x2 = T(g["xyz1"]), T(g["xyz2"])
        b, n, _ = x1.shape
        m = x2.shape[1]
        d1, d2 = torch.zeros(b, n, device=dev), torch.zeros(b, m, device=dev)
        i1 = torch.zeros(b, n, dtype=torch.int32, device=dev)
        i2 = torch.zeros(b, m, dtype=torch.int32, device=dev)
        losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
        g1, g2 = torch.zeros_like(x1), torch.zeros_like(x2)
        losses.nmdistance_backward(x1, x2, g1, g2, T(g["graddist1"]), T(g["graddist2"]), i1, i2)
        torch.cuda.synchronize()
        for k, v in (("dist1", d1), ("idx1", i1), ("dist2", d2), ("idx2", i2), ("gradxyz1", g1), ("gradxyz2", g2)):
            out["%s/%s/%s" % (tag, name, k)] = H(v)
        print(name, "ok", flush=True)
    g = np.load(os.path.join(GOLDEN, "labeled_b1_n512_m700.npz"))
    x1, x2 = T(g["xyz1"]), T(g["xyz2"])
    l1, l2 = T(g["label1"].astype(f32)), T(g["label2"].astype(f32))
    b, n, _ = x1.shape
    m = x2.shape[1]
    d1, d2 = torch.zeros(b, n, device=dev), torch.zeros(b, m, device=dev)
    i1 = torch.zeros(b, n, dtype=torch.int32, device=dev)
    i2 = torch.zeros(b, m, dtype=torch.int32, device=dev)
    losses.labeled_nmdistance_forward(x1, x2, l1, l2, d1, d2, i1, i2)
    torch.cuda.synchronize()
    for k, v in (("dist1", d1), ("idx1", i1), ("dist2", d2), ("idx2", i2)):
        out["%s/labeled_b1_n512_m700/%s" % (tag, k)] = H(v)
    # FPS: _ext/sampling.cpp furthest_sampling(m, seedIdx, input, temp, idx)
    for path in sorted(glob.glob(os.path.join(GOLDEN, "fps_*.npz"))):
        name = os.path.basename(path)[:-4]
        g = np.load(path)
        x = T(g["xyz"])
        b, n, _ = x.shape
        mm = g["idx"].shape[1]
        temp = torch.full((b, n), 1e10, device=dev)
        idx = torch.zeros(b, mm, dtype=torch.int32, device=dev)
        sampling.furthest_sampling(mm, int(g["seed"]), x, temp, idx)
        torch.cuda.synchronize()
        out["%s/%s/idx" % (tag, name)] = H(idx)
        out["%s/%s/temp" % (tag, name)] = H(temp)
        print(name, "ok", flush=True)
    g = np.load(os.path.join(GOLDEN, "ball_query_b2_n2048_m256.npz"))
    x, ctr = T(g["xyz"]), T(g["new_xyz"])
    b, n, _ = x.shape
    mm = ctr.shape[1]
    for key in g.files:
        if key.startswith("idx_r"):
            r = float(key.split("_")[1][1:])
            ns = int(key.split("_")[2][2:])
            out["%s/ball_query_b2_n2048_m256/%s" % (tag, key)] = H(sampling.ball_query(ctr, x, r, ns)).astype(i32)
    idx = T(out["%s/ball_query_b2_n2048_m256/idx_r0.2_ns16" % tag])
    cfe = 6
    feats = T(S.normal(900, (b, cfe, n)))
    out["%s/group_points/out" % tag] = H(sampling.group_points(feats, idx))
    out["%s/group_points/grad" % tag] = H(sampling.group_points_grad(T(S.normal(901, (b, cfe, mm, 16))), idx, n))
    gi = idx[:, :, 0].contiguous()
    gath = torch.zeros(b, cfe, mm, device=dev)
    sampling.gather_forward(b, cfe, n, mm, feats, gi, gath)
    gg = torch.zeros(b, cfe, n, device=dev)
    sampling.gather_backward(b, cfe, n, mm, T(S.normal(902, (b, cfe, mm))), gi, gg)
    out["%s/gather/out" % tag] = H(gath)
    out["%s/gather/grad" % tag] = H(gg)
    for path in sorted(glob.glob(os.path.join(GOLDEN, "three_nn_*.npz"))):
        name = os.path.basename(path)[:-4]
        g = np.load(path)
        u, k = T(g["unknown"]), T(g["known"])
        b2, n2, _ = u.shape
        m2 = k.shape[1]
        d2 = torch.zeros(b2, n2, 3, device=dev)
        ti = torch.zeros(b2, n2, 3, dtype=torch.int32, device=dev)
        sampling.three_nn_wrapper(b2, n2, m2, u, k, d2, ti)
        out["%s/%s/dist2" % (tag, name)] = H(d2)
        out["%s/%s/idx" % (tag, name)] = H(ti)
        if m2 >= 3:
            w = T(S.uniform01(903, (b2, n2, 3)).astype(f32).reshape(b2, n2, 3))
            pts = T(S.normal(904, (b2, cfe, m2)))
            o = torch.zeros(b2, cfe, n2, device=dev)
            sampling.three_interpolate_wrapper(b2, cfe, m2, n2, pts, ti, w, o)
            gp = torch.zeros(b2, cfe, m2, device=dev)
            sampling.three_interpolate_grad_wrapper(b2, cfe, n2, m2, T(S.normal(905, (b2, cfe, n2))), ti, w, gp)
            out["%s/%s/interp" % (tag, name)] = H(o)
            out["%s/%s/interp_grad" % (tag, name)] = H(gp)
    torch.cuda.synchronize()
    out["provenance"] = np.frombuffer(json.dumps(_provenance(torch, pkg), indent=1).encode(), dtype=np.uint8)
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, ":", len(out) - 1, "arrays + provenance")


if __name__ == "__main__":
    main()
