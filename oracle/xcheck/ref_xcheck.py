#!/usr/bin/env python3
"""Cross-check O2 of SURVEY.md §8c: run the REFERENCE'S OWN kernel bodies on the golden inputs and record what
they produce, so that tests/test_oracle.py can hold the CPU oracle (O1) against it.

    python oracle/xcheck/ref_xcheck.py          # build container only: needs /root/reference

What it does: reads the `__global__` kernels of the reference by line range from /root/reference AT RUN TIME
(their text is never stored in this repository: the generated C++ file and the binary live under a temporary
directory and are deleted), compiles them with g++ against oracle/xcheck/cuda_shim.h -- a CPU emulation of the
CUDA execution model -- twice (-ffp-contract=off, and -ffp-contract=fast -mfma: the contraction of the
reference's `a*a+b*b+c*c` and `d += t*t` is nvcc's choice), runs them on the inputs of
tests/golden/*.npz and writes tests/golden/ref_xcheck.npz: OUTPUT ARRAYS ONLY (indices, distances, gradients).

What it is NOT: a build of the reference, and not evidence that pins parity: the kernels run on stand-ins for
the CUDA runtime, and one of them is patched (the FPS kernel's shared-memory race, SURVEY.md F9: a
`__syncthreads()` is inserted after `old=dists_i[0]`, the barrier-separated semantics the source intends).
DESIGN.md §3 keeps saying "parity unpinned".  The value of the exercise: if the line-by-line restatement in
oracle/pp_oracle.c misread the reference anywhere (a comparison direction, a tie rule, an index expression),
the two would disagree here.
"""
import ctypes
import glob
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/pytorch_points/_ext"
GOLDEN = os.path.join(ROOT, "tests", "golden")

# (file, first line, last line) of every kernel on the path -- SURVEY.md §2.3
RANGES = [
    ("nmdistance_cuda.cu", 5, 5),      # const int BATCH = 512;
    ("nmdistance_cuda.cu", 7, 49),     # K1 NmDistanceKernel
    ("nmdistance_cuda.cu", 55, 115),   # K2 LabeledNmDistanceKernel
    ("nmdistance_cuda.cu", 167, 185),  # K3 NmDistanceGradKernel
    ("sampling_cuda.cu", 9, 25),       # K4 gather_points_kernel_fast
    ("sampling_cuda.cu", 47, 64),      # K5 gather_points_grad_kernel_fast
    ("sampling_cuda.cu", 162, 233),    # K6 furthest_point_sampling_forward_kernel
    ("sampling_cuda.cu", 340, 376),    # K7 ball_query_kernel_fast
    ("sampling_cuda.cu", 447, 467),    # K8 group_points_kernel
    ("sampling_cuda.cu", 482, 503),    # K9 group_points_grad_kernel
    ("interpolate_gpu.cu", 9, 52),     # K10 three_nn_kernel_fast
    ("interpolate_gpu.cu", 77, 97),    # K11 three_interpolate_kernel_fast
    ("interpolate_gpu.cu", 120, 142),  # K12 three_interpolate_grad_kernel_fast
]


def generate(tmp):
    parts = ['#include "%s"' % os.path.join(HERE, "cuda_shim.h"), '#include "%s"' % os.path.join(REF, "cuda_utils.h")]
    for fname, a, b in RANGES:
        lines = open(os.path.join(REF, fname)).read().split("\n")[a - 1:b]
        text = "\n".join(lines)
        # the dynamic shared array is the shim's global buffer
        text = "\n".join(ln for ln in text.split("\n") if "extern __shared__" not in ln)
        if fname == "sampling_cuda.cu" and a == 162:
            assert "old=dists_i[0];" in text
            text = text.replace("old=dists_i[0];", "old=dists_i[0]; __syncthreads(); /* F9: see ref_xcheck.py */", 1)
        parts.append(text)
    parts.append('#include "%s"' % os.path.join(HERE, "driver.inc"))
    src = os.path.join(tmp, "ref_kernels_generated.cpp")
    open(src, "w").write("\n".join(parts) + "\n")
    return src


def build(tmp, src, tag, flags):
    so = os.path.join(tmp, "libref_%s.so" % tag)
    subprocess.run(["g++", "-O2", "-std=c++20", "-pthread", "-fPIC", "-shared", *flags, src, "-o", so], check=True)
    return ctypes.CDLL(so)


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def run_all(L, out, tag):
    f32, i32 = np.float32, np.int32
    for path in sorted(glob.glob(os.path.join(GOLDEN, "chamfer_*.npz"))):
        name = os.path.basename(path)[:-4]
        g = np.load(path)
        x1, x2 = np.ascontiguousarray(g["xyz1"]), np.ascontiguousarray(g["xyz2"])
        b, n, c = x1.shape
        m = x2.shape[1]
        d1, d2 = np.zeros((b, n), f32), np.zeros((b, m), f32)
        i1, i2 = np.zeros((b, n), i32), np.zeros((b, m), i32)
        L.x_chamfer_forward(P(x1), P(x2), b, n, m, c, P(d1), P(i1), P(d2), P(i2))
        g1, g2 = np.zeros_like(x1), np.zeros_like(x2)
        gd1, gd2 = np.ascontiguousarray(g["graddist1"]), np.ascontiguousarray(g["graddist2"])
        L.x_chamfer_backward(P(x1), P(x2), P(gd1), P(gd2), P(i1), P(i2), b, n, m, c, P(g1), P(g2))
        for k, v in (("dist1", d1), ("idx1", i1), ("dist2", d2), ("idx2", i2), ("gradxyz1", g1), ("gradxyz2", g2)):
            out["%s/%s/%s" % (tag, name, k)] = v
        print(tag, name, "ok", flush=True)
    g = np.load(os.path.join(GOLDEN, "labeled_b1_n512_m700.npz"))
    x1, x2 = np.ascontiguousarray(g["xyz1"]), np.ascontiguousarray(g["xyz2"])
    l1, l2 = np.ascontiguousarray(g["label1"].astype(f32)), np.ascontiguousarray(g["label2"].astype(f32))
    b, n, c = x1.shape
    m = x2.shape[1]
    d1, d2 = np.zeros((b, n), f32), np.zeros((b, m), f32)
    i1, i2 = np.zeros((b, n), i32), np.zeros((b, m), i32)
    L.x_labeled_chamfer_forward(P(x1), P(x2), P(l1), P(l2), b, n, m, c, P(d1), P(i1), P(d2), P(i2))
    for k, v in (("dist1", d1), ("idx1", i1), ("dist2", d2), ("idx2", i2)):
        out["%s/labeled_b1_n512_m700/%s" % (tag, k)] = v
    for path in sorted(glob.glob(os.path.join(GOLDEN, "fps_*.npz"))):
        name = os.path.basename(path)[:-4]
        g = np.load(path)
        x = np.ascontiguousarray(g["xyz"])
        b, n, _ = x.shape
        mm = g["idx"].shape[1]
        temp = np.full((b, n), 1e10, f32)
        idx = np.zeros((b, mm), i32)
        L.x_furthest_sampling(P(x), P(temp), P(idx), b, n, mm, int(g["seed"]))
        out["%s/%s/idx" % (tag, name)] = idx
        out["%s/%s/temp" % (tag, name)] = temp
        print(tag, name, "ok", flush=True)
    g = np.load(os.path.join(GOLDEN, "ball_query_b2_n2048_m256.npz"))
    x, ctr = np.ascontiguousarray(g["xyz"]), np.ascontiguousarray(g["new_xyz"])
    b, n, _ = x.shape
    mm = ctr.shape[1]
    for key in g.files:
        if key.startswith("idx_r"):
            r = float(key.split("_")[1][1:])
            ns = int(key.split("_")[2][2:])
            idx = np.zeros((b, mm, ns), i32)
            L.x_ball_query(P(ctr), P(x), P(idx), b, n, mm, ctypes.c_float(r), ns)
            out["%s/ball_query_b2_n2048_m256/%s" % (tag, key)] = idx
    # group / gather / interpolate on the ball-query indices and synthetic features
    sys.path.insert(0, ROOT)
    from pytorch_points_amd import synthetic as S
    idx = out["%s/ball_query_b2_n2048_m256/idx_r0.2_ns16" % tag]
    cfe = 6
    feats = S.normal(900, (b, cfe, n))
    grouped = np.zeros((b, cfe, mm, 16), f32)
    L.x_group_points(P(feats), P(idx), P(grouped), b, cfe, n, mm, 16)
    gout = S.normal(901, (b, cfe, mm, 16))
    ggrad = np.zeros((b, cfe, n), f32)
    L.x_group_points_grad(P(gout), P(idx), P(ggrad), b, cfe, n, mm, 16)
    out["%s/group_points/out" % tag] = grouped
    out["%s/group_points/grad" % tag] = ggrad
    gi = np.ascontiguousarray(idx[:, :, 0])
    gath = np.zeros((b, cfe, mm), f32)
    L.x_gather(P(feats), P(gi), P(gath), b, cfe, n, mm)
    gg = np.zeros((b, cfe, n), f32)
    go = S.normal(902, (b, cfe, mm))
    L.x_gather_grad(P(go), P(gi), P(gg), b, cfe, n, mm)
    out["%s/gather/out" % tag] = gath
    out["%s/gather/grad" % tag] = gg
    for path in sorted(glob.glob(os.path.join(GOLDEN, "three_nn_*.npz"))):
        name = os.path.basename(path)[:-4]
        g = np.load(path)
        u, k = np.ascontiguousarray(g["unknown"]), np.ascontiguousarray(g["known"])
        b2, n2, _ = u.shape
        m2 = k.shape[1]
        d2 = np.zeros((b2, n2, 3), f32)
        ti = np.zeros((b2, n2, 3), i32)
        L.x_three_nn(P(u), P(k), P(d2), P(ti), b2, n2, m2)
        out["%s/%s/dist2" % (tag, name)] = d2
        out["%s/%s/idx" % (tag, name)] = ti
        if m2 >= 3:
            w = S.uniform01(903, (b2, n2, 3)).astype(f32).reshape(b2, n2, 3)
            pts = S.normal(904, (b2, cfe, m2))
            o = np.zeros((b2, cfe, n2), f32)
            L.x_three_interpolate(P(pts), P(ti), P(w), P(o), b2, cfe, m2, n2)
            gin = S.normal(905, (b2, cfe, n2))
            gp = np.zeros((b2, cfe, m2), f32)
            L.x_three_interpolate_grad(P(gin), P(ti), P(w), P(gp), b2, cfe, n2, m2)
            out["%s/%s/interp" % (tag, name)] = o
            out["%s/%s/interp_grad" % (tag, name)] = gp


def main():
    if not os.path.isdir(REF):
        raise SystemExit("ref_xcheck.py: %s is not here (build container only); tests/golden/ref_xcheck.npz is the "
                         "committed result" % REF)
    tmp = tempfile.mkdtemp(prefix="pp_ref_xcheck_")
    try:
        src = generate(tmp)
        out = {}
        for tag, flags in (("nocontract", ["-ffp-contract=off"]), ("fma", ["-ffp-contract=fast", "-mfma"])):
            run_all(build(tmp, src, tag, flags), out, tag)
        np.savez_compressed(os.path.join(GOLDEN, "ref_xcheck.npz"), **out)
        print("wrote tests/golden/ref_xcheck.npz:", len(out), "arrays")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
