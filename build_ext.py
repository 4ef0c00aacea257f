#!/usr/bin/env python3
"""Build the native extension ``pytorch_distributed_training_amd/_C.so`` in-tree.

* ``csrc/kernels/*.hip``  -> gfx950 device code, compiled directly with ``hipcc --offload-arch=gfx950``
  (no torch headers: a kernel TU compiles in seconds).
* ``csrc/runtime/*.cpp``  -> host-only C++ (torch/pybind11 bindings, layer schedule, bucket reducer).

Objects are cached under ``build/`` keyed by source + header mtimes, compiled in parallel, and linked
against the torch libraries WITHOUT an rpath into /opt/rocm, so that the HIP runtime torch already
loaded is the one the kernels run on.
"""
from __future__ import annotations

import argparse
import hashlib
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "pytorch_distributed_training_amd")
OUT = os.path.join(PKG, "_C.so")
BUILD = os.path.join(ROOT, "build")
ARCH = "gfx950"


def _torch_paths():
    import torch
    from torch.utils import cpp_extension as ce

    tdir = os.path.dirname(torch.__file__)
    inc = ce.include_paths() + ["/opt/rocm/include"]
    import sysconfig

    inc.append(sysconfig.get_paths()["include"])
    return inc, os.path.join(tdir, "lib"), int(torch._C._GLIBCXX_USE_CXX11_ABI)


def _sources():
    kdir, rdir = os.path.join(ROOT, "csrc", "kernels"), os.path.join(ROOT, "csrc", "runtime")
    hip = sorted(os.path.join(kdir, f) for f in os.listdir(kdir) if f.endswith(".hip"))
    cpp = sorted(os.path.join(rdir, f) for f in os.listdir(rdir) if f.endswith(".cpp"))
    hdr = sorted(os.path.join(d, f) for d in (kdir, rdir) for f in os.listdir(d) if f.endswith((".h", ".hpp")))
    return hip, cpp, hdr


# attention.hip writes its packed fp32 math as explicit vector operations; the SLP vectoriser on top of that pairs up the
# scalar adds of the DPP row reductions (v_mov_b32_dpp + v_pk_add_f32, 6 instructions per 4 values and step) where the
# un-paired adds fold into v_add_f32_dpp (4 instructions)
PER_SOURCE_FLAGS = {"attention.hip": ["-fno-slp-vectorize"]}


def _stamp(paths, flags):
    h = hashlib.sha1(" ".join(flags).encode())
    for p in paths:
        st = os.stat(p)
        h.update(f"{p}:{st.st_mtime_ns}:{st.st_size}".encode())
    return h.hexdigest()


def _compile(src, obj, cmd, stamp, verbose):
    stamp_file = obj + ".stamp"
    if os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return src, 0.0, True
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"compile failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return src, time.time() - t0, False


def build(jobs: int = 8, verbose: bool = False, force: bool = False, kernel_defines=(), variant: str = "") -> str:
    """Build the extension.  ``variant`` + ``kernel_defines`` (``-D`` macros of the kernel sources, e.g.
    ``PDT_PHASES=4``) build an A/B copy ``_C_<variant>.so`` next to ``_C.so`` with its own object cache;
    ``PDT_NATIVE_LIB=<path>`` makes the package load it instead of the default build."""
    global BUILD, OUT
    if variant:
        BUILD, OUT = os.path.join(ROOT, "build", "variant_" + variant), os.path.join(PKG, f"_C_{variant}.so")
    os.makedirs(BUILD, exist_ok=True)
    inc, tlib, abi = _torch_paths()
    hip, cpp, hdr = _sources()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    dev_flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
    dev_flags += [f"-D{d}" for d in kernel_defines]
    host_flags = ["-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                  "-DTORCH_EXTENSION_NAME=_C", "-DTORCH_API_INCLUDE_EXTENSION_H",
                  f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-Wno-deprecated-declarations"] + [f"-I{p}" for p in inc]
    tasks = []
    for s in hip:
        obj = os.path.join(BUILD, os.path.basename(s) + ".o")
        flags = dev_flags + PER_SOURCE_FLAGS.get(os.path.basename(s), [])
        cmd = [hipcc] + flags + ["-c", s, "-o", obj]
        tasks.append((s, obj, cmd, _stamp([s] + hdr, flags)))
    for s in cpp:
        obj = os.path.join(BUILD, os.path.basename(s) + ".o")
        cmd = [hipcc] + host_flags + ["-c", s, "-o", obj]
        # the host runtime only sees the plain-C launch interface of the kernels, not their device headers
        host_hdr = [h for h in hdr if h.endswith("api.h") or os.sep + "runtime" + os.sep in h]
        tasks.append((s, obj, cmd, _stamp([s] + host_hdr, host_flags)))
    if force:
        for _, obj, _, _ in tasks:
            if os.path.exists(obj + ".stamp"):
                os.remove(obj + ".stamp")
    rebuilt = False
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        for src, dt, cached in ex.map(lambda t: _compile(*t, verbose), tasks):
            rebuilt |= not cached
            if verbose or not cached:
                print(f"[build_ext] {'cached ' if cached else 'compiled'} {os.path.relpath(src, ROOT)} ({dt:.1f}s)")
    objs = [t[1] for t in tasks]
    if rebuilt or not os.path.exists(OUT):
        link = [os.environ.get("CXX", "g++"), "-shared", "-o", OUT] + objs + [f"-L{tlib}", "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python",
                                                      "-lc10_hip", "-ltorch_hip", "-lamdhip64", "-lrccl"]
        r = subprocess.run(link, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed: {' '.join(link)}\n{r.stdout}\n{r.stderr}")
        print(f"[build_ext] linked {os.path.relpath(OUT, ROOT)}")
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", "--jobs", type=int, default=min(8, os.cpu_count() or 8))
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-D", dest="defines", action="append", default=[], help="macro for the kernel sources (A/B builds)")
    ap.add_argument("--variant", default="", help="build _C_<variant>.so instead of _C.so")
    a = ap.parse_args()
    if a.defines and not a.variant:
        ap.error("-D needs --variant: the default build always uses the defaults in the sources")
    build(a.jobs, a.verbose, a.force, a.defines, a.variant)
