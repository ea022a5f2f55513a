"""ORACLE — TEST INFRASTRUCTURE ONLY.  Builds `oracle/_ref/`: the REFERENCE'S OWN render kernels, compiled for gfx950.

What this does.  The reference's three native extensions are plain CUDA + ATen:
    /root/reference/raymarching/src/{raymarching.cu,raymarching.h,bindings.cpp}
    /root/reference/gridencoder/src/{gridencoder.cu,gridencoder.h,bindings.cpp}
    /root/reference/shencoder/src/{shencoder.cu,shencoder.h,bindings.cpp}
Their sources are read WHERE THEY LIE (never written to, never copied into this repository): a scratch copy is made
under a temporary directory outside the repository, torch's own source translator (`torch.utils.hipify`, what
`torch.utils.cpp_extension` applies to every `.cu` file on a ROCm build of torch — API renames only, cuda* -> hip*) is
run on that copy, and the result is compiled with `hipcc --offload-arch=gfx950 -std=c++17` against the torch headers.
Only the resulting shared objects are written, into `oracle/_ref/` (git-ignored, but NOT gpurun-ignored: like the
product's own `.so` it travels to the GPU box, where `/root/reference` does not exist).  No stand-in header, library or
generated file is involved: the translation units compile as they are.

ONE source line does not exist on HIP and is left out of the scratch copy (nothing is put in its place):
    gridencoder.cu:330  `atomicAdd((__half2*)&grad_grid[index + c], v);`  — CUDA's __half2 atomicAdd has no HIP
    counterpart.  The statement sits in a run-time branch taken only when scalar_t == at::Half, so every float
    instantiation behaves exactly as written; the at::Half instantiation of grid_encode_BACKWARD in this build is
    invalid and is never called (the tests use float tensors; fp16 training is out of scope).

Each extension is built twice:
    _ref_<name>.so     -ffp-contract=off   every float operation rounds once, in source order — the semantics the CPU
                                           restatement (`oracle/render_oracle.cpp`, same flag) and the product's
                                           bit-exact kernels are written to.  Bit-for-bit agreement with THIS build
                                           pins "restatement == reference source".
    _ref_<name>_fma.so hipcc's default     (-ffp-contract=fast: a*b+c fused wherever the compiler likes), the analogue
                                           of nvcc's default -fmad=true the reference's own binary is built with.  The
                                           mismatch rate against this build measures what contraction alone changes; it
                                           is reported by the tests, not hidden (nvcc's own choices stay unpinnable).

The modules are TEST-ONLY: `tests/test_gpu_ref.py` and nothing else loads them (they are torch extensions running
the reference's kernels on the GPU — a checker, never the product path and never the thing measured).

Usage:  python -m oracle.ref_build          (in the build container; needs /root/reference)
        oracle.ref_build.load("raymarching", fma=False)  -> the imported module, or None when the .so is absent
"""
import importlib.util
import os
import shutil
import sys
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference"
OUT_DIR = os.path.join(_HERE, "_ref")
EXTS = {
    "raymarching": ["raymarching.cu", "bindings.cpp", "raymarching.h"],
    "gridencoder": ["gridencoder.cu", "bindings.cpp", "gridencoder.h"],
    "shencoder": ["shencoder.cu", "bindings.cpp", "shencoder.h"],
}


# (file, exact statement) pairs removed from the scratch copy because HIP has no such intrinsic; each must match exactly once
OMIT = {"gridencoder": [("gridencoder.cu", "atomicAdd((__half2*)&grad_grid[index + c], v);")]}


def so_path(ext, fma=False):
    return os.path.join(OUT_DIR, f"_ref_{ext}{'_fma' if fma else ''}.so")


def available():
    return os.path.isdir(REF_ROOT) and all(os.path.exists(os.path.join(REF_ROOT, e, "src", f)) for e, fs in EXTS.items() for f in fs)


def _stale(ext, fma):
    out = so_path(ext, fma)
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(REF_ROOT, ext, "src", f) for f in EXTS[ext]]
    return any(os.path.getmtime(d) > t for d in deps)


def build_one(ext, fma=False, verbose=False, extra=(), out=None):
    """hipify (scratch copy, outside the repository) + hipcc; writes only oracle/_ref/_ref_<ext>[_fma].so.
    `extra` / `out`: additional device-compile flags and another output path, for experiments (tools/)."""
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    from torch.utils import cpp_extension

    name = f"_ref_{ext}{'_fma' if fma else ''}"
    scratch = tempfile.mkdtemp(prefix=f"pn_ref_{ext}_")
    try:
        src = os.path.join(scratch, "src")
        os.makedirs(src)
        for f in EXTS[ext]:
            shutil.copyfile(os.path.join(REF_ROOT, ext, "src", f), os.path.join(src, f))
        for f, stmt in OMIT.get(ext, []):
            text = open(os.path.join(src, f)).read()
            assert text.count(stmt) == 1, (f, stmt, text.count(stmt))
            open(os.path.join(src, f), "w").write(text.replace(stmt, ""))
        bdir = os.path.join(scratch, "build")
        os.makedirs(bdir)
        contract = [] if fma else ["-ffp-contract=off"]
        cpp_extension.load(
            name=name,
            sources=[os.path.join(src, f) for f in EXTS[ext] if not f.endswith(".h")],
            extra_cflags=["-O2", "-std=c++17"],
            # the reference's own nvcc flags (`<ext>/backend.py`: -O3, -U__CUDA_NO_HALF_OPERATORS__, -U__CUDA_NO_HALF_CONVERSIONS__,
            # -U__CUDA_NO_HALF2_OPERATORS__) under their HIP names; -std=c++17 because torch 2.10's headers need it
            # -fno-strict-return: `__device__ double minus(...)` (raymarching.cu:930-934) has no return statement.  nvcc compiles the fall-off as a
            # plain return; clang treats it as unreachable at -O1+ and deletes what follows (the kernel then reads garbage IP ids and trips
            # its own `assert(IPs[k] < n_vtx)` — observed on the first GPU run).  The flag gives clang nvcc's behaviour; the source is untouched.
            extra_cuda_cflags=["-O3", "-std=c++17", "--offload-arch=gfx950", "-U__HIP_NO_HALF_OPERATORS__", "-U__HIP_NO_HALF_CONVERSIONS__",
                               "-U__HIP_NO_HALF2_OPERATORS__", "-fno-strict-return"] + contract + list(extra),
            build_directory=bdir,
            verbose=verbose,
            is_python_module=False,
        )
        os.makedirs(OUT_DIR, exist_ok=True)
        shutil.copyfile(os.path.join(bdir, name + ".so"), out or so_path(ext, fma))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    return out or so_path(ext, fma)


def build(force=False, verbose=False):
    """Build every variant that is missing or older than its sources.  No-op (returns []) when /root/reference is absent."""
    if not available():
        return []
    built = []
    for ext in EXTS:
        for fma in (False, True):
            if force or _stale(ext, fma):
                built.append(build_one(ext, fma, verbose))
    return built


def load(ext, fma=False):
    """Import oracle/_ref/_ref_<ext>[_fma].so (a torch extension with the reference's pybind functions); None if not built."""
    path = so_path(ext, fma)
    if not os.path.exists(path):
        return None
    import torch  # noqa: F401  (the extension links against libtorch)

    name = os.path.basename(path)[:-3]
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules[name] = mod
    return mod


if __name__ == "__main__":
    outs = build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print("built:" if outs else "nothing to build", *outs)
