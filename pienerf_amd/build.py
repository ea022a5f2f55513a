"""Builds libpienerf_hip.so (hand-written HIP, gfx950 only) with hipcc.

    python -m pienerf_amd.build [--force] [--save-temps]

hipcc cross-compiles without a GPU.  The library is built in-tree (pienerf_amd/lib/) so that it
travels with the source tree; `*.so` is git-ignored.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libpienerf_hip.so")
TEMPS = os.path.join(HERE, "..", "build", "temps")
ARCH = "gfx950"

# -fno-slp-vectorize: -O3 does not pack adjacent scalar fp32 adds / multiplies into v_pk_{add,mul,fma}_f32 on its own.  Beside MFMA chains
# (the network kernels) each packed op costs ~20 extra cycles (MI355X_MICROARCH.md, fillers beside MFMAs), and the bit-exact ray-side code
# wants the operation order of its source.  (Packed fp32 written explicitly for the march's candidate scan — quads of candidates in SoA
# form, two distances per instruction — was measured and is no faster: the padding it needs costs what the packing saves.  The network kernels' hash-grid
# phase, fenced from the MFMA layers by scheduling barriers, does use explicit v_pk_mul_f32 / v_pk_fma_f32 for the corner weights and channel sums: 2 %.)  Round 1 justified the flag with a suspected gfx950 erratum (packed VALU corrupting another wave's
# bf16 MFMA); tools/repro_pk_mfma.hip did not reproduce it (profiles/r02_repro_pk_mfma.json) and the claim is withdrawn.
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-Wall",
          "-Wno-unused-function", "-Wno-unused-variable"]
# per-translation-unit flags: the ray-side kernels round every operation once, in source order (bit-exact integer
# decisions vs the CPU oracle); the encoder/MLP and the fp64 simulator let the compiler contract to FMA.
UNITS = {
    "pn_render_ops.hip": ["-ffp-contract=off"],
    "pn_train_ops.hip": ["-ffp-contract=off"],
    "pn_grid_state.hip": ["-ffp-contract=off"],
    "pn_nerf_forward.hip": ["-ffp-contract=fast"],
    "pn_encoder_grad.hip": ["-ffp-contract=fast"],
    "pn_grid_nd.hip": ["-ffp-contract=fast"],
    "pn_sim.hip": ["-ffp-contract=fast"],
    "pn_copier.hip": [],  # host code only: frame copies through the HSA runtime (links libhsa-runtime64)
}


def source_hash():
    """SHA-1 over the kernel sources, headers and compile flags: identifies the code a measurement was taken on (profiles/pmc_traffic.json is
    stamped with it; bench.py refuses a stamp that does not match the tree it runs from)."""
    import hashlib
    h = hashlib.sha1()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) 
    for f in files:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(HERE, "..", "include", "pienerf_hip.h"), "rb") as fh:
        h.update(fh.read())
    h.update(repr((COMMON, sorted(UNITS.items()))).encode())
    return h.hexdigest()


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libpienerf_hip.so cannot be built (there is no CPU fallback)")


def _stale(out, deps):
    return (not os.path.exists(out)) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force=False, save_temps=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "pienerf_hip.h"),
                                                                                     os.path.abspath(__file__)]
    cc = hipcc()
    objs = []
    for src, extra in UNITS.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [cc] + COMMON + extra + ["-c", s, "-o", o]
            cwd = OBJ
            if save_temps:  # intermediates (.hipi/.bc/.s) go to <repo>/build/temps: git-ignored AND gpurun-ignored, never beside the objects
                cwd = TEMPS
                os.makedirs(TEMPS, exist_ok=True)
                cmd += ["-save-temps=cwd", "-Rpass-analysis=kernel-resource-usage"]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True, cwd=cwd)
    if force or _stale(LIB, objs):
        rocm = os.environ.get("ROCM_PATH") or os.path.dirname(os.path.dirname(os.path.realpath(cc)))  # <rocm>/bin/hipcc
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-L" + os.path.join(rocm, "lib"), "-lhsa-runtime64", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True))
