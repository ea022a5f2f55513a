"""Host-side logic and the C-ABI surface, on CPU (no kernel is launched here)."""
import hashlib
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, make_oracle_sim, rel_err
from pienerf_amd import scene


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pienerf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pn_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pienerf_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 20
    h = _lib.lib()  # loads libpienerf_hip.so and resolves every name in SIGNATURES (AttributeError otherwise)
    for n in names:
        assert hasattr(h, n), f"{n} is declared in include/pienerf_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert b"gfx950" in h.pn_version()
    assert h.pn_compact_scratch_ints(1000) >= 4 and h.pn_sim_work_doubles(10, 20) >= 10 * 30 * 4


def test_no_mfma_overwrites_its_own_sources(tmp_path):
    """hipcc (ROCm 7.2) does not mark the destination of v_mfma_f32_32x32x16_bf16 early-clobber; when a source dies in the first MFMA
    of an accumulator chain the allocator may hand its registers to the destination (pn_nerf_forward.hip, split_mac orders the
    products so that it cannot).  Precaution: disassemble the shipped device code and check no MFMA has such an overlap."""
    import shutil
    import subprocess
    from pienerf_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    so = tmp_path / "lib.so"
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, capture_output=True)
    images = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert images
    pat = re.compile(r"v_mfma\S*\s+[av]\[(\d+):(\d+)\], ([av]\[\d+:\d+\]|[av]\d+), ([av]\[\d+:\d+\]|[av]\d+),")
    n = 0

    def span(tok):
        m = re.match(r"[av]\[(\d+):(\d+)\]", tok)
        if m:
            return int(m.group(1)), int(m.group(2))
        r = int(tok[1:])
        return r, r
    for img in images:
        dis = subprocess.run([objdump, "-d", str(tmp_path / img)], check=True, capture_output=True, text=True).stdout
        for line in dis.splitlines():
            m = pat.search(line)
            if not m:
                continue
            n += 1
            d0, d1 = int(m.group(1)), int(m.group(2))
            for tok in (m.group(3), m.group(4)):
                if tok[0] != line[m.start(1) - 2]:  # VGPR vs AGPR files do not alias
                    continue
                s0, s1 = span(tok)
                assert s1 < d0 or s0 > d1, line.strip()
    assert n >= 120  # the fused network kernel alone carries 120


def test_no_packed_fp32_valu_beside_mfma_chains(tmp_path):
    """No packed-fp32 VALU (v_pk_{mul,add,fma}_f32) among the MFMAs of a kernel: beside an MFMA chain each one costs ~20 extra cycles
    (MI355X_MICROARCH.md, "price of one filler beside MFMAs"; measured here too: the bf16 split's subtractions as v_pk_add_f32 made
    k_nerf_forward 1.5 % slower), and -O3's SLP vectoriser inserts them on its own — hence -fno-slp-vectorize (pienerf_amd/build.py).
    The network kernels' hash-grid phase, fenced from the layers by scheduling barriers, does use them (corner weights and channel sums:
    2 % faster) — so the rule is about distance: no packed op within 12 instructions of an MFMA.  Round 1 also blamed packed ops for
    corrupted MFMA results; tools/repro_pk_mfma.hip did not reproduce that (profiles/r02_repro_pk_mfma.json: 0 wrong of 819 M MFMAs), so
    this is a performance rule, not an erratum workaround."""
    import shutil
    import subprocess
    from pienerf_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    so = tmp_path / "lib.so"
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, capture_output=True)
    images = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert images
    n_mfma_kernels = 0
    for img in images:
        dis = subprocess.run([objdump, "-d", str(tmp_path / img)], check=True, capture_output=True, text=True).stdout
        for body in re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", dis):
            if "v_mfma_" not in body:
                continue
            n_mfma_kernels += 1
            lines = body.splitlines()
            mfma = [i for i, ln in enumerate(lines) if "v_mfma_" in ln]
            for i, ln in enumerate(lines):
                if re.search(r"\bv_pk_(mul|add|fma)_f32\b", ln):
                    near = min(abs(i - j) for j in mfma)
                    assert near > 12, (lines[0], ln.strip(), near)
    assert n_mfma_kernels >= 2


def test_ops_fail_loudly_without_gpu_tensors():
    from pienerf_amd import gridencoder, raymarching, shencoder
    with pytest.raises(RuntimeError, match="GPU only"):
        shencoder.sh_encode(torch.zeros(4, 3), 4)
    with pytest.raises(RuntimeError, match="GPU only"):
        gridencoder.grid_encode(torch.zeros(4, 3), torch.zeros(16, 2), torch.tensor([0, 8, 16], dtype=torch.int32), 2.0, 16)
    with pytest.raises(RuntimeError, match="GPU only"):
        raymarching.composite_rays(1, 1, torch.zeros(1, dtype=torch.int32), torch.zeros(1), torch.zeros(1), torch.zeros(1, 3), torch.zeros(1, 2),
                                   torch.zeros(1), torch.zeros(1), torch.zeros(1, 3))
    with pytest.raises(RuntimeError):
        shencoder.sh_encode(torch.zeros(4, 3), 4, True)  # no dy_dx on the inference path


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from pienerf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under pienerf_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pienerf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports oracle"
                assert "liboracle" not in src and "oracle/" not in src.replace("the oracle/", ""), f"{f} references oracle/"


def test_ply_round_trip(tmp_path, small_cloud):
    for binary in (True, False):
        p = tmp_path / f"c{int(binary)}.ply"
        scene.write_ply(str(p), small_cloud, binary=binary)
        back = scene.cloud_from_ply(str(p))
        assert np.array_equal(back["pos"], small_cloud["pos"]) and np.array_equal(back["mass"], small_cloud["mass"])
        assert np.array_equal(back["pin"], small_cloud["pin"].astype(bool))
    assert small_cloud["pin"].sum() > 0 and small_cloud["pin"].sum() < len(small_cloud["pin"]) / 4  # leg tips only


def test_scene_generators_are_deterministic(ckpt):
    c2 = scene.make_checkpoint(bound=1.0, seed=0)
    for k in ("embeddings", "W0", "W1", "W2", "W3", "W4", "density_bitfield"):
        assert hashlib.sha1(c2[k].tobytes()).hexdigest() == hashlib.sha1(ckpt[k].tobytes()).hexdigest()
    assert ckpt["cascade"] == 1 and ckpt["density_bitfield"].shape == (128 ** 3 // 8,)
    occ = np.unpackbits(ckpt["density_bitfield"]).sum()
    assert 0.02 < occ / 128 ** 3 < 0.2  # a chair-sized solid in the unit box
    # occupancy bits are in morton order: the voxel at the seat centre is set, a far corner is not
    m = scene.morton3D([64], [64], [64])[0]
    assert (ckpt["density_bitfield"][m // 8] >> (m % 8)) & 1
    m = scene.morton3D([2], [2], [2])[0]
    assert not (ckpt["density_bitfield"][m // 8] >> (m % 8)) & 1


def test_hashgrid_geometry_matches_encoder_module(ckpt):
    from pienerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048)
    assert np.array_equal(enc.offsets.numpy(), ckpt["offsets"]) and abs(enc.per_level_scale - ckpt["per_level_scale"]) < 1e-12
    assert enc.embeddings.shape == (6119864, 2) and enc.output_dim == 32
    assert set(enc.state_dict().keys()) == {"offsets", "embeddings"}


def test_network_state_dict_keys_match_reference():
    from pienerf_amd.nerf.network import NeRFNetwork
    net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1)
    keys = set(net.state_dict().keys())
    want = {"aabb_train", "aabb_infer", "density_grid", "density_bitfield", "step_counter", "encoder.offsets", "encoder.embeddings",
            "sigma_net.0.weight", "sigma_net.1.weight", "color_net.0.weight", "color_net.1.weight", "color_net.2.weight"}
    assert keys == want
    assert net.sigma_net[0].weight.shape == (64, 32) and net.sigma_net[1].weight.shape == (16, 64)
    assert net.color_net[0].weight.shape == (64, 31) and net.color_net[2].weight.shape == (3, 64)
    assert net.cascade == 1 and net.grid_size == 128


def test_default_options_follow_get_opts():
    o = scene.default_opt()
    assert abs(o["hash_grid_size"] - 0.06) < 1e-12 and o["W"] == o["H"] == 800 and o["bound"] == 1.0 and o["dt_gamma"] == 0.0  # get_opts.py:96,100-105
    assert scene.default_opt(num_seek_IP=7)["num_seek_IP"] == 3 and scene.default_opt(num_seek_IP=0)["num_seek_IP"] == 1     # :97,117-120


def test_simulator_precompute_matches_oracle_init(small_cloud, small_opt, oracle_sim):
    """The product's init (torch, matrix-free GMLS formulation) against the oracle's (numpy, explicit matrices)."""
    from pienerf_amd.simulator.solver import Simulator
    o = small_opt
    s = Simulator(dt=o["sim_dt"], iters=o["sim_iters"], bbox=torch.tensor([2.0 * o["bound"]] * 3), dx=o["sim_dx"], stiff=o["sim_stiff"],
                  base=torch.tensor([-o["bound"]] * 3), device="cpu")
    c = small_cloud
    s.pos, s.mass, s.mu, s.lam = (torch.from_numpy(np.asarray(c[k], np.float64)) for k in ("pos", "mass", "mu", "lam"))
    s.is_pin = torch.from_numpy(c["pin"].astype(bool))
    s.precompute()
    r = oracle_sim
    assert (s.n_IP, s.n_k) == (r.n_IP, r.n_k) and np.array_equal(s.IP_kernel.numpy(), r.IP_kernel.numpy())
    assert np.array_equal(s.kernel_pos.numpy(), r.kernel_pos.numpy()) and np.array_equal(s.IP_pos.numpy(), r.IP_pos.numpy())
    for a, b in ((s.IP_Nx, r.IP_Nx), (s.IP_dNx, r.IP_dNx), (s.IP_ddNx, r.IP_ddNx), (s.pts_Nx, r.pts_Nx), (s.Mmat, r.Mmat)):
        assert rel_err(a.numpy(), b) < 1e-12
    assert rel_err(s.Ainv.numpy(), r.Ainv) < 1e-8
    assert rel_err(s.rhs_gravity.numpy().reshape(-1, 3), r.rhs_gravity) < 1e-13
    assert np.array_equal(s.active_kernels.numpy(), r.active)
    # CSR of (IP, slot) pairs per kernel: ascending, complete
    buf, bg, cnt = s.buffer.numpy(), s.kernel_bg.numpy(), s.kernel_cnt.numpy()
    assert cnt.sum() == 8 * s.n_IP and np.array_equal(np.sort(buf), np.arange(8 * s.n_IP))
    topo = s.IP_kernel.numpy().reshape(-1)
    for k in (0, s.n_k // 2, s.n_k - 1):
        seg = buf[bg[k]:bg[k] + cnt[k]]
        assert np.all(topo[seg] == k) and np.all(np.diff(seg) > 0)
    # the cell form's layout (pn_sim_stepforward_cells): every (point, slot) pair exactly once, a chunk's points share their 8 kernels, and the data flow of
    # k_cells_elastic_gather restated in numpy on this layout — per-chunk partial sums of P dNx added per kernel in kp_list order — is collect_rhs_IP
    cl = s._cells
    B, nch = cl["B"], cl["n_chunks"]
    src, valid, tab = cl["src"].numpy(), cl["valid"].numpy(), cl["tab"].numpy()
    assert B == 8 * (B // 8) and valid.sum() == s.n_IP and np.array_equal(np.sort(src[valid]), np.arange(s.n_IP))
    assert np.array_equal(tab[:, 0], valid.sum(1)) and tab[:, 0].min() >= 1 and tab[:, 0].max() <= B
    tk = s.IP_kernel.numpy()
    for c in range(nch):
        assert np.all(tk[src[c][valid[c]]] == tab[c, 1:9][None, :])
    kp_bg, kp_list = cl["kp_bg"].numpy(), cl["kp_list"].numpy()
    assert kp_bg[-1] == 8 * nch and np.array_equal(np.sort(kp_list), np.arange(8 * nch))
    for k in range(s.n_k):
        seg = kp_list[kp_bg[k]:kp_bg[k + 1]]
        assert len(seg) >= 1 and np.all(np.diff(seg) > 0) and np.all(tab[seg // 8, 1 + seg % 8] == k)
    rng = np.random.default_rng(3)
    P = rng.standard_normal((s.n_IP, 3, 3))
    g = cl["dNx"].numpy().reshape(nch, B // 8, 15, 8, 8, 2).transpose(0, 1, 3, 4, 2, 5).reshape(nch, B, 8, 3, 10)   # [chunk][point][slot][c][x]
    assert np.array_equal(g[valid], s.IP_dNx.numpy()[src[valid]]) and not g[~valid].any()
    part = np.einsum("cprk,cpskx->csxr", np.where(valid[:, :, None, None], P[src], 0.0), g)                         # [chunk][slot][x][r]
    rhs = np.stack([part.reshape(nch * 8, 10, 3)[kp_list[kp_bg[k]:kp_bg[k + 1]]].sum(0) for k in range(s.n_k)])
    want = np.zeros((s.n_k, 10, 3))
    np.add.at(want, tk.reshape(-1), np.einsum("prk,pskx->psxr", P, s.IP_dNx.numpy()).reshape(-1, 10, 3))
    assert rel_err(rhs, want) < 1e-13
    # the reference's (30 n_k)^2 views
    assert s.global_matrix.shape == (s.n_k * 30, s.n_k * 30)
    x = torch.randn(s.n_k * 30, dtype=torch.float64)
    assert torch.allclose(s.global_matrix @ x, (s.Ainv @ x.view(-1, 3)).reshape(-1), atol=1e-9 * float(s.Ainv.abs().max()) * 100)
    assert s.step == s.stepforward


def test_shim_backends_match_the_reference_binding_signatures():
    """shim/_raymarching.py, _gridencoder.py, _shencoder.py (the modules the reference's wrappers import as `_backend`) export every function of the
    reference's pybind modules with the same parameter names in the same order — tests/golden/ref_bindings.json was read from the reference's own
    headers by tests/golden/make_golden_ref.py."""
    import importlib
    import inspect
    import json
    shim = os.path.join(ROOT, "shim")
    sys.path.insert(0, shim)
    try:
        with open(os.path.join(ROOT, "tests", "golden", "ref_bindings.json")) as f:
            ref = json.load(f)
        assert set(ref) == {"_raymarching", "_gridencoder", "_shencoder"}
        for mod, fns in ref.items():
            m = importlib.import_module(mod)
            assert m.__file__.startswith(shim)
            for name, params in fns.items():
                got = list(inspect.signature(getattr(m, name)).parameters)
                assert got == params, (mod, name, got, params)
        # the wrapper packages resolve under the reference's names
        for pkg, attr in (("raymarching", "march_rays_quadratic_bending"), ("gridencoder", "GridEncoder"), ("shencoder", "SHEncoder"), ("simulator.solver", "Simulator")):
            mod = importlib.import_module(pkg)
            assert mod.__file__.startswith(shim) and hasattr(mod, attr)
    finally:
        sys.path.remove(shim)
        for k in [k for k in sys.modules if k.split(".")[0] in ("_raymarching", "_gridencoder", "_shencoder", "raymarching", "gridencoder", "shencoder", "simulator")]:
            del sys.modules[k]


def test_persistent_substep_plan_limits_are_host_side_checks():
    """pn_sim_coop_bytes is pure host arithmetic (no HIP call): which scenes the persistent substep takes, without a GPU."""
    from pienerf_amd._lib import lib
    L = lib()
    assert int(L.pn_sim_coop_bytes(139, 3576, 248)) > 0          # the chair on 248 workgroups
    assert int(L.pn_sim_coop_bytes(42, 1208, 256)) > 0           # the trex cloud
    assert int(L.pn_sim_coop_bytes(343, 20000, 256)) == 0        # 3430 unknowns per component > 2048 register-resident columns
    assert int(L.pn_sim_coop_bytes(139, 3576, 4)) == 0           # 894 integration points per workgroup
    assert int(L.pn_sim_coop_bytes(139, 3576, 512)) == 0         # more workgroups than the barrier's groups are sized for
    assert int(L.pn_sim_coop_bytes(0, 10, 256)) == 0


def test_render_opts_mirror_follows_the_header():
    """pienerf_amd._lib.RenderOpts is the ctypes mirror of pn_render_opts (include/pienerf_hip.h): same field names in the same order, same
    element types — a field added on one side only would shift every later one silently."""
    import ctypes as C
    from pienerf_amd._lib import RenderOpts
    text = open(os.path.join(ROOT, "include", "pienerf_hip.h")).read()
    body = text[:text.index("} pn_render_opts;")]
    body = re.sub(r"/\*.*?\*/", "", body[body.rindex("typedef struct {"):], flags=re.S)
    decl = re.findall(r"\b(int|float|uint32_t)\s+(\w+)(?:\[(\d+)\])?\s*;", body)
    ctype = {"int": C.c_int32, "float": C.c_float, "uint32_t": C.c_uint32}
    want = [(name, ctype[t] * int(n) if n else ctype[t]) for t, name, n in decl]
    got = list(RenderOpts._fields_)
    assert [n for n, _ in got] == [n for n, _ in want]
    for (n, a), (_, b) in zip(got, want):
        assert C.sizeof(a) == C.sizeof(b) and getattr(a, "_type_", a) == getattr(b, "_type_", b), n
    assert C.sizeof(RenderOpts) == sum(C.sizeof(t) for _, t in want)
