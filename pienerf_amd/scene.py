"""Synthetic inputs for the simulate-and-render path.

The reference's assets (``chair_0.ply``, ``model/chair/checkpoints/ngp_ep0300.pth``) are
hosted off-repo and absent, so every test and benchmark runs on a
procedural stand-in built here, deterministically from a numpy seed:

  * point cloud   — lattice samples of a box-union "chair" with the vertex attributes the
                    simulator reads (x,y,z,mass,mu,lam,pin; solver.py:116-135, README.md:98-108);
  * checkpoint    — the inference state-dict keys of SURVEY.md §5: hash-grid ``offsets`` /
                    ``embeddings`` (gridencoder/grid.py:96-134), the five bias-free MLP weights
                    (nerf/network.py:36-71) and a morton-ordered ``density_bitfield``
                    (nerf/renderer.py:94-111, raymarching.cu:1398-1399) rasterised from the solid;
  * camera        — ``OrbitCamera`` pose / intrinsics (nerf/gui.py:13-44);
  * PLY IO        — reader/writer for the one-element vertex PLY (``plyfile`` is not installed).

Pure numpy: no GPU, no torch, no oracle.
"""
import math
import struct

import numpy as np

# ------------------------------------------------------------------ solid
# (xmin, xmax, ymin, ymax, zmin, zmax), y up
CHAIR_BOXES = np.array([
    [-0.40, 0.40, -0.05, 0.05, -0.40, 0.40],   # seat
    [-0.40, 0.40, 0.05, 0.60, -0.40, -0.30],   # back
    [-0.40, -0.30, -0.60, -0.05, -0.40, -0.30],  # legs
    [0.30, 0.40, -0.60, -0.05, -0.40, -0.30],
    [-0.40, -0.30, -0.60, -0.05, 0.30, 0.40],
    [0.30, 0.40, -0.60, -0.05, 0.30, 0.40],
    [-0.40, -0.30, 0.20, 0.28, -0.30, 0.30],   # arm rests
    [0.30, 0.40, 0.20, 0.28, -0.30, 0.30],
    [-0.40, -0.30, 0.05, 0.20, 0.20, 0.30],    # arm supports
    [0.30, 0.40, 0.05, 0.20, 0.20, 0.30],
], dtype=np.float64) * 1.4  # the blender-synthetic chair at scale 0.8 spans roughly +-0.8 of the bound-1 box


def chair_solid(p, margin=0.0, boxes=CHAIR_BOXES):
    """Boolean inside-test for points p [...,3] against the box union grown by `margin`."""
    p = np.asarray(p, np.float64)
    inside = np.zeros(p.shape[:-1], bool)
    for b in boxes:
        inside |= ((p[..., 0] >= b[0] - margin) & (p[..., 0] <= b[1] + margin) & (p[..., 1] >= b[2] - margin) & (p[..., 1] <= b[3] + margin)
                   & (p[..., 2] >= b[4] - margin) & (p[..., 2] <= b[5] + margin))
    return inside


def make_chair_points(sub_res=60, bound=1.0, lam=1e6, mu=1e6, density=1e3, pin_height=0.05, boxes=CHAIR_BOXES, hgs=0.06):
    """Lattice point cloud like main_sample.py's uniform stage (main_sample.py:214-223): centres of a
    sub_res^3 lattice over [-bound,bound]^3 that fall inside the solid.  vp = hgs^3 / count_in_cell
    (main_sample.py:181-200); mass = density * vp (README.md:106-108); pin = lowest `pin_height` of the solid."""
    h = 2.0 * bound / sub_res
    c = -bound + (np.arange(sub_res) + 0.5) * h
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    pts = np.stack([X, Y, Z], -1).reshape(-1, 3)
    pts = pts[chair_solid(pts, boxes=boxes)]
    cell = np.floor((pts + bound) / hgs).astype(np.int64)
    key = (cell[:, 0] * 100003 + cell[:, 1]) * 100003 + cell[:, 2]
    _, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    vp = hgs ** 3 / cnt[inv]
    ymin = boxes[:, 2].min()
    return dict(pos=pts, vp=vp, mass=density * vp, mu=np.full(len(pts), mu), lam=np.full(len(pts), lam),
                pin=(pts[:, 1] < ymin + pin_height).astype(np.int32))


# ------------------------------------------------------------------ PLY
_PLY_PROPS = [("x", "f8"), ("y", "f8"), ("z", "f8"), ("mass", "f8"), ("mu", "f8"), ("lam", "f8"), ("pin", "i4")]
_PLY_TYPES = {"f8": "double", "f4": "float", "i4": "int", "u1": "uchar", "i2": "short", "u2": "ushort", "u4": "uint", "i1": "char"}
_PLY_NAMES = {"double": "f8", "float64": "f8", "float": "f4", "float32": "f4", "int": "i4", "int32": "i4", "uchar": "u1", "uint8": "u1",
              "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "uint": "u4", "uint32": "u4", "char": "i1", "int8": "i1"}


def write_ply(path, cloud, binary=True, props=None):
    """Vertex-only PLY.  Default: the simulator's input attributes (solver.py:116-135); props=("x","y","z"): positions only, the
    schema Simulator.OutputToPly writes (solver.py:109-113)."""
    n = len(cloud["pos"])
    sel = [(k, t) for k, t in _PLY_PROPS if props is None or k in props]
    rec = np.zeros(n, dtype=[(k, "<" + t) for k, t in sel])
    rec["x"], rec["y"], rec["z"] = cloud["pos"][:, 0], cloud["pos"][:, 1], cloud["pos"][:, 2]
    for k, _ in sel[3:]:
        rec[k] = cloud[k]
    hdr = ["ply", "format binary_little_endian 1.0" if binary else "format ascii 1.0", f"element vertex {n}"]
    hdr += [f"property {_PLY_TYPES[t]} {k}" for k, t in sel] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        if binary:
            f.write(rec.tobytes())
        else:
            for r in rec:
                f.write((" ".join(repr(float(v)) if isinstance(v, np.floating) else str(int(v)) for v in r) + "\n").encode())


def read_ply(path):
    """Reads the first (vertex) element of an ascii / binary_little_endian PLY into a dict of columns."""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply", "not a PLY file"
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline().decode().strip()
            if line == "end_header":
                break
            tok = line.split()
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = (n == 0 and not props)  # only the first element is read (plydata.elements[0])
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                assert tok[1] != "list", "list properties are not supported"
                props.append((tok[2], _PLY_NAMES[tok[1]]))
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(n)]
            cols = {k: np.array([r[i] for r in rows], dtype=np.float64).astype(t) for i, (k, t) in enumerate(props)}
        elif fmt == "binary_little_endian":
            rec = np.frombuffer(f.read(n * sum(np.dtype(t).itemsize for _, t in props)), dtype=[(k, "<" + t) for k, t in props], count=n)
            cols = {k: np.array(rec[k]) for k, _ in props}
        else:
            raise ValueError(f"unsupported PLY format {fmt}")
    return cols


def cloud_from_ply(path):
    c = read_ply(path)
    missing = [k for k in ("mass", "mu", "lam", "pin") if k not in c]
    if missing:
        raise ValueError(f"{path}: not a simulator input PLY, vertex properties {missing} are missing (OutputToPly files hold x, y, z only)")
    return dict(pos=np.stack([c["x"], c["y"], c["z"]], 1).astype(np.float64), mass=c["mass"].astype(np.float64), mu=c["mu"].astype(np.float64),
                lam=c["lam"].astype(np.float64), pin=c["pin"].astype(bool))


# ------------------------------------------------------------------ checkpoint
def _expand_bits(v):
    v = (v * 0x00010001) & 0xFF0000FF
    v = (v * 0x00000101) & 0x0F00F00F
    v = (v * 0x00000011) & 0xC30C30C3
    v = (v * 0x00000005) & 0x49249249
    return v


def morton3D(x, y, z):
    x, y, z = (np.asarray(a, np.uint64) for a in (x, y, z))
    return (_expand_bits(x) | (_expand_bits(y) << np.uint64(1)) | (_expand_bits(z) << np.uint64(2))).astype(np.uint32)


def hashgrid_offsets(bound=1.0, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, input_dim=3):
    """gridencoder/grid.py:96-134 with desired_resolution = 2048*bound (nerf/network.py:35)."""
    per_level_scale = np.exp2(np.log2(2048 * bound / base_resolution) / (num_levels - 1))
    offsets, offset = [], 0
    for i in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(2 ** log2_hashmap_size, (res + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, np.int32), float(per_level_scale)


def make_density_bitfield(bound=1.0, grid_size=128, solid=chair_solid):
    """cascade x grid_size^3 occupancy bits in morton order (bit i%8 of byte i/8).  Level c covers
    [-min(2^c,bound), +...]^3 (raymarching.cu:1389-1396).  A cell is occupied when its centre is inside the
    solid grown by half a cell diagonal-ish margin (trained density grids are slightly fat)."""
    cascade = 1 + math.ceil(math.log2(bound))
    H = grid_size
    bits = np.zeros(cascade * H ** 3 // 8, np.uint8)
    idx = np.arange(H)
    X, Y, Z = np.meshgrid(idx, idx, idx, indexing="ij")
    mort = morton3D(X.ravel(), Y.ravel(), Z.ravel()).astype(np.int64)
    for c in range(cascade):
        mb = min(2.0 ** c, bound)
        cs = 2.0 * mb / H
        ctr = (np.stack([X, Y, Z], -1).reshape(-1, 3) + 0.5) * cs - mb
        occ = solid(ctr, margin=cs)
        lin = c * H ** 3 + mort[occ]
        np.bitwise_or.at(bits, lin // 8, (1 << (lin % 8)).astype(np.uint8))
    return bits, cascade


def make_checkpoint(bound=1.0, seed=0, sigma_target=60.0, grid_size=128, solid=chair_solid, shaped=False, sigma_outside=0.02):
    """Random-init network of the reference architecture with an analytically calibrated density:

    feature 0 (level 0, channel 0) is the constant 0.5 (level 0 is dense, so trilinear interpolation
    returns it everywhere), hidden unit 0 = ReLU(2 * 0.5) = 1 and sigma_net[1].weight[0,0] = ln(sigma_target),
    so sigma = sigma_target * exp(small random term) — median ~ sigma_target without a forward pass.

    shaped=True (point sampling, pienerf_amd/sampling.py): the density FIELD itself has the solid's shape, not only the bitfield —
    channel 0 of the finest dense level holds the solid's vertex occupancy, hidden unit 1 = its trilinear interpolation, and
    sigma = sigma_outside * (sigma_target / sigma_outside)^occupancy * exp(small random term).
    """
    rng = np.random.default_rng(seed)
    offsets, pls = hashgrid_offsets(bound)
    n = int(offsets[-1])
    emb = rng.uniform(-0.5, 0.5, size=(n, 2)).astype(np.float32)
    emb[offsets[0]:offsets[1], 0] = 0.5

    def w(o, i):
        return (rng.standard_normal((o, i)) * math.sqrt(2.0 / i)).astype(np.float32)

    W0, W1, W2, W3, W4 = w(64, 32), w(16, 64), w(64, 31), w(64, 64), w(3, 64)
    W0[0, :] = 0.0
    W0[0, 0] = 2.0
    W1[0, :] *= 0.25
    W1[0, 0] = math.log(sigma_target)
    if shaped:
        S = np.float32(np.log2(pls))
        lvl = max(l for l in range(16) if (int(np.ceil(np.float32(np.exp2(np.float32(l) * S) * 16 - 1))) + 2) ** 3 <= offsets[l + 1] - offsets[l])
        scale = np.float32(np.exp2(np.float32(lvl) * S) * 16 - 1)                    # gridencoder.cu:133
        r1 = int(np.ceil(scale)) + 2                                                 # resolution + 1 vertices per axis (:134, :75)
        gidx = np.arange(r1)
        X, Y, Z = np.meshgrid(gidx, gidx, gidx, indexing="ij")
        P = (np.stack([X, Y, Z], -1) - 0.5) / float(scale) * (2 * bound) - bound     # vertex g sits at u = (g - 0.5) / scale
        emb[offsets[lvl] + (X + Y * r1 + Z * r1 * r1).reshape(-1), 0] = solid(P.reshape(-1, 3)).astype(np.float32)
        W0[1, :] = 0.0
        W0[1, 2 * lvl] = 1.0
        W1[0, 0] = math.log(sigma_outside)
        W1[0, 1] = math.log(sigma_target / sigma_outside)
    bits, cascade = make_density_bitfield(bound, grid_size, solid)
    return dict(embeddings=emb, offsets=offsets, per_level_scale=pls, base_resolution=16, W0=W0, W1=W1, W2=W2, W3=W3, W4=W4,
                density_bitfield=bits, cascade=cascade, grid_size=grid_size, bound=float(bound), min_near=0.2, density_scale=1.0)


# ------------------------------------------------------------------ camera
def orbit_pose(radius=5.0, azimuth_deg=0.0, elevation_deg=0.0, center=(0.0, 0.0, 0.0)):
    """OrbitCamera.pose (nerf/gui.py:29-39) for rot = R_y(az) R_x(el) * from_quat([1,0,0,0])."""
    res = np.eye(4, dtype=np.float32)
    res[2, 3] -= radius
    base = np.diag([1.0, -1.0, -1.0])
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    Ry = np.array([[math.cos(az), 0, math.sin(az)], [0, 1, 0], [-math.sin(az), 0, math.cos(az)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(el), -math.sin(el)], [0, math.sin(el), math.cos(el)]])
    rot = np.eye(4, dtype=np.float32)
    rot[:3, :3] = (Ry @ Rx @ base).astype(np.float32)
    res = rot @ res
    res[:3, 3] -= np.asarray(center, np.float32)
    return res.astype(np.float32)


def _rotvec_matrix(v):
    """Rotation matrix of the rotation vector v (Rodrigues) — what scipy's Rotation.from_rotvec(v).as_matrix() returns."""
    v = np.asarray(v, np.float64)
    th = float(np.linalg.norm(v))
    if th < 1e-300:
        return np.eye(3)
    k = v / th
    K = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + math.sin(th) * K + (1.0 - math.cos(th)) * (K @ K)


class OrbitCamera:
    """The GUI camera of the reference (nerf/gui.py:13-61) without scipy: same state (rot, radius, center, up), same mouse-delta
    conventions for orbit / scale / pan, same pose and intrinsics.  tests/test_golden_ref.py holds it to values returned by the
    reference class itself."""

    def __init__(self, W, H, r=2, fovy=60):
        self.W, self.H = W, H
        self.radius = r
        self.fovy = fovy
        self.center = np.array([0, 0, 0], dtype=np.float32)
        self.rot = np.diag([1.0, -1.0, -1.0])  # R.from_quat([1, 0, 0, 0]): half a turn about x (ngp convention)
        self.up = np.array([0, 1, 0], dtype=np.float32)

    def pose_to_params(self, pose):
        self.radius = -self.center[2] + pose[:3, 3][2]
        self.rot = np.asarray(pose[:3, :3], np.float64)

    @property
    def pose(self):
        res = np.eye(4, dtype=np.float32)
        res[2, 3] -= self.radius
        rot = np.eye(4, dtype=np.float32)
        rot[:3, :3] = self.rot
        res = rot @ res
        res[:3, 3] -= self.center
        return res

    @property
    def intrinsics(self):
        return orbit_intrinsics(self.W, self.H, self.fovy)

    def orbit(self, dx, dy):
        side = self.rot[:3, 0]
        self.rot = _rotvec_matrix(self.up * np.radians(-0.1 * dx)) @ _rotvec_matrix(side * np.radians(-0.1 * dy)) @ self.rot

    def scale(self, delta):
        self.radius *= 1.1 ** (-delta)

    def pan(self, dx, dy, dz=0):
        self.center += (0.0005 * self.rot[:3, :3] @ np.array([dx, dy, dz])).astype(np.float32)


def orbit_intrinsics(W, H, fovy=50.0):
    """OrbitCamera.intrinsics (nerf/gui.py:41-44)."""
    focal = H / (2 * np.tan(np.radians(fovy) / 2))
    return np.array([focal, focal, W // 2, H // 2], dtype=np.float64)


# ------------------------------------------------------------------ options
def default_opt(**over):
    """The option names render_deformed / Simulator read (get_opts.py), chair-demo values (README.md:123)."""
    opt = dict(bound=1.0, scale=0.8, dt_gamma=0.0, W=800, H=800, max_steps=1024, T_thresh=1e-2, min_near=0.2, density_thresh=10, bg_radius=-1,
               radius=5.0, fovy=50.0, max_iter_num=1, num_seek_IP=3, sim_dt=1e-2, sim_dx=0.05, sim_iters=10, sim_stiff=1e5, cut=False,
               cut_bounds=[0.0, 2.0, -2.0, 1.0, -1.42, 0.92], timing_on=False,
               fp16=False)  # the GUI / main_render path renders in fp32: main_gui.py:36 builds its Trainer without fp16= although -O sets opt.fp16
    opt.update(over)
    opt["hash_grid_size"] = 1.2 * opt["sim_dx"]
    opt["num_seek_IP"] = max(min(3, opt["num_seek_IP"]), 1)
    return opt


def trex_opt(**over):
    """The option set of the reference's second demo (README.md:134: trex, llff): bound 2 (get_opts default), scale 0.33, dt_gamma 1/128, two
    cascades, --cut with its bounds, max_steps 300, T_thresh 5e-2, 1008 x 756, num_seek_IP 1.  tests/test_golden_ref.py compares it with the
    namespace the reference's get_opts.py returns for that command line."""
    opt = dict(bound=2.0, scale=0.33, dt_gamma=1.0 / 128, W=1008, H=756, max_steps=300, T_thresh=5e-2, num_seek_IP=1, max_iter_num=1, cut=True,
               cut_bounds=[-0.62, 1.0, -0.82, 0.42, -0.52, 0.28], sim_dx=0.05)
    opt.update(over)
    return default_opt(**opt)


def stress_opt(**over):
    """BASELINE.json configs[4] ("stress"): the dense sub_res = 180 point cloud, max_iter_num = 5 Newton iterations of the inverse warp,
    num_seek_IP = 3, the 800 x 800 frame rendered in ray batches of 4096 (max_ray_batch, get_opts.py:24), and the network under autocast:
    fp16 hash tables (gridencoder/grid.py:43-44) + fp16 MFMA layers (main_train.py:52's Trainer(fp16=opt.fp16))."""
    opt = dict(max_iter_num=5, num_seek_IP=3, fp16=True, max_ray_batch=4096, sub_res=180)
    opt.update(over)
    return default_opt(**opt)
