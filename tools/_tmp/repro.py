import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pienerf_amd import scene
from pienerf_amd.harness import SimRenderHarness
def hh(t):
    return hashlib.md5(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:10]
h = SimRenderHarness(scene.default_opt(), device="cuda:0")
s = h.sim
print("Ainv", hh(s.Ainv), "Mmat", hh(s.Mmat), "dNx", hh(s.IP_dNx), "rhs_rest", hh(s.rhs_rest) if hasattr(s, "rhs_rest") else "-")
for i in range(20):
    s.stepforward()
    if i in (0, 1, 19):
        print("dof", i, hh(s.dof))
with torch.no_grad():
    out = h.step(simulate=True)
    torch.cuda.synchronize()
    print("dof after", hh(s.dof), "image", hh(out["image"]), "stats", h.model.last_stats if hasattr(h.model, "last_stats") else None)
    out = h.step(simulate=False)
    torch.cuda.synchronize()
    print("image again", hh(out["image"]))
