"""Register / spill / scratch summary of the kernels in a HIP translation unit (reads the code-object metadata of an -S dump).

    python tools/kernel_regs.py pienerf_amd/csrc/pn_render_ops.hip [name-filter]
"""
import re
import subprocess
import sys
import tempfile

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pienerf_amd import build


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    import os
    unit = os.path.basename(src)
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        cmd = [build.hipcc()] + build.COMMON + build.UNITS.get(unit, []) + ["-S", "--cuda-device-only", "-o", f.name, src]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(f.name).read()
        if len(sys.argv) > 3:
            open(sys.argv[3], "w").write(text)
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        name, body = m.group(1), m.group(2)
        if flt not in name:
            continue
        g = lambda k: re.search(r"\." + k + r":\s+(\d+)", body)
        vals = {k: int(g(k).group(1)) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size") if g(k)}
        print(name[:60].ljust(60), vals)


if __name__ == "__main__":
    main()
