"""Registers / scratch / occupancy / LDS of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py   (CPU only: cross-compiles)"""
import re
import subprocess
import sys

flags = "-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off --offload-arch=gfx950 -DO3D_MATCH_WAVES=5 -DO3D_SEARCH_WAVES=4".split()
out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, *sys.argv[1:], "-Rpass-analysis=kernel-resource-usage", "-c",
                      "open3d_slam_private_amd/csrc/reg_core.hip", "-o", "/tmp/o3d_rc.o"], capture_output=True, text=True).stderr
keys = {"VGPRs": "vgpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ", "VGPRs Spill": "spill",
        "LDS Size [bytes/block]": "lds"}
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for k, short in keys.items():
        m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", line)
        if m and cur is not None:
            cur[short] = int(m.group(1))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.split("(")[0].replace("void ", "")
    if "rocprim" in name:
        continue
    print(f"{name[:44]:46s} vgpr {r.get('vgpr', 0):4d}  scratch {r.get('scratch', 0):5d}  occupancy {r.get('occ', 0):2d}  "
          f"spilled {r.get('spill', 0):4d}  lds {r.get('lds', 0):6d}")
