"""Per-kernel register / LDS / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py dffinthewild_amd/csrc/dffw_conv_tile.hip [substring filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-DDFFW_TILE_PREC=0", "-c", src,
                      "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("void dffw::", "")}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':58s} VGPR AGPR SGPR  LDS    occ spill")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name']:58s} {r.get('VGPRs','?'):>4} {r.get('AGPRs','?'):>4} {r.get('TotalSGPRs','?'):>4} {r.get('LDS Size','?'):>6} {r.get('Occupancy','?'):>4} "
              f"{r.get('VGPRs Spill','0')}/{r.get('SGPRs Spill','0')}")
