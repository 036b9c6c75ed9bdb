"""Registers / spills / LDS per kernel of the engine (hipcc -Rpass-analysis=kernel-resource-usage): python tools/kernel_regs.py [filter] [-Dflags...]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else ""
flags = [a for a in sys.argv[1:] if a.startswith("-")]
p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
                    "-Rpass-analysis=kernel-resource-usage", os.path.join(root, "popvae_amd/csrc/pvae_engine.hip"), "-o", "/dev/null"] + flags,
                   capture_output=True, text=True)
cur = {}
rows = []
for line in p.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?): (.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt in name:
        print(f"{name:60s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>4s} spill {r.get('VGPRs Spill','?'):>3s} sgpr-spill {r.get('SGPRs Spill','?'):>3s} "
              f"occ {r.get('Occupancy [waves/SIMD]','?')} LDS {r.get('LDS Size [bytes/block]','?')}")
