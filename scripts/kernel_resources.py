"""register / LDS / occupancy table of the kernels in a hipcc -Rpass-analysis=kernel-resource-usage report (stderr of
`hipcc ... -Rpass-analysis=kernel-resource-usage -c file.hip`): python scripts/kernel_resources.py report.txt [filter]"""
import re
import subprocess
import sys

PATS = dict(VGPR=r"VGPRs: (\d+)", AGPR=r"AGPRs: (\d+)", scratch=r"ScratchSize \[bytes/lane\]: (\d+)",
            occ=r"Occupancy \[waves/SIMD\]: (\d+)", LDS=r"LDS Size \[bytes/block\]: (\d+)")
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip().split()[0]
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in d:
        continue
    vals = {k: (re.search(p, b) or [None, "?"])[1] for k, p in PATS.items()}
    short = re.sub(r"\(anonymous namespace\)::", "", d)
    short = re.sub(r"^void ", "", short)
    m = re.match(r"([\w:]+(<[^()]*>)?)", short)
    short = (m.group(1) if m else short)[:90]
    print(f"{short:90s} " + " ".join(f"{k} {v:>5s}" for k, v in vals.items()))
