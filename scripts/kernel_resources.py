"""register / LDS / occupancy table of the kernels in a hipcc -Rpass-analysis=kernel-resource-usage report (stderr of
`hipcc ... -Rpass-analysis=kernel-resource-usage -c file.hip`): python scripts/kernel_resources.py report.txt [filter]

    python scripts/kernel_resources.py --check report.txt      (run by cn-rma_amd/csrc/Makefile on every build of sparse.hip)

--check is the build-time guard of the gather-once convolution kernels: their weight fragments are fetched by
inline-asm loads with hand-counted s_waitcnt, and hipcc treats the destination registers as defined right behind the asm -- a
spill (scratch store of a register whose load has not landed) silently corrupts the operands.  The build fails when any
sparse_conv_go* instantiation uses scratch memory."""
import re
import subprocess
import sys

PATS = dict(VGPR=r"VGPRs: (\d+)", AGPR=r"AGPRs: (\d+)", scratch=r"ScratchSize \[bytes/lane\]: (\d+)",
            occ=r"Occupancy \[waves/SIMD\]: (\d+)", LDS=r"LDS Size \[bytes/block\]: (\d+)")
check = len(sys.argv) > 1 and sys.argv[1] == "--check"
if check:
    del sys.argv[1]
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ("sparse_conv_go" if check else "")
bad, seen = [], 0
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
    seen += 1
    if check:
        if vals["scratch"] != "0":
            bad.append(f"{short}: scratch {vals['scratch']} bytes/lane, {vals['VGPR']} VGPRs")
        continue
    print(f"{short:90s} " + " ".join(f"{k} {v:>5s}" for k, v in vals.items()))
if check:
    if bad or not seen:
        print("kernel_resources --check FAILED (inline-asm fragment loads must never meet a register spill):", *bad, sep="\n  ")
        sys.exit(1)
    print(f"kernel_resources --check: {seen} {flt}* instantiations, no scratch")
