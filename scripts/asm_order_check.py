"""Build-time guard of the gather-once convolution kernels' hand-counted waits:
    python scripts/asm_order_check.py cn-rma_amd/csrc/sparse.o        (run by the Makefile after every compile of sparse.hip)

The weight fragments of sparse_conv_go2 / gof are fetched by inline-asm `global_load_dwordx4 v, v, s[base]` groups (behind an
`s_nop 4`) and released by `s_waitcnt vmcnt(N)` with N counted by hand: N = the fragment loads that may still be in flight.  That is
only right while the steady-state offset loop issues NO other vector-memory instruction -- one the compiler hoists or sinks into the
loop (a gather, a split-slab store; stores count in vmcnt too) would be newer than the fragment being waited for, and the counted
wait would release registers whose load has not landed: silently corrupted operands.  This check disassembles the gfx950 code
object and, in every such kernel, takes the INNERMOST loops (a backward branch whose body holds no other backward branch) that
contain a fragment-load group: their bodies must hold no global_ / buffer_ / scratch_ / flat_ instruction outside the groups."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
# (the experimental third form, sparse_conv_go3_kernel, issues one counted inline-asm row load per step on purpose: not checked)
KERNELS = ("sparse_conv_go2_kernel", "sparse_conv_gof_kernel")
obj = os.path.abspath(sys.argv[1])
with tempfile.TemporaryDirectory() as tmp:
    # (clang-offload-bundler does not know this object layout; llvm-objdump --offloading writes <input>.<n>.<triple> beside it)
    import glob
    import shutil
    local = os.path.join(tmp, "in.o")
    shutil.copy(obj, local)
    r = subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], capture_output=True, text=True, cwd=tmp)
    cos = [f for f in glob.glob(os.path.join(tmp, "in.o.*gfx950*")) if os.path.getsize(f) > 0]
    if r.returncode != 0 or len(cos) != 1:
        print("asm_order_check: could not extract the gfx950 code object:", r.stderr.strip(), cos)
        sys.exit(1)
    co = cos[0]
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout

funcs, cur = {}, None
for line in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
    if m:
        cur = funcs.setdefault(m.group(1), [])
        continue
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
    if m and cur is not None:
        cur.append((int(m.group(3), 16), m.group(1), m.group(2), line))

VMEM = ("global_", "buffer_", "scratch_", "flat_")
bad, checked, groups_total = [], 0, 0
for name, ins in funcs.items():
    if not any(k in name for k in KERNELS):
        continue
    base = ins[0][0]
    # fragment-load groups: s_nop 4, then >= 2 global_load_dwordx4 with an SGPR base
    in_group = set()
    i = 0
    while i < len(ins):
        if ins[i][1] == "s_nop" and ins[i][2].strip() == "4":
            j = i + 1
            while j < len(ins) and ins[j][1] == "global_load_dwordx4" and re.search(r",\s*s\[\d+:\d+\]", ins[j][2]):
                j += 1
            if j - (i + 1) >= 2:
                in_group.update(range(i + 1, j))
                groups_total += 1
            i = j
        else:
            i += 1
    if not in_group:
        bad.append(f"{name}: no inline-asm fragment-load group found (the pattern this guard knows has changed)")
        continue
    # backward branches: (index of the branch, index of its target)
    addr_to_idx = {a: k for k, (a, _, _, _) in enumerate(ins)}
    loops = []
    for k, (a, op, args, line) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", line)
            t = base + int(m.group(1), 16) if m else (base if re.search(r"<[^+>]+>\s*$", line) else None)
            if t is not None and t <= a and t in addr_to_idx:
                loops.append((addr_to_idx[t], k))
    inner = [(s, e) for (s, e) in loops if not any((s2, e2) != (s, e) and s <= s2 and e2 <= e for (s2, e2) in loops)]
    checked += 1
    for s, e in inner:
        body = range(s, e + 1)
        if not any(k in in_group for k in body):
            continue
        for k in body:
            if k not in in_group and ins[k][1].startswith(VMEM):
                bad.append(f"{name}: `{ins[k][1]} {ins[k][2]}` at {ins[k][0]:#x} inside the steady-state offset loop "
                           f"[{ins[s][0]:#x}, {ins[e][0]:#x}] next to the counted fragment loads")
if bad or not checked:
    print("asm_order_check FAILED (hand-counted s_waitcnt vmcnt(N) of the gather-once kernels):", *bad, sep="\n  ")
    sys.exit(1)
print(f"asm_order_check: {checked} gather-once instantiations, {groups_total} fragment-load groups, no foreign vector-memory "
      f"instruction in a steady-state offset loop")
