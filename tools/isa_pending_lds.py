#!/usr/bin/env python3
"""Check the hand-pipelined LDS reads of a unit's gfx950 ISA: a ds_read issued from inline asm (no automatic s_waitcnt) leaves a
PENDING write on its destination registers; nothing may read or write those registers until enough LDS operations have retired
(`s_waitcnt lgkmcnt(N)`: LDS operations retire in order, so at most the N youngest are still in flight).  hipcc cannot see the
pending write: if it parks another value in such a register, copies it, or lets the kernel's tail reuse it, the late LDS return
silently corrupts data (round 5: the first pipelined form of gemv_i8q4_p16_kernel failed the small-shape parity tests this way).

Rule 2: no asm-issued read may be pending at a label or a branch.  hipcc inserts copies at control-flow merges (it did, in front
of the wait), so the hand pipelines live inside straight-line regions; with that rule the linear scan over a function's layout is
exact.

usage: isa_pending_lds.py <unit, e.g. gemv_ref> | --file <asm file>     exit status 1 if a violation is found"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def scan(lines):
    """-> list of (function, line number, instruction, registers) violations"""
    fn, in_asm, queue, bad = None, False, [], []   # queue: in-flight LGKM operations, oldest first; entry = set of pending asm destinations (or empty)
    for ln, raw in enumerate(lines, 1):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            fn, queue, in_asm = m.group(1), [], False
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if re.match(r"^\.LBB\w+:", raw) and queue and set().union(*queue):
            bad.append((fn, ln, line, sorted(set().union(*queue))))        # pending across a label
            queue = []
        if not line or line.startswith(";") or line.startswith("."):
            continue
        ins = line.split(";")[0].strip()
        op = ins.split()[0]
        if re.match(r"s_c?branch|s_setpc", op) and queue and set().union(*queue):
            bad.append((fn, ln, ins, sorted(set().union(*queue))))         # pending across a branch
            queue = []
        if op == "s_endpgm":
            queue = []
            continue
        pending = set().union(*queue) if queue else set()
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ins)
            if m:
                n = int(m.group(1))
                while len(queue) > n:
                    queue.pop(0)
            continue
        touched = regs_of(ins)
        if op.startswith("ds_read") and in_asm:
            dst = regs_of(ins.split(",")[0])
            hit = (touched - dst) & pending                     # its address register must not be pending; re-targeting a pending destination
            if hit:                                             # is legal for the hardware (in-order writes) and used by the in-place refills
                bad.append((fn, ln, ins, sorted(hit)))
            queue.append(set(dst))
            continue
        hit = touched & pending
        if hit:
            bad.append((fn, ln, ins, sorted(hit)))
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memrealtime") or op.startswith("s_memtime"):
            queue.append(set())                                 # compiler-managed LGKM operation: occupies a slot, has its own wait
    return bad


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--file":
        name, asm = sys.argv[2], sys.argv[2]
    else:
        unit = sys.argv[1] if len(sys.argv) > 1 else "gemv_ref"
        name = unit
        src = os.path.join(ROOT, "jlama_amd", "csrc", unit + ".hip")
        asm = f"/tmp/_isa_{unit}.s"
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value",
                        "-S", "--offload-device-only", src, "-o", asm], check=True, capture_output=True)
    lines = open(asm).read().splitlines()
    n_asm_reads = sum(1 for i, l in enumerate(lines) if l.strip().startswith("ds_read") and i > 0 and lines[i - 1].strip().startswith(";;#ASMSTART"))
    bad = scan(lines)
    if not bad:
        print(f"{name}: {n_asm_reads} asm LDS reads, no register touched while its LDS write is pending")
        return 0
    for fn, ln, ins, regs in bad[:40]:
        dem = subprocess.run(["c++filt", fn or ""], capture_output=True, text=True).stdout.strip()
        print(f"{dem[:100]}: line {ln}: `{ins}` touches v{regs} with an LDS write pending")
    print(f"{name}: {len(bad)} violations")
    return 1


if __name__ == "__main__":
    sys.exit(main())
