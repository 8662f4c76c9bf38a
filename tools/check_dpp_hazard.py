#!/usr/bin/env python3
"""Static check of the built gfx950 code for the one hazard the compiler cannot see: recursion_x.hip issues its broadcast FMAs
(v_fmac_f64_dpp / v_fmac_f32_dpp row_newbcast) from inline asm, and on GFX9 a DPP instruction must not read a VGPR that a
VALU instruction wrote in the two preceding issue slots (nor follow a VALU write of EXEC within five).  The compiler inserts
the wait states for DPP instructions it generates itself, but it does not look inside inline asm, so a register copy it places
right in front of one of ours would go unnoticed.  This script disassembles the device code of an object file and reports
every such pair.

    python tools/check_dpp_hazard.py [build/obj/recursion_x_32.o]      exit status 1 if a hazard is found
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    tmp = tempfile.mkdtemp(prefix="dpphaz_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not dev:
            raise RuntimeError("no gfx950 bundle in " + obj)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, dev[0])],
                              check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def regs(tok):
    """VGPR indices named by an operand token: v7 -> {7}, v[4:5] -> {4, 5}."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def scan(text):
    hazards, ndpp = [], 0
    func = "?"
    window = []          # (mnemonic, written VGPRs, writes_exec, slots) of the preceding instructions, newest last
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func, window = m.group(1), []
            continue
        line = line.split("//")[0].strip()
        if not line or line.endswith(":"):
            continue
        parts = line.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        ops = [o.split()[0] for o in ops if o]
        if mn.endswith("_dpp") or "_dpp" in mn:
            ndpp += 1
            src = regs(ops[1]) if len(ops) > 1 else set()
            dist = 0
            for pm, wr, wexec, slots in reversed(window):
                if dist < 2 and wr & src:
                    hazards.append((func, pm, mn, sorted(wr & src), dist))
                if dist < 5 and wexec:
                    hazards.append((func, pm, mn, "exec", dist))
                dist += slots
                if dist >= 5:
                    break
        # what this instruction writes
        wr, wexec, slots = set(), False, 1
        if mn == "s_nop":
            slots = int(ops[0], 0) + 1 if ops else 1
        elif mn.startswith("v_") and not mn.startswith(("v_readlane", "v_readfirstlane", "v_cmp_")):
            wr = regs(ops[0]) if ops else set()
            wexec = mn.startswith("v_cmpx")
        window.append((mn, wr, wexec, slots))
        if len(window) > 8:
            window.pop(0)
    return hazards, ndpp


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "obj", "recursion_x_32.o")
    hazards, ndpp = scan(disassemble(obj))
    print(f"{obj}: {ndpp} DPP instructions, {len(hazards)} hazards")
    for h in hazards[:20]:
        print("  HAZARD in %s: %s -> %s on %s (distance %d)" % h)
    return 1 if hazards else 0


if __name__ == "__main__":
    sys.exit(main())
