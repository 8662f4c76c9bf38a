#!/usr/bin/env python3
"""Register / LDS / scratch use of the gfx950 kernels in an object file, one line per kernel (from the code object's metadata notes).
    python tools/kres.py build/obj/recursion_x_32.o [substring ...]      only kernels whose demangled name holds every substring"""
import os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"


def device_object(obj, tmp):
    local = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dev = [f for f in os.listdir(tmp) if "gfx950" in f]
    return os.path.join(tmp, dev[0])


def main():
    obj, subs = sys.argv[1], sys.argv[2:]
    tmp = tempfile.mkdtemp(prefix="kres_")
    try:
        dev = device_object(obj, tmp)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", dev], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kernels, cur = [], {}
    for ln in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("symbol"):       # first key of a kernel's record (keys are sorted): start a new one
            kernels.append(cur); cur = {}
        if k in ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "name", "symbol", "max_flat_workgroup_size"):
            cur.setdefault(k, v)
    if cur.get("symbol"):
        kernels.append(cur)
    names = subprocess.run(["c++filt"], input="\n".join(k.get("name", "?") for k in kernels), capture_output=True, text=True).stdout.splitlines()
    for k, n in zip(kernels, names):
        n = re.sub(r"moihgp::\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*$", "", n)
        if all(s in n for s in subs):
            print(f"v{k.get('vgpr_count','?'):>4} a{k.get('agpr_count','?'):>3} s{k.get('sgpr_count','?'):>4} lds{k.get('group_segment_fixed_size','?'):>7} scr{k.get('private_segment_fixed_size','?'):>5} "
                  f"spill v{k.get('vgpr_spill_count','0')}/s{k.get('sgpr_spill_count','0')} wg{k.get('max_flat_workgroup_size','?'):>5}  {n}")


if __name__ == "__main__":
    main()
