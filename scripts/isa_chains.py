#!/usr/bin/env python3
"""Scan the gfx950 code object of an object file / the built library for dependent memory round trips: a vector-memory load whose
result is waited for (s_waitcnt vmcnt(0)) within a few instructions, more than once per kernel -- the pattern behind this round's
GEMM-tail, LayerNorm and fp8-producer fixes.  Also counts scratch and v_readlane / v_writelane traffic.

  python scripts/isa_chains.py diffusionkit_amd/libdk_hip.so [name-filter]
"""
import re, subprocess, sys, tempfile, os

B = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    t = tempfile.mkdtemp()
    subprocess.check_call([f"{B}/llvm-objcopy", f"--dump-section=.hip_fatbin={t}/fb.bin", obj, f"{t}/copy.o"])
    subprocess.check_call([f"{B}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={t}/fb.bin",
                           f"--output={t}/k.co", "--unbundle"])
    return subprocess.check_output([f"{B}/llvm-objdump", "-d", f"{t}/k.co"], text=True)


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else "diffusionkit_amd/libdk_hip.so"
    flt = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    name, rows, cur = None, [], None
    for line in disassemble(obj).splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            if cur: rows.append(cur)
            name = m.group(1)
            cur = {"name": name, "n": 0, "loads": 0, "chains": 0, "scratch": 0, "lane": 0, "since": None}
            continue
        if cur is None or "\t" not in line: continue
        ins = line.split("\t")[1].split("//")[0].strip() if line.count("\t") else ""
        if not ins: continue
        cur["n"] += 1
        op = ins.split()[0]
        if op.startswith(("global_load", "buffer_load", "flat_load")) and " lds" not in ins:
            cur["loads"] += 1; cur["since"] = 0
        elif op.startswith("scratch_"): cur["scratch"] += 1
        elif op in ("v_readlane_b32", "v_writelane_b32"): cur["lane"] += 1
        elif cur["since"] is not None:
            cur["since"] += 1
            if op == "s_waitcnt" and re.search(r"vmcnt\(0\)", ins) and cur["since"] <= 6: cur["chains"] += 1
            if cur["since"] > 6: cur["since"] = None
    if cur: rows.append(cur)
    print(f"{'kernel':90s} {'insts':>6s} {'loads':>5s} {'load->wait(0)':>13s} {'scratch':>7s} {'lane r/w':>8s}")
    for r in sorted(rows, key=lambda r: -r["chains"]):
        if flt.search(r["name"]) and (r["chains"] > 1 or r["scratch"] or r["lane"] > 8):
            print(f"{r['name'][:90]:90s} {r['n']:6d} {r['loads']:5d} {r['chains']:13d} {r['scratch']:7d} {r['lane']:8d}")


if __name__ == "__main__":
    main()
