#!/usr/bin/env python3
"""Scratch (private-memory) traffic inside MFMA loops, per kernel of a built library.

    tools/check_scratch.py [library.so] [--all] [--json]        (default library: motionrag_amd/libmrag_hip.so)

Every scratch access is a vector-memory operation: the `s_waitcnt vmcnt(0)` hipcc puts behind it also drains the LDS-DMA ring, so a private
array that the compiler leaves in scratch inside a K loop stalls the loop once per K-tile (found in topk.hip in round 5 -- `Cand tau[TN]` --
and by the round-5 review in the 256x256 implicit-GEMM convolutions: `cv_src[APW]` / `cv_step[APW]`).

The pass: take the gfx950 code objects out of the library's .hip_fatbin, disassemble them (llvm-objdump), and per kernel
  * find its LOOPS: a backward branch at address a to target t < a is the loop [t, a];
  * for every INNERMOST loop that contains MFMAs (the K loops / key sweeps), count the scratch_* instructions inside it (`in_loop`), and
  * count the scratch_* instructions between the kernel's first and last MFMA (`in_span`: the review's measure) and in the whole kernel (`total`).
A kernel FAILS when a loop with MFMAs holds a scratch instruction, unless the allow-list below names the kernel AND the loop is not one of
its first `clean_loops` MFMA loops in text order (attn16 / attn8: the fast sweep comes first in the kernel text and must be clean; the
checked re-run sweep behind it -- taken by a workgroup whose row sums leave the fp32 range, i.e. never on real activations -- may spill).
Exit code 1 on failure.  tests/test_scratch_cpu.py runs this on the shipped library.
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
# kernel-name regex -> number of leading MFMA loops (text order) that must be scratch-free; loops behind them are the checked slow path
#   attn16_kernel / attn8_kernel: the fast (optimistic) sweep is the first MFMA loop of the kernel text; the loops behind it are the checked re-run sweep
#   attn_fwd_kernel<8, ...> (32x32x16 flash kernel of the masked / short launches, 128 VGPRs at two workgroups per CU): its three steady-state key loops
#     come first; the loops behind them handle the ragged last stage once per workgroup and reload 1-3 spilled words there
ALLOW = {r"attn16_kernel<": 1, r"attn8_kernel<": 1, r"attn_fwd_kernel<8, ": 3}


def code_objects(lib, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, os.devnull], check=True, capture_output=True)
    blob = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            break
        cnt = struct.unpack_from("<Q", blob, i + 24)[0]
        p = i + 32
        for _ in range(cnt):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                path = os.path.join(tmp, f"co{len(out)}.elf")
                open(path, "wb").write(blob[i + off:i + off + size])
                out.append(path)
        pos = i + 24
    return out


def kernels_of(elf):
    """yield (mangled name, [(address, opcode, branch target or None)])"""
    text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", elf], check=True, capture_output=True, text=True).stdout
    name, base, ins = None, 0, []
    head = re.compile(r"^([0-9a-f]+) <(.+)>:$")
    line = re.compile(r"^\s+(\S+)\s.*//\s*([0-9A-Fa-f]+):(?:.*<[^>+]+\+0x([0-9a-f]+)>)?")
    for l in text.split("\n"):
        m = head.match(l)
        if m:
            if name:
                yield name, ins
            name, base, ins = m.group(2), int(m.group(1), 16), []
            continue
        m = line.match(l)
        if m and name:
            op, addr = m.group(1), int(m.group(2), 16)
            tgt = base + int(m.group(3), 16) if (m.group(3) and op.startswith(("s_cbranch", "s_branch"))) else None
            ins.append((addr, op, tgt))
    if name:
        yield name, ins


def analyse(name, ins):
    mf = [a for a, op, _ in ins if op.startswith(("v_mfma", "v_smfmac"))]
    sc = [a for a, op, _ in ins if op.startswith("scratch_")]
    if not mf or not sc:
        return None
    loops = sorted({(t, a) for a, op, t in ins if t is not None and t <= a})
    mloops = [(t, a) for t, a in loops if any(t <= x <= a for x in mf)]
    # INNERMOST loops with MFMAs only: an outer loop (the persistent GEMM's tile loop, a retry loop around a sweep) also spans epilogues and
    # set-up code, where a spill costs one latency per tile, not one per K-tile
    mloops = [(t, a) for t, a in mloops if not any((t2, a2) != (t, a) and t <= t2 and a2 <= a for t2, a2 in mloops)]
    rows = []
    for t, a in mloops:
        rows.append({"start": t - ins[0][0], "mfma": sum(t <= x <= a for x in mf), "scratch": sum(t <= x <= a for x in sc)})
    return {"kernel": name, "total": len(sc), "in_span": sum(mf[0] < x < mf[-1] for x in sc), "mfma": len(mf), "loops": rows}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main(argv):
    args = [a for a in argv if not a.startswith("--")]
    lib = args[0] if args else os.path.join(ROOT, "motionrag_amd", "libmrag_hip.so")
    res = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            for name, ins in kernels_of(co):
                r = analyse(name, ins)
                if r:
                    res.append(r)
    dm = demangle([r["kernel"] for r in res])
    bad = 0
    for r in res:
        r["kernel"] = dm[r["kernel"]].replace("(anonymous namespace)::", "")
        clean = next((n for pat, n in ALLOW.items() if re.search(pat, r["kernel"])), None)
        r["in_loop"] = sum(l["scratch"] for l in r["loops"])
        viol = [l for i, l in enumerate(r["loops"]) if l["scratch"] and (clean is None or i < clean)]
        r["status"] = "FAIL" if viol else ("allow-listed slow path" if r["in_loop"] else "ok")
        bad |= bool(viol)
    res.sort(key=lambda r: (-(r["status"] == "FAIL"), -r["in_loop"], -r["in_span"], -r["total"]))
    if "--json" in argv:
        print(json.dumps(res, indent=1))
    else:
        print("in_loop in_span total mfma  status  kernel   [MFMA loops: start+scratch/mfma]")
        for r in res:
            if r["in_loop"] or r["in_span"] or "--all" in argv:
                loops = " ".join(f"+{l['start']:#x}:{l['scratch']}/{l['mfma']}" for l in r["loops"] if l["scratch"] or "--all" in argv)
                print(f"{r['in_loop']:7d} {r['in_span']:7d} {r['total']:5d} {r['mfma']:4d}  {r['status']}  {r['kernel']}  [{loops}]")
        print(f"{len(res)} kernels with MFMAs and scratch instructions; {'FAIL: scratch inside an MFMA loop' if bad else 'no scratch inside a hot MFMA loop'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
