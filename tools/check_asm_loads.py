#!/usr/bin/env python3
"""ISA check for the kernels that issue their vector-memory loads as inline asm with hand-counted waits
(csrc/dense_bf16x3.h, csrc/class_bf16x3.h): no instruction may read or write a load's destination registers while the load
can still be in flight.

hipcc regards an asm load's destination as written when the asm statement ends, so nothing stops it from copying those
registers (loop back-edge copies, live-range splits) or reusing them before the hand-placed `s_waitcnt vmcnt(N)` — and the
hardware writes them whenever the data arrives.  Parity tests do not see this when the inputs are cache resident (round 5:
two rmat-22 runs of the class-layer forward differed; 32 `v_mov_b64` at the loop's back edge).  This walks a kernel's
assembly in program order along every fall-through / branch path a simple CFG gives, keeps the list of issued loads, retires
all but the youngest N at `s_waitcnt vmcnt(N)` (loads complete in order among themselves; stores are ignored, which only makes
the check stricter), and reports any other instruction that touches a register of an unretired load.

    python tools/check_asm_loads.py cuda_gcn_amd/csrc/matmul.hip class_fwd_bf16x3 class_bwd_bf16x3
    python tools/check_asm_loads.py cuda_gcn_amd/csrc/spmm.hip dense_fwd_bf16x3 dense_bwd_bf16x3
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOAD = re.compile(r"^\s*(buffer_load_dword(?:x[234])?|global_load_dword(?:x[234])?)\s+(v\[\d+:\d+\]|v\d+)(?=[,\s])")
REG = re.compile(r"\b(v)\[(\d+):(\d+)\]|\b(v)(\d+)\b|\b(a)\[(\d+):(\d+)\]|\b(a)(\d+)\b")
WAIT = re.compile(r"s_waitcnt\b(.*)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^\s*(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update(("v", i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        elif m.group(4):
            out.add(("v", int(m.group(5))))
    return out


def compile_to_asm(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "cuda_gcn_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def kernels(asm_path, patterns):
    """{mangled name: [instruction lines]} of the kernels whose name contains one of the patterns"""
    res, cur, name = {}, None, None
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m and any(p in m.group(1) for p in patterns):
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                res[name] = cur
                cur = None
                continue
            cur.append(line.rstrip("\n"))
    return res


def check(lines):
    """walk the CFG (labels, conditional / unconditional branches) with the list of in-flight loads as state; returns violations"""
    label_at = {}
    for i, l in enumerate(lines):
        m = LABEL.match(l)
        if m:
            label_at[m.group(1)] = i
    violations, seen = [], set()
    work = [(0, ())]
    while work:
        pc, flight = work.pop()
        flight = list(flight)
        while pc < len(lines):
            key = (pc, tuple(flight))
            l = lines[pc]
            if LABEL.match(l):
                if key in seen:
                    break
                seen.add(key)
            code = l.split(";")[0]
            m = LOAD.match(code)
            w = WAIT.search(code)
            b = BRANCH.match(code)
            if m:
                dst = regs_of(m.group(2))
                rest = code[m.end():]
                for used in regs_of(rest) | dst:
                    for (ln, d) in flight:
                        if used in d:
                            violations.append((pc, l.strip(), ln))
                flight.append((pc, frozenset(dst)))
            elif w:
                vm = re.search(r"vmcnt\((\d+)\)", w.group(1))
                if vm:
                    n = int(vm.group(1))
                    flight = flight[len(flight) - n:] if n < len(flight) else flight
                    if n == 0:
                        flight = []
            elif code.strip() and not code.strip().startswith((".", "s_", "ds_", ";")):
                used = regs_of(code)
                for (ln, d) in flight:
                    if used & d:
                        violations.append((pc, l.strip(), ln))
            elif code.strip().startswith("ds_"):
                used = regs_of(code)
                for (ln, d) in flight:
                    if used & d:
                        violations.append((pc, l.strip(), ln))
            if b:
                tgt = label_at.get(b.group(2))
                if tgt is not None:
                    work.append((tgt, tuple(flight)))
                if b.group(1) == "s_branch":
                    break
            if "s_endpgm" in code:
                break
            pc += 1
    # unique
    uniq = {}
    for v in violations:
        uniq.setdefault((v[0], v[2]), v)
    return sorted(uniq.values())


def main():
    src = sys.argv[1]
    pats = sys.argv[2:] or ["bf16x3"]
    asm = compile_to_asm(os.path.join(ROOT, src) if not os.path.isabs(src) else src)
    ks = kernels(asm, pats)
    bad = 0
    for name, lines in sorted(ks.items()):
        v = check(lines)
        n_loads = sum(1 for l in lines if LOAD.match(l.split(";")[0]))
        print(f"{name}: {len(lines)} lines, {n_loads} asm/vector loads, {len(v)} accesses to registers of loads in flight")
        for pc, text, issued in v[:12]:
            print(f"    line {pc}: {text}    (load issued at line {issued}: {lines[issued].strip()})")
        bad += len(v)
    os.unlink(asm)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
