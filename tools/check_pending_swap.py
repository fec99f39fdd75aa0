#!/usr/bin/env python3
"""The frame kernel issues its STARTED swap from inline assembly and waits for the answer much later (mark_tile_started_issue /
mark_tile_started_answer, instance_kernel.hpp): the compiler does not know that the two destination registers are pending in
between. This checks the generated ISA: from every such swap to the hand-written `s_waitcnt vmcnt(0)` that follows it, no
instruction may read, write, spill or copy those registers (the walk follows the layout order and unconditional forward
branches; a conditional branch is followed on its fall-through side, the blocks behind it are walked when the layout reaches them).

  python tools/check_pending_swap.py [asm dir]      (default: builds it with `make -C renderer_amd/csrc asm ASM_DIR=<tmp>`)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def registers(line):
    out = set()
    for m in REG.finditer(line):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_file(path):
    """-> (swaps found, list of violations)"""
    swaps, bad = 0, []
    func, pending, pending_line = None, None, 0
    in_asm = False
    skip_to = None  # after an unconditional branch: the label where the walk goes on (the blocks in between belong to other paths)
    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            if skip_to is not None:
                if raw.startswith(skip_to + ":"):
                    skip_to = None
                elif re.match(r"^\w+:", raw) and not raw.startswith(".L"):
                    bad.append(f"{path}:{ln}: {func}: branch target {skip_to} not found before the next function")
                    skip_to, pending = None, None
                else:
                    continue
            line = raw.split(";", 1)[0].rstrip() if not raw.lstrip().startswith(";;#") else raw.strip()
            if raw.lstrip().startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if raw.lstrip().startswith(";;#ASMEND"):
                in_asm = False
                continue
            m = re.match(r"^(\w+):", raw)
            if m and not raw.startswith(".L"):
                func = m.group(1)
                if pending:
                    bad.append(f"{path}:{pending_line}: {func}: swap without a hand-written wait before the next function")
                pending = None
            if "global_atomic_swap_x2" in line and in_asm:
                swaps += 1
                dst = re.search(r"global_atomic_swap_x2\s+v\[(\d+):(\d+)\]", line)
                pending, pending_line = set(range(int(dst.group(1)), int(dst.group(2)) + 1)), ln
                continue
            if pending is None:
                continue
            if in_asm and re.match(r"\s*s_waitcnt\s+vmcnt\(0\)", line):
                pending = None
                continue
            jump = re.match(r"\s*s_branch\s+(\.L\w+)", line)
            if jump:
                skip_to = jump.group(1)
                continue
            if "s_endpgm" in line:
                pending = None  # (a path that leaves without asking for the answer)
                continue
            touched = registers(line) & pending
            if touched:
                bad.append(f"{path}:{ln}: {func}: `{line.strip()}` touches v{sorted(touched)} while the swap of line {pending_line} is pending")
    return swaps, bad


def main():
    if len(sys.argv) > 1:
        d = sys.argv[1]
    else:
        d = tempfile.mkdtemp(prefix="mip_asm_")
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "renderer_amd", "csrc"), "asm", f"ASM_DIR={d}"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    total, bad = 0, []
    for path in sorted(glob.glob(os.path.join(d, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
        n, b = check_file(path)
        total += n
        bad += b
    print(f"{total} pending swaps checked, {len(bad)} violations")
    for b in bad:
        print("  " + b)
    if total == 0:
        print("  no swap found: the check does not see the kernels it is meant for")
    return 1 if bad or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
