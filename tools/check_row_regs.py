#!/usr/bin/env python3
"""Checks the compiled shade phase of the render kernels that read a row ahead (k_render_persistent<..., WIDE = true, ...>).

The loads of a row are issued by inline assembly the compiler does not track (row_request in rtx_kernels.hip); their
destination registers are being written until the statement that waits for them (row_wait) has run.  Nothing the compiler
emits may read those registers in between, and that is a property of its register allocation, not of the source: this
script disassembles nothing, it reads the compiler's assembly output (hipcc -S) and fails if any instruction other than the
loads themselves and the copies that directly follow a wait mentions one of those registers.

    tools/check_row_regs.py            compiles raytracing_amd/csrc/rtx_kernels.hip to a temporary .s and checks it
    tools/check_row_regs.py file.s     checks an existing assembly file
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "raytracing_amd", "csrc", "rtx_kernels.hip")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-munsafe-fp-atomics"]
ROW_LOADS = 15          # loads per row_request


def compile_to_asm(path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, *FLAGS, "-S", "--cuda-device-only", "-o", path, "-x", "hip", SRC], check=True, stderr=subprocess.DEVNULL)


def registers(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def kernels(lines):
    """(name, first line, last line) of every WIDE instance of the render kernel"""
    found = []
    name, start = None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN3rtx19k_render_persistentILb[01]ELb[01]ELb[01]ELb1ELb[01]EE\w*):", l)
        if m:
            name, start = m.group(1), i
        if name and ".end_amdhsa_kernel" in l:
            found.append((name, start, i))
            name = None
    return found


def check(lines, name, a, b):
    body = lines[a:b]
    code = [(i, l) for i, l in enumerate(body) if l.strip() and not l.strip().startswith(";")]
    # The rows live between the first row's loads and the kernel's last inline-assembly block that loads, waits, adds or stores (row_request, row_wait, row_tail);
    # what follows in the text (the tail phase, the sums of the work counters) may use any register and has loads of its own.
    last_asm, inside, row_block = len(body), False, False
    in_asm = set()   # lines inside inline-assembly blocks
    for i, l in enumerate(body):   # inline-assembly blocks of row_request / row_wait / row_tail: they load, wait, add or store
        if ";;#ASMSTART" in l:
            inside, row_block = True, False
        elif ";;#ASMEND" in l:
            if inside and row_block:
                last_asm = i
            inside = False
        elif inside:
            in_asm.add(i)
            if re.search(r"global_load_dword|s_waitcnt vmcnt|global_atomic_add_f64|global_store_dword", l):
                row_block = True
    # the rows' loads are the non-temporal loads issued by inline assembly (the closest-hit walk has non-temporal loads of its
    # own, emitted by the compiler, wherever the compiler places that code)
    nt = [(i, l) for i, l in code if i in in_asm and re.search(r"global_load_dword(x2)?\s.*off nt", l)]
    if len(nt) != 2 * ROW_LOADS:
        return ["%s: %d non-temporal loads in inline assembly, expected %d (has the kernel changed?)" % (name, len(nt), 2 * ROW_LOADS)]
    first, again = nt[:ROW_LOADS], nt[ROW_LOADS:]
    dest = set()
    for _, l in again:
        dest |= registers(l.split(",")[0])
    dest_first = set()
    for _, l in first:
        dest_first |= registers(l.split(",")[0])
    errors = []
    if dest != dest_first:
        errors.append("%s: the first row and the later rows are loaded into different registers" % name)
    load_lines = {i for i, _ in nt}
    last_wait = None
    for i, l in code:
        if i < first[0][0] or i > last_asm:
            continue
        if re.search(r"s_waitcnt vmcnt\((0|16)\)", l):
            last_wait = i
            copies = 0
            continue
        if i in load_lines:
            last_wait = None
            continue
        if registers(l) & dest:
            ok = last_wait is not None and re.match(r"\s*v_mov_b(32|64) ", l)
            if not ok:
                errors.append("%s: line %d reads a register of the row in flight: %s" % (name, a + i + 1, l.strip()))
    return errors


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
        lines = open(path).read().split("\n")
    else:
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "rtx_kernels.s")
            compile_to_asm(path)
            lines = open(path).read().split("\n")
    ks = kernels(lines)
    if not ks:
        print("no WIDE instance of k_render_persistent found")
        return 1
    errors = []
    for name, a, b in ks:
        errors += check(lines, name, a, b)
    for e in errors:
        print(e)
    print("%d kernels checked, %d problems" % (len(ks), len(errors)))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
