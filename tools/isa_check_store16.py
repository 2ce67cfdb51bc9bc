#!/usr/bin/env python3
"""Build-time check of the 16-byte row stores of the 16384-point kernels (stft16384_d.hip, stft16384_q.hip).

A `buffer_store_dwordx4` whose scalar offset sits in an SGPR, followed at once by a vector write of one of its data registers,
stored the NEW value on this device (rows with LDS addresses in them, now and then: DESIGN section 4 K16).  LLVM's hazard
table knows that hazard only for an immediate offset, so the kernels put `s_nop` wait states behind every such store -- as
a separate `asm volatile`, which the post-RA scheduler is free to move a register-reusing VALU instruction in front of (only
the WAR dependence on the store orders that instruction).  Writing the store itself in inline asm would hide it from the
compiler's vmcnt bookkeeping (its waits for the prefetched samples are counted in stores issued since), so the store stays
a builtin and this script reads the device assembly (hipcc -S --cuda-device-only) and fails the build unless, behind EVERY
buffer_store_dwordx4, kWaitStates wait states pass (an `s_nop N` counts N + 1, any other instruction 1) before any
instruction that WRITES one of its data registers: a VALU destination, a v_swap operand, a v_readlane-style scalar
destination does not count, LDS / vector-memory loads into them do (their write comes later still, but nothing is gained by
allowing it).

usage: tools/isa_check_store16.py file.s [file.s ...]      (exit 1 and a report on violation)"""
import re
import sys

kWaitStates = 2


def reg_set(operand):
    """VGPR numbers of ONE operand (v12, v[10:13]); empty for anything else"""
    operand = operand.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", operand)
    return {int(m.group(1))} if m else set()


def written_vgprs(code):
    """VGPRs an instruction writes, conservatively (first operand of v_*, ds_read*, *_load_*; both operands of v_swap)"""
    m = re.match(r"(\S+)\s+(.*)", code)
    if not m:
        return set()
    op, rest = m.group(1), m.group(2)
    ops = [o for o in re.split(r",\s*", rest)]
    if op.startswith("v_swap"):
        return reg_set(ops[0]) | (reg_set(ops[1]) if len(ops) > 1 else set())
    if op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return set()
    if op.startswith("v_") or op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_permute") \
            or re.match(r"(buffer|global|flat|scratch)_load", op) or op.startswith("ds_swizzle"):
        return reg_set(ops[0])
    return set()


def check_file(path):
    lines = open(path).read().split("\n")
    code = [re.sub(r";.*", "", ln).strip() for ln in lines]
    stores, problems = 0, []
    for i, c in enumerate(code):
        if not c.startswith("buffer_store_dwordx4 "):
            continue
        stores += 1
        data = reg_set(re.split(r",\s*", c.split(None, 1)[1])[0])
        waited, j = 0, i + 1
        while waited < kWaitStates and j < len(code):
            cj = code[j]
            j += 1
            if not cj or cj.endswith(":") or cj.startswith(".") or cj.startswith(";;#"):
                continue
            m = re.match(r"s_nop (\d+)", cj)
            if m:
                waited += int(m.group(1)) + 1
                continue
            hit = written_vgprs(cj) & data
            if hit:
                problems.append(f"{path}: line {j}: `{cj}` writes v{sorted(hit)} {waited} wait state(s) behind `{c}` (line {i + 1})")
                break
            if re.match(r"s_cbranch|s_branch|s_endpgm|s_setpc", cj):
                break      # control leaves the straight line: the branch itself and the fetch are more than the wait states asked for
            waited += 1
    return stores, problems


def main():
    total, problems = 0, []
    for path in sys.argv[1:]:
        n, p = check_file(path)
        if n == 0:
            p.append(f"{path}: no buffer_store_dwordx4 found at all (was the kernel renamed or the store changed?)")
        total += n
        problems += p
    for p in problems:
        print("isa_check_store16:", p)
    print(f"isa_check_store16: {total} 16-byte store(s) checked, {len(problems)} problem(s)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
