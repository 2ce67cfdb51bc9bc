#!/usr/bin/env python3
"""Build-time check of the hand-written prefetch in stft4096_wg.hip.

The mono kernels request the two sliding-window rows with inline `global_load_dword` and wait for them an iteration
later with a hand-counted inline `s_waitcnt vmcnt(N)`.  The compiler believes the two destination VGPRs are written
at the asm statement, so any instruction it places between the request and the wait that READS or MOVES them (a phi
copy at the loop header, a spill) would see stale data -- gfx9 has no interlock for that.  This script reads the
device assembly (hipcc -S --cuda-device-only) and fails unless, in every kernel that contains the request:

  * nothing outside the inline-asm blocks touches the two destination registers between the request and the end of the
    loop body (its backward branch), nor between the loop header and the first hand-written vmcnt wait;
  * no scratch instruction mentions them anywhere.

usage: tools/isa_check_prefetch.py file.s      (exit 1 and a report on violation)"""
import re
import sys


def regs_of(line):
    """VGPR numbers a line mentions (v12, v[10:13])"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    return out


def check_function(name, lines):
    # annotate inline-asm blocks
    in_asm, asm_flag = False, []
    for ln in lines:
        if "#ASMSTART" in ln:
            in_asm = True
        asm_flag.append(in_asm)
        if "#ASMEND" in ln:
            in_asm = False
    code = [re.sub(r";.*", "", ln).strip() for ln in lines]
    labels = {m.group(1): i for i, ln in enumerate(lines) for m in [re.match(r"^(\.LBB\w+):", ln)] if m}
    requests = [i for i in range(len(code) - 1) if asm_flag[i] and code[i].startswith("global_load_dword ")
                and code[i + 1].startswith("global_load_dword ")]
    problems = []
    for i in requests:
        dst = {int(re.match(r"global_load_dword v(\d+)", code[j]).group(1)) for j in (i, i + 1)}
        # the loop's backward branch after the request
        back = None
        for j in range(i + 2, len(code)):
            m = re.match(r"s_cbranch_\w+ (\.LBB\w+)|s_branch (\.LBB\w+)", code[j])
            if m:
                lab = m.group(1) or m.group(2)
                if lab in labels and labels[lab] < i:
                    back, header = j, labels[lab]
                    break
        if back is None:
            problems.append(f"{name}: no backward branch after the request at line {i}")
            continue
        wait = next((j for j in range(header, i) if asm_flag[j] and "vmcnt" in code[j]), None)
        if wait is None:
            problems.append(f"{name}: no hand-written vmcnt wait between the loop header and the request")
            continue
        forbidden = list(range(i + 2, back + 1)) + list(range(header, wait))
        for j in forbidden:
            if asm_flag[j] or not code[j] or code[j].endswith(":"):
                continue
            hit = regs_of(code[j]) & dst
            if hit:
                problems.append(f"{name}: line {j}: `{code[j]}` touches v{sorted(hit)} while its load is still pending")
        for j, c in enumerate(code):
            if c.startswith("scratch_") and regs_of(c) & dst:
                problems.append(f"{name}: line {j}: `{c}` spills a prefetch destination")
    return len(requests), problems


def main():
    text = open(sys.argv[1]).read().split("\n")
    starts = [(i, m.group(1)) for i, ln in enumerate(text) for m in [re.match(r"^(_Z\w+):", ln)] if m]
    total, problems = 0, []
    for k, (i, name) in enumerate(starts):
        end = next((j for j in range(i, len(text)) if text[j].startswith(".Lfunc_end")), len(text))
        n, p = check_function(name, text[i:end])
        total += n
        problems += p
    if total == 0:
        problems.append("no hand-written prefetch found at all (was the kernel renamed or the asm removed?)")
    for p in problems:
        print("isa_check_prefetch:", p)
    print(f"isa_check_prefetch: {total} prefetch request(s) checked, {len(problems)} problem(s)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
