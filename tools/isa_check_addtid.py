#!/usr/bin/env python3
"""Build-time check on device assembly: the hardware wants one wait state between a scalar write of M0 and an LDS add-TID instruction
(ds_write_addtid_b32 / ds_read_addtid_b32 take their base from M0), and the compiler cannot see into the inline statements that issue
them (stft4096_wg.hip, stft4096_real.hip: the plane transposes).  Without it the FIRST transpose write of a wave used the M0 it was
launched with: the first transform of every workgroup wrong now and then (round 5).  Also fails if anything between two such statements
writes M0 to something the next add-TID would inherit.  usage: isa_check_addtid.py file.s ..."""
import re
import sys

bad = total = 0
# instructions that read M0 without naming it: s_set_gpr_idx_* / v_movrel* (index), buffer_load ... lds and global_load_lds (LDS base),
# ds_gws_* , s_sendmsg*, s_ttracedata, v_interp_* (parameter base), ds_*_addtid itself is the one expected user
IMPLICIT_M0 = re.compile(r"^(s_set_gpr_idx|v_movrel|global_load_lds|ds_gws|s_sendmsg|s_ttracedata|v_interp|ds_ordered_count)")
_addtid_files = {}


def uses_addtid(path):
    if path not in _addtid_files:
        _addtid_files[path] = "addtid" in open(path).read()
    return _addtid_files[path]


for path in sys.argv[1:]:
    since_m0 = None   # instructions issued since the last scalar write of M0 (None: M0 never written in this function)
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if not t or t.startswith((";", "//", ".")) or t.endswith(":"):
            if t.endswith(":") and not t.startswith(".L"):
                since_m0 = None
            continue
        op = t.split()[0]
        if re.match(r"s_\w+\s+m0\b", t):
            since_m0 = 0
            last_m0_write = (path, n, t)
            continue
        if re.search(r"\bm0\b", t) and uses_addtid(path):
            # anything else that reads or writes M0 in a file whose kernels issue add-TID stores: the statements set M0 behind the
            # compiler's back, so compiler-generated users of M0 (s_movrel, LDS-DMA, readlane by M0 ...) must not exist beside them
            bad += 1
            print("%s:%d: %s -- another user of M0 beside the add-TID statements" % (path, n, t))
        if uses_addtid(path) and IMPLICIT_M0.match(op):
            # ... and the users whose assembly text never names m0: indexed register moves, LDS-DMA, GWS, messages
            bad += 1
            print("%s:%d: %s -- reads M0 implicitly, beside add-TID statements that set it behind the compiler's back" % (path, n, t))
        if uses_addtid(path) and op.startswith("buffer_load") and re.search(r"\blds\b", t):
            bad += 1
            print("%s:%d: %s -- LDS-DMA takes its LDS base from M0" % (path, n, t))
        if "addtid" in op:
            total += 1
            if since_m0 is None or since_m0 < 1:
                bad += 1
                print("%s:%d: %s -- %s" % (path, n, t, "M0 never set in this function" if since_m0 is None else "no wait state behind the write of M0"))
        if since_m0 is not None:
            since_m0 += 1
print("isa_check_addtid: %d add-TID instruction(s) checked, %d problem(s)" % (total, bad))
sys.exit(1 if bad else 0)
