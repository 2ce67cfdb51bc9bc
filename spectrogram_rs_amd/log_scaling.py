"""Host-side mirror of src/log_scaling.rs (LogCoordf64, a reversible log axis).

Only what the pixel path uses is mirrored: construction from a range with zero_point/base
(log_scaling.rs:126-191), `map` (:47-51) and `unmap` (:114-119).  `key_points` (:53-107) is
axis decoration and out of scope.  The engine evaluates the same expression in C++ when a
context is created (sgx_bin_edges); this class is the host mirror the tests compare against.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple


class LogCoordf64:
    def __init__(self, start: float, end: float, base: float = 10.0, zero_point: float = 0.0):
        # From<ReversibleLogRangeExt> (log_scaling.rs:160-191)
        self.logic = (float(start), float(end))
        s, e = start - zero_point, end - zero_point
        self.negative = s < 0.0 or e < 0.0
        if self.negative:
            s, e = -s, -e
        if s < e:
            if s == 0.0:
                s = max(s, e * 1e-5)
        elif e == 0.0:
            e = max(e, s * 1e-5)
        self.normalized = (s, e)
        self.linear = (math.log(s), math.log(e))
        self.base = base
        self.zero_point = zero_point

    @classmethod
    def reversible_log_scale(cls, start: float, end: float) -> "LogCoordf64":
        """(start..end).reversible_log_scale()  (log_scaling.rs:121-133)"""
        return cls(start, end)

    def with_base(self, base: float) -> "LogCoordf64":
        return LogCoordf64(self.logic[0], self.logic[1], base if self.base > 1.0 else self.base, self.zero_point)

    def with_zero_point(self, z: float) -> "LogCoordf64":
        return LogCoordf64(self.logic[0], self.logic[1], self.base, z)

    def map(self, value: float, limit: Tuple[int, int]) -> int:
        # :47-51 through plotters RangedCoordf64::map
        fv = value - self.zero_point
        if self.negative:
            fv = -fv
        lo, hi = self.linear
        t = (math.log(fv) - lo) / (hi - lo)
        return limit[0] + int(math.floor((limit[1] - limit[0]) * t + 1e-3))

    def unmap(self, p: int, limit: Tuple[int, int]) -> Optional[float]:
        # :114-119; plotters RangedCoordf64::unmap: (hi - lo) * ((p - min) / (max - min)) + lo
        mn, mx = limit
        if p < min(mn, mx) or p > max(mn, mx) or mn == mx:
            return None
        lo, hi = self.linear
        off = float(p - mn) / float(mx - mn)
        fv = math.exp((hi - lo) * off + lo)
        if self.negative:
            fv = -fv
        return fv + self.zero_point
