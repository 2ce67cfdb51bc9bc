#!/usr/bin/env python3
"""K1 into outputs assembled from separately created physical chunks (HIP virtual memory management): does the rate depend on how
the 16.4 GB are laid out physically?  (profiles/r03_k1_slow_box.txt, part 6)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Flags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32", C.c_void_p), ("allocFlags", Flags)]


class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


def ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")


prop = Prop(type=1, requestedHandleType=0, location=Loc(1, 0))
gran = C.c_size_t(0)
ck(hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1), "granularity")
G = gran.value
print(f"granularity {G}", flush=True)


def create(nbytes):
    h = C.c_void_p()
    ck(hip.hipMemCreate(C.byref(h), C.c_size_t(nbytes), C.byref(prop), C.c_ulonglong(0)), f"hipMemCreate({nbytes})")
    return h


def release(h):
    ck(hip.hipMemRelease(h), "release")


class Mapped:
    def __init__(self, handles_sizes):
        self.total = sum(s for _, s in handles_sizes)
        self.ptr = C.c_void_p()
        ck(hip.hipMemAddressReserve(C.byref(self.ptr), C.c_size_t(self.total), C.c_size_t(0), C.c_void_p(0), C.c_ulonglong(0)), "reserve")
        off = 0
        acc = Access(Loc(1, 0), 3)
        for h, s in handles_sizes:      # (access is set per mapping: HIP resolves the range to ONE mapped handle)
            ck(hip.hipMemMap(C.c_void_p(self.ptr.value + off), C.c_size_t(s), C.c_size_t(0), h, C.c_ulonglong(0)), "map")
            ck(hip.hipMemSetAccess(C.c_void_p(self.ptr.value + off), C.c_size_t(s), C.byref(acc), C.c_size_t(1)), "access")
            off += s
        self.hs = handles_sizes
        # every chunk must be reachable before a kernel is pointed at it: a 4-byte copy from the end of each
        probe = C.c_uint32(0)
        off = 0
        for _, s in handles_sizes:
            ck(hip.hipMemcpy(C.byref(probe), C.c_void_p(self.ptr.value + off + s - 4), C.c_size_t(4), 2), "probe copy")
            off += s

    def close(self):
        torch.cuda.synchronize()
        off = 0
        for h, s in self.hs:
            ck(hip.hipMemUnmap(C.c_void_p(self.ptr.value + off), C.c_size_t(s)), "unmap")
            release(h)
            off += s
        ck(hip.hipMemAddressFree(self.ptr, C.c_size_t(self.total)), "free")


F = 1_000_000
eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=1)
pcm = eng.white_noise((F - 1) * 256 + 2048)
n_samples = pcm.numel()
need = F * 2047 * 8
eng.use_current_stream()


def k1(ptr):
    got = C.c_size_t(0)

    def go():
        rc = eng._lib.sgx_stft_batch(eng._ctx, C.c_void_p(pcm.data_ptr()), n_samples, 0, F, C.c_void_p(ptr), C.byref(got))
        assert rc == 0 and got.value == F
    for _ in range(2):
        go()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        go()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 4


def report(name, ptr):
    ms = k1(ptr)
    print(f"{name:44s} {ms:.3f} ms = {F * 17400 / ms / 1e6 / 8000:.3f}", flush=True)


def up(n, g=G):
    return (n + g - 1) // g * g


GiB = 1 << 30
t = torch.empty(need // 4, dtype=torch.float32, device="cuda")
report("hipMalloc (torch.empty)", t.data_ptr())
del t
torch.cuda.empty_cache()

m = Mapped([(create(up(need)), up(need))])
report("one physical allocation", m.ptr.value)
m.close()

for chunk in (2 * GiB, 256 << 20):
    n = (need + chunk - 1) // chunk
    m = Mapped([(create(chunk), chunk) for _ in range(n)])
    report(f"{n} chunks of {chunk >> 20} MiB, created in order", m.ptr.value)
    m.close()

# two halves with 64 GiB between their creation
half = up(need // 2, GiB)
h1 = create(half)
spacer = create(64 * GiB)
h2 = create(half)
release(spacer)
m = Mapped([(h1, half), (h2, half)])
report("two halves created 64 GiB apart", m.ptr.value)
m.close()

for chunk in (GiB, 64 << 20):
    n = (need + 2 * chunk - 1) // (2 * chunk)
    a = [create(chunk) for _ in range(n)]
    spacer = create(64 * GiB)
    b = [create(chunk) for _ in range(n)]
    release(spacer)
    hs = []
    for x, y in zip(a, b):
        hs += [(x, chunk), (y, chunk)]
    m = Mapped(hs)
    report(f"alternating {chunk >> 20} MiB chunks of two far groups", m.ptr.value)
    m.close()
