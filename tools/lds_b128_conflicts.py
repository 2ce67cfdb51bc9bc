#!/usr/bin/env python3
"""Bank conflicts of the 16-byte read sides of the plane transposes (stft4096_wg.hip, stft4096_real.hip: `TR`), on the model of the
guide's LDS table: ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, the same + 32), a lane
touches banks (a/4) mod 64 .. +3; ds_write_addtid_b32 writes 64 consecutive words.  Prints the worst number of lanes per bank for
every read of every wave (1 = conflict-free)."""
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def worst(addr_of_tid):
    w = 0
    for wave in range(4):
        for g in GROUPS:
            for c in range(4):
                banks = {}
                for lane in g:
                    a = addr_of_tid(64 * wave + lane) + 4 * c
                    assert a % 4 == 0
                    for b in range(4):
                        banks.setdefault((a + b) % 64, set()).add(lane)
                w = max(w, max(len(v) for v in banks.values()))
    return w


def pos(writer_tid, pad=68):            # word of a writer thread inside a plane row: its wave's 64 words + the pad behind every wave
    return pad * (writer_tid >> 6) + (writer_tid & 63)


def shipped():
    """(name, worst lanes per bank) for every 16-byte read side the kernels ship"""
    return [
        # K1 (4096 points): both read sides are row tid >> 4, the 16 words of writer threads 16 (tid & 15) .. + 15
        ("K1 image 1 / 2, row stride 272", worst(lambda tid: 272 * (tid >> 4) + pos(16 * (tid & 15)))),
        # K1R image 1: the same shape (rows 8 F + q1)
        ("K1R image 1, row stride 272", worst(lambda tid: 272 * (tid >> 4) + pos(16 * (tid & 15)))),
        # K1R image 2: reader tid = 128 F + q1 + 8 q2 reads row q2, writer threads 16 (8 F + q1) .. + 15
        ("K1R image 2, row stride 280", worst(lambda tid: 280 * ((tid & 127) >> 3) + pos(16 * (8 * (tid >> 7) + (tid & 7))))),
    ]


if __name__ == "__main__":
    for name, w in shipped():
        print("%-34s: %d" % (name, w))
    for stride in (272, 276, 288):   # what K1R's second image would do at other strides
        print("K1R image 2, row stride %d (not used): %d" % (stride, worst(lambda tid: stride * ((tid & 127) >> 3) + pos(16 * (8 * (tid >> 7) + (tid & 7))))))
