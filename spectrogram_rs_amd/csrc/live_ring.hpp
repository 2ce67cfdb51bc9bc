// live_ring.hpp -- the bookkeeping of the capture ring, host code only (no HIP call): what the audio callback and the GUI tick share.
//
//   producer   audio_input_list_model.rs:63-75   cpal callback -> HeapRb::push_iter (mono -> (s, s), overflow dropped)
//   consumer   audio_transform.rs:34-42           hop loop over the ring (peek W, skip H)
//
// Sample positions are counted in (l, r) pairs since creation:
//   skipped <= uploaded <= pushed <= skipped + capacity
//   [skipped, uploaded)   resident on the consumer's side (the device image of sgx_live.hip), at offset position - skipped
//   [uploaded, pushed)    in the host ring, slot = position % capacity
// The producer writes `pushed` (release) after the slots; the consumer writes `skipped` (release) after it has finished with the
// slots it frees; each reads the other's counter with acquire.  `uploaded` belongs to the consumer alone.  Lock-free, one producer
// thread and one consumer thread (the reference's HeapRb is the same single-producer single-consumer ring behind an Arc<Mutex>).
//
// Kept free of HIP so that it is built and run under -fsanitize=thread and -fsanitize=address,undefined on the CPU
// (tests/cpp/host_sanitize.cpp, tests/test_host_sanitizers.py); sgx_live.hip adds the copies and the launches.
#pragma once

#include <atomic>
#include <cstddef>
#include <cstdint>

namespace sgx {

struct RingPair { float l, r; };   // StereoMagnitude-shaped sample, src/fourier/mod.rs:13 (same layout as float2)

struct LiveRingState {
    RingPair *slots = nullptr;      // [capacity], owned by the caller (pinned host memory in the product)
    size_t capacity = 0;
    std::atomic<unsigned long long> pushed{0}, skipped{0};
    unsigned long long uploaded = 0;

    // ---- producer side -------------------------------------------------------------------------------------------------------
    // the input callback: interleaved values, 1 or 2 channels; returns the pairs accepted (what does not fit is dropped, :72)
    size_t push(const float *h_samples, size_t n_values, uint32_t channels)
    {
        const unsigned long long head = pushed.load(std::memory_order_relaxed);
        const unsigned long long tail = skipped.load(std::memory_order_acquire);
        const size_t vacant = capacity - (size_t)(head - tail);
        size_t n = channels == 1 ? n_values : n_values / 2;  // tuples() drops a trailing odd value (:71)
        if (n > vacant) n = vacant;
        size_t slot = (size_t)(head % capacity);
        for (size_t i = 0; i < n; ++i) {
            slots[slot] = channels == 1 ? RingPair{h_samples[i], h_samples[i]}  // :67-69
                                        : RingPair{h_samples[2 * i], h_samples[2 * i + 1]};
            if (++slot == capacity) slot = 0;
        }
        pushed.store(head + n, std::memory_order_release);
        return n;
    }

    size_t occupied() const
    {
        const unsigned long long tail = skipped.load(std::memory_order_acquire);
        const unsigned long long head = pushed.load(std::memory_order_acquire);
        return (size_t)(head - tail);
    }

    // ---- consumer side -------------------------------------------------------------------------------------------------------
    // What a tick has to move to the consumer's image: slots [slot, slot + first) then [0, second), to image offset `dst`.
    struct Upload { size_t dst, slot, first, second, occupied; unsigned long long tail; };
    Upload begin_tick()
    {
        const unsigned long long tail = skipped.load(std::memory_order_relaxed);
        const unsigned long long head = pushed.load(std::memory_order_acquire);
        Upload u{0, 0, 0, 0, (size_t)(head - tail), tail};
        if (head > uploaded) {
            const size_t fresh = (size_t)(head - uploaded);
            u.slot = (size_t)(uploaded % capacity);
            u.first = fresh < capacity - u.slot ? fresh : capacity - u.slot;
            u.second = fresh - u.first;
            u.dst = (size_t)(uploaded - tail);
            uploaded = head;
        }
        return u;
    }

    // ring.skip(H) per yielded frame; the reference's loop also skips on the read that returns None (audio_transform.rs:37-41)
    // unless the caller's max_frames ended this tick early; HeapRb::skip stops at the end of the ring
    static size_t skip_of(size_t frames, size_t H, bool reference_skip, bool truncated, size_t occupied)
    {
        size_t skip = frames * H;
        if (reference_skip && !truncated) skip += H;
        return skip > occupied ? occupied : skip;
    }

    // the slots uploaded by this tick are reusable from here on: call only after the copies out of them have completed
    void end_tick(const Upload &u, size_t skip) { skipped.store(u.tail + skip, std::memory_order_release); }
};

}  // namespace sgx
