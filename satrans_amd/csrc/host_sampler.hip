// Host side of fit(): the sample order of a shuffled epoch, drawn the way the reference's loader draws it
// (models/meta_basemodel.py:279-280: DataLoader(shuffle=True) -> RandomSampler -> torch.randperm(n, generator=g) with g seeded
// from the global generator).  torch's CPU randperm is a Fisher-Yates pass over 0..n-1 with one 32-bit Mersenne-Twister draw per
// position, z = draw % (n - i), swap(r[i], r[i + z]); it runs at 11-75 ns per row because every swap partner is a cache miss
// that the next iteration waits for - 17 ms for the 1.6 M rows of the bench's fit leg, 1.4 s for AliCCP's 42 M, in front of an
// epoch's first step.  The swap POSITIONS do not depend on the data, so this pass draws them a block ahead and prefetches the
// partners: the same permutation bit for bit (satrans_amd/basemodel.py checks that against torch.randperm once per process and
// falls back to torch when it differs), several times faster, and positions [0, i] are final once iteration i is done -
// `progress` publishes how far that is for a consumer that wants to start on the head of the order.
#include <atomic>
#include <cstdint>
#include <vector>

#include "../../include/satrans_hip.h"
#include "common.h"

namespace {

// MT19937 as torch seeds and steps it (at::mt19937: 32-bit seeding by the Knuth multiplier, the standard twist and tempering)
struct Mt {
    uint32_t s[624];
    int at;
    explicit Mt(uint64_t seed) {
        s[0] = (uint32_t)(seed & 0xffffffffu);
        for (int j = 1; j < 624; ++j) s[j] = 1812433253u * (s[j - 1] ^ (s[j - 1] >> 30)) + (uint32_t)j;
        at = 624;
    }
    void twist() {
        for (int k = 0; k < 624; ++k) {
            const uint32_t y = (s[k] & 0x80000000u) | (s[(k + 1) % 624] & 0x7fffffffu);
            s[k] = s[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        at = 0;
    }
    uint32_t next() {
        if (at == 624) twist();
        uint32_t y = s[at++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};

}  // namespace

extern "C" int satrans_host_randperm(uint64_t seed, int64_t n, int64_t* out, int64_t* progress) {
    SATRANS_REQUIRE(n >= 0 && (out || n == 0), SATRANS_E_BADARG, "host_randperm: bad arguments");
    // (torch switches to another algorithm at n >= 2^32 / 20: the caller keeps torch.randperm there)
    SATRANS_REQUIRE(n < (int64_t)(0xffffffffu / 20), SATRANS_E_UNSUPPORTED, "host_randperm: n = %lld is beyond the 32-bit form", (long long)n);
    std::atomic<int64_t>* done = reinterpret_cast<std::atomic<int64_t>*>(progress);
    for (int64_t i = 0; i < n; ++i) out[i] = i;
    Mt g(seed);
    constexpr int kBlock = 64;
    int64_t z[2][kBlock];
    auto draw = [&](int64_t i0, int64_t (&dst)[kBlock]) {      // swap partners of positions [i0, i0 + kBlock), prefetched
        for (int k = 0; k < kBlock; ++k) {
            const int64_t i = i0 + k;
            if (i >= n - 1) break;
            dst[k] = i + (int64_t)(g.next() % (uint32_t)(n - i));
            __builtin_prefetch(out + dst[k], 1, 1);
        }
    };
    int cur = 0;
    if (n > 1) draw(0, z[0]);
    for (int64_t i0 = 0; i0 < n - 1; i0 += kBlock) {
        if (i0 + kBlock < n - 1) draw(i0 + kBlock, z[cur ^ 1]);
        const int64_t hi = i0 + kBlock < n - 1 ? i0 + kBlock : n - 1;
        for (int64_t i = i0; i < hi; ++i) {
            const int64_t j = z[cur][i - i0];
            const int64_t sav = out[i];
            out[i] = out[j];
            out[j] = sav;
        }
        cur ^= 1;
        if (done && (i0 & 0xffff) == 0) done->store(hi, std::memory_order_release);
    }
    if (done) done->store(n, std::memory_order_release);
    return SATRANS_OK;
}
