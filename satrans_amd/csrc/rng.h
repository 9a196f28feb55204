// Counter-based dropout for the four dropout sites of a layer (reference satrans.py:27-28,87,94 and
// submodules.py:97, all p = 0.1).  torch's CPU generator stream cannot be replayed inside a fused kernel, so
// a mask bit is a pure function of (seed, step, layer, site, sample, element): the backward kernel
// regenerates exactly the forward's mask and nothing is stored.  oracle/satrans_oracle.py:dropout_keep
// restates this function in numpy so tests can feed the same masks to the CPU oracle.
#pragma once
#include <stdint.h>

namespace satrans {

enum DropSite : int { kSiteMetaQ = 0, kSiteMetaK = 1, kSiteAttn = 2, kSiteOut = 3 };

__host__ __device__ __forceinline__ uint32_t satrans_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

// key shared by every element of one (step, layer, site)
__host__ __device__ __forceinline__ uint32_t drop_site_key(uint32_t seed, uint32_t step, int layer, int site) {
    uint32_t key = satrans_mix32(seed ^ (step * 0x9E3779B9u));
    return satrans_mix32(key ^ ((uint32_t)(layer * 4 + site + 1) * 0x85EBCA6Bu));
}

// key of one sample; `sample` is the index inside the GLOBAL batch so that the mask does not depend on how
// samples are tiled over workgroups (or sharded over ranks: callers add their rank's batch offset)
__host__ __device__ __forceinline__ uint32_t drop_sample_key(uint32_t site_key, uint32_t sample) {
    return satrans_mix32((sample * 0xC2B2AE35u) ^ site_key);
}

// keep <=> 24-bit uniform >= p * 2^24
__host__ __device__ __forceinline__ bool drop_keep(uint32_t sample_key, uint32_t elem, uint32_t thresh24) {
    return (satrans_mix32(sample_key ^ (elem * 0x27D4EB2Fu)) >> 8) >= thresh24;
}

__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) { return (uint32_t)(p * 16777216.0f); }

}  // namespace satrans
