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

// Elements are hashed in BLOCKS of four consecutive indices (a 32-bit integer multiply is a quarter-rate instruction
// and the finaliser needs three of them): the block's hash gives the uniform of its first element, one xorshift32 step
// each of the next three.  xorshift32 is a bijection on the non-zero words, so every element still sees a uniform
// 32-bit word.  keep <=> its top 24 bits >= p * 2^24.
__host__ __device__ __forceinline__ uint32_t satrans_xs32(uint32_t x) {
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
}

__host__ __device__ __forceinline__ uint32_t drop_block_hash(uint32_t sample_key, uint32_t block) {
    return satrans_mix32(sample_key ^ (block * 0x27D4EB2Fu));
}

// keep flags of the elements 4*block .. 4*block+3 in bits 0..3
__host__ __device__ __forceinline__ uint32_t drop_keep4(uint32_t sample_key, uint32_t block, uint32_t thresh24) {
    uint32_t h = drop_block_hash(sample_key, block);
    uint32_t bits = (h >> 8) >= thresh24 ? 1u : 0u;
    h = satrans_xs32(h);
    bits |= (h >> 8) >= thresh24 ? 2u : 0u;
    h = satrans_xs32(h);
    bits |= (h >> 8) >= thresh24 ? 4u : 0u;
    h = satrans_xs32(h);
    bits |= (h >> 8) >= thresh24 ? 8u : 0u;
    return bits;
}

// one element (generic kernels)
__host__ __device__ __forceinline__ bool drop_keep(uint32_t sample_key, uint32_t elem, uint32_t thresh24) {
    return (drop_keep4(sample_key, elem >> 2, thresh24) >> (elem & 3u)) & 1u;
}

// element index of attention probability (head h, query i, key j): rows padded to a multiple of four keys so that a row
// starts a block
__host__ __device__ __forceinline__ uint32_t drop_attn_elem(int h, int F, int i, int j) {
    return (uint32_t)((h * F + i) * ((F + 3) & ~3) + j);
}

__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) { return (uint32_t)(p * 16777216.0f); }

}  // namespace satrans
