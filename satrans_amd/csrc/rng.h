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

// Elements are hashed in BLOCKS of four consecutive indices.  The generator sits inside both fused kernels' VALU budget (a
// quarter of the forward's instructions with the round-2 form: a three-multiply finaliser per block and a xorshift32 step per
// element), and a 32-bit integer multiply is a quarter-rate instruction while the 24-bit multiply-add (v_mad_u32_u24) is full
// rate, so:
//   block word   t = key + block * 0x9E3779 (24-bit multiply-add: block < 2^24);  t ^= t >> 15;  t *= 0x2C1B3C6D;  t ^= t >> 12
//                (ONE 32-bit multiply; key is already a finalised hash of (seed, step, layer, site, sample))
//   elements     u0 = t >> 8 (the top 24 bits), u_{k+1} = (0xF1EA5D u_k + 0x3C6EF3) mod 2^24: a full-period LCG (a = 5 mod 8, c odd),
//                i.e. a bijection of the 24-bit words - every element still sees an exactly uniform word when t is uniform.
// keep <=> u >= p * 2^24.  Checked exhaustively over all 2^24 seeds: each of the 16 keep patterns of a block is within 7.6e-4
// (relative) of its independent-Bernoulli probability; per-block rates and block-pair correlations: tests/test_host_cpu.py.
__host__ __device__ __forceinline__ uint32_t satrans_mul24(uint32_t a, uint32_t b) {      // low 32 bits of a[23:0] * b[23:0]
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#endif
}

__host__ __device__ __forceinline__ uint32_t drop_block_hash(uint32_t sample_key, uint32_t block) {
    uint32_t t = sample_key + satrans_mul24(block, 0x9E3779u);
    t ^= t >> 15;
    t *= 0x2C1B3C6Du;
    t ^= t >> 12;
    return t;
}

__host__ __device__ __forceinline__ uint32_t satrans_lcg24(uint32_t u) {
    return (satrans_mul24(u, 0xF1EA5Du) + 0x3C6EF3u) & 0xFFFFFFu;
}

// keep flags of the elements 4*block .. 4*block+3 in bits 0..3
__host__ __device__ __forceinline__ uint32_t drop_keep4(uint32_t sample_key, uint32_t block, uint32_t thresh24) {
    uint32_t u = drop_block_hash(sample_key, block) >> 8;
    uint32_t bits = u >= thresh24 ? 1u : 0u;
    u = satrans_lcg24(u);
    bits |= u >= thresh24 ? 2u : 0u;
    u = satrans_lcg24(u);
    bits |= u >= thresh24 ? 4u : 0u;
    u = satrans_lcg24(u);
    bits |= u >= thresh24 ? 8u : 0u;
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
