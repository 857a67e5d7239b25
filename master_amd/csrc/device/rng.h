// rng.h — per-path random stream of the device path.
// The reference PT cannot be seeded (Options.cpp:821-833, Technique.cpp:170-174) and draws from mt19937 (Sample.hpp:9-31), so the stream is
// defined by this build — any counter-keyed stream is equally faithful.  Round 4: a 32-bit generator in ONE register.  The round-3 stream (PCG XSH-RR
// 64/32 seeded by three splitmix64 rounds) cost four 32-bit multiplies and a 64-bit carry chain per draw and twenty-eight multiplies per path on a
// chip without a 64-bit integer multiplier, and two VGPRs across the whole path loop; the C2 kernel is VALU-issue bound and every HBM-resident variant
// runs at its register limit.  Now:
//   state (32 bits);  draw = SplitMix32: state += 0x9E3779B9 (a Weyl sequence), output = the "lowbias32" finaliser of the state (full avalanche, two
//   multiplies);  seed = two rounds of the same finaliser over (seed, pixel, sample) — the seed's own round is wave-uniform (scalar ALU).  A path
//   is a window of ~30 draws at a hashed position of one 2^32-long sequence; windows of different paths meet with probability 2^-27 per pair.
//   (Measured: a second register for a per-pixel increment — PCG-RXS-M-XS-32 — cost the 80-register kernels 4-8 more spilled dwords per lane,
//   profiles/r04/ab_c2_instruction_cuts.txt.)
// oracle/pt_oracle.c states the same definition independently; both are pinned by tests/golden/rng_kat.json.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi {

struct Rng { uint32_t state; };

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x21F0AAADu;
  x ^= x >> 15; x *= 0x735A2D97u;
  x ^= x >> 15; return x;
}
__device__ __forceinline__ Rng rng_seed(uint64_t seed, uint32_t pixel_index, uint64_t sample_index) {
  const uint32_t a = mix32(uint32_t(seed) ^ mix32(uint32_t(seed >> 32) + 0x9E3779B9u));  // wave-uniform: scalar instructions, once per launch
  Rng r;
  r.state = mix32(mix32(a ^ pixel_index) ^ uint32_t(sample_index) ^ (uint32_t(sample_index >> 32) * 0x9E3779B1u));
  return r;
}
__device__ __forceinline__ uint32_t rng_u32(Rng& r) {
  r.state += 0x9E3779B9u;
  return mix32(r.state);
}
// uniform in [0,1), 24 bits (random_generator_t::sample<float>, Sample.inl:259-262)
__device__ __forceinline__ float rng_f(Rng& r) { return float(rng_u32(r) >> 8) * 0x1p-24f; }

// the wavefront pipeline keeps a path's stream in one word of its state arrays
__device__ __forceinline__ uint64_t rng_pack(const Rng& r) { return r.state; }
__device__ __forceinline__ Rng rng_unpack(uint64_t v) { Rng r; r.state = uint32_t(v); return r; }

}  // namespace mi
