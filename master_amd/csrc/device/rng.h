// rng.h — per-path random stream of the device path.
// The reference PT cannot be seeded (Options.cpp:821-833, Technique.cpp:170-174) and draws from mt19937 (Sample.hpp:9-31), so the stream is
// defined by this build — any counter-keyed stream is equally faithful.  Round 4: a 32-bit generator.  The round-3 stream (PCG XSH-RR 64/32 seeded
// by three splitmix64 rounds) cost four quarter-rate 32-bit multiplies per draw and twenty-eight per path on a chip without a 64-bit integer
// multiplier; the C2 kernel is VALU-issue bound.  Now:
//   state, inc (32 bits each);  draw = PCG-RXS-M-XS-32 (O'Neill 2014: LCG step x 747796405 + inc, output permutation with one multiply): two
//   multiplies per draw;  seed = two rounds of a 32-bit finaliser ("lowbias32", full avalanche, two multiplies each) over (seed, pixel, sample): the
//   seed's own round is wave-uniform (scalar ALU), the pixel's hash is also the stream's odd increment, so streams of different pixels are
//   different sequences and the samples of a pixel start at hashed positions of theirs.
// oracle/pt_oracle.c states the same definition independently; both are pinned by tests/golden/rng_kat.json.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi {

struct Rng { uint32_t state, inc; };

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x21F0AAADu;
  x ^= x >> 15; x *= 0x735A2D97u;
  x ^= x >> 15; return x;
}
__device__ __forceinline__ Rng rng_seed(uint64_t seed, uint32_t pixel_index, uint64_t sample_index) {
  const uint32_t a = mix32(uint32_t(seed) ^ mix32(uint32_t(seed >> 32) + 0x9E3779B9u));  // wave-uniform: scalar instructions, once per launch
  const uint32_t hp = mix32(a ^ pixel_index);
  Rng r;
  r.inc = hp | 1u;
  r.state = mix32(hp ^ uint32_t(sample_index) ^ (uint32_t(sample_index >> 32) * 0x9E3779B1u));
  return r;
}
__device__ __forceinline__ uint32_t rng_u32(Rng& r) {
  const uint32_t old = r.state;
  // v_mul_lo_u32 + v_add_u32.  Left alone the compiler fuses the step into v_mad_u64_u32, whose 64-bit addend and result cost two more VGPRs held
  // across the whole path loop (the 80-register C2 kernel then spills five dwords per lane): the empty asm keeps the product a 32-bit value of its own.
  uint32_t prod = old * 747796405u;
  asm volatile("" : "+v"(prod));
  r.state = prod + r.inc;
  const uint32_t w = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
  return (w >> 22u) ^ w;
}
// uniform in [0,1), 24 bits (random_generator_t::sample<float>, Sample.inl:259-262)
__device__ __forceinline__ float rng_f(Rng& r) { return float(rng_u32(r) >> 8) * 0x1p-24f; }

// the wavefront pipeline keeps a path's stream in one 64-bit word of its state arrays
__device__ __forceinline__ uint64_t rng_pack(const Rng& r) { return (uint64_t(r.inc) << 32) | r.state; }
__device__ __forceinline__ Rng rng_unpack(uint64_t v) { Rng r; r.state = uint32_t(v); r.inc = uint32_t(v >> 32); return r; }

}  // namespace mi
