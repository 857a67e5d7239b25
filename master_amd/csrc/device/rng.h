// rng.h — per-path random stream of the device path.
// The reference PT cannot be seeded (Options.cpp:821-833, Technique.cpp:170-174), so the
// stream is defined by this build: PCG32 (XSH-RR 64/32) whose state is seeded from
// splitmix64 over (seed, pixel index, sample index).  oracle/pt_oracle.c states the same
// definition independently; both are pinned by tests/golden/rng_kat.json.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi {

struct Rng { uint64_t state; };

#define MI_PCG_MULT 6364136223846793005ULL
#define MI_PCG_INC 0xDA3E39CB94B95BDBULL

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27; z *= 0x94D049BB133111EBULL;
  z ^= z >> 31; return z;
}
__device__ __forceinline__ Rng rng_seed(uint64_t seed, uint32_t pixel_index, uint64_t sample_index) {
  uint64_t h = splitmix64(seed + 0x9E3779B97F4A7C15ULL);
  h = splitmix64(h ^ uint64_t(pixel_index));
  h = splitmix64(h ^ sample_index);
  Rng r; r.state = h * MI_PCG_MULT + MI_PCG_INC;
  return r;
}
__device__ __forceinline__ uint32_t rng_u32(Rng& r) {
  uint64_t old = r.state;
  r.state = old * MI_PCG_MULT + MI_PCG_INC;
  uint32_t xorshifted = uint32_t(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = uint32_t(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}
// uniform in [0,1), 24 bits (random_generator_t::sample<float>, Sample.inl:259-262)
__device__ __forceinline__ float rng_f(Rng& r) { return float(rng_u32(r) >> 8) * 0x1p-24f; }

}  // namespace mi
