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

// A 64-bit constant in scalar registers, materialised where it is used.  gfx9 VOP3 takes no literals, so the compiler puts the multipliers of
// the 64-bit products below into VGPR pairs, hoists them out of the path loop (they are loop invariants) and — the loop needs every VGPR —
// spills them to scratch: 28 of the megakernel's 60 bytes of scratch per lane were these constants, reloaded from memory every trip.  The
// empty asm makes the value opaque (not hoistable, not foldable) and pins it to SGPRs (two s_mov_b32 at the use).
__device__ __forceinline__ uint64_t sconst(uint64_t k) {
  asm volatile("" : "+s"(k));
  return k;
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z ^= z >> 30; z *= sconst(0xBF58476D1CE4E5B9ULL);
  z ^= z >> 27; z *= sconst(0x94D049BB133111EBULL);
  z ^= z >> 31; return z;
}
__device__ __forceinline__ Rng rng_seed(uint64_t seed, uint32_t pixel_index, uint64_t sample_index) {
  uint64_t h = splitmix64(seed + 0x9E3779B97F4A7C15ULL);
  h = splitmix64(h ^ uint64_t(pixel_index));
  h = splitmix64(h ^ sample_index);
  Rng r; r.state = h * sconst(MI_PCG_MULT) + sconst(MI_PCG_INC);
  return r;
}
__device__ __forceinline__ uint32_t rng_u32(Rng& r) {
  uint64_t old = r.state;
  r.state = old * sconst(MI_PCG_MULT) + sconst(MI_PCG_INC);
  uint32_t xorshifted = uint32_t(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = uint32_t(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}
// uniform in [0,1), 24 bits (random_generator_t::sample<float>, Sample.inl:259-262)
__device__ __forceinline__ float rng_f(Rng& r) { return float(rng_u32(r) >> 8) * 0x1p-24f; }

}  // namespace mi
