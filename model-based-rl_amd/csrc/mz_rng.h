// mz_rng.h -- counter-based RNG for the throughput (non-parity) mode: Philox4x32-10 keyed by the
// engine seed, counter = (env, episode|move, t, purpose).  Parity mode never uses this: there the
// host's numpy draws (np.random.dirichlet / np.random.choice, reference mcts.py:59, config.py:77)
// are passed in.  Host and device run the same integer code, so integer-derived quantities
// (synthetic observations, rewards, uniforms) are bit-identical on both sides.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MZ_RNG_OBS 1u
#define MZ_RNG_REWARD 2u
#define MZ_RNG_DIRICHLET 3u
#define MZ_RNG_ACTION 4u

struct mz_u4 { uint32_t x, y, z, w; };

__host__ __device__ inline mz_u4 mz_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int i = 0; i < 10; ++i) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  mz_u4 r = {c0, c1, c2, c3};
  return r;
}

// uniform double in [0,1) from 53 random bits
__host__ __device__ inline double mz_u01(uint32_t hi, uint32_t lo) {
  uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
  return (double)v * (1.0 / 9007199254740992.0);
}

// Synthetic observation element i of (env, episode, t): Irwin-Hall(4) on 16-bit uniforms, scaled to
// unit variance -- integer sum then ONE float multiply, so host and device agree bit for bit.
__host__ __device__ inline float mz_synth_obs_elem(uint64_t seed, uint32_t env, uint32_t episode, uint32_t t,
                                                  uint32_t i) {
  mz_u4 r = mz_philox(seed, env, episode, t, (MZ_RNG_OBS << 24) | (i >> 1));
  uint32_t a = (i & 1) ? r.z : r.x, b = (i & 1) ? r.w : r.y;
  int32_t s = (int32_t)((a & 0xFFFFu) + (a >> 16) + (b & 0xFFFFu) + (b >> 16)) - 131070;
  return (float)s * 2.6429e-05f;   // 1/(65536*sqrt(4/12)) ~ unit variance
}

// uint8-valued observation element (the -ram- environments: 128 bytes of console RAM, wrappers-free gym obs of dtype
// uint8): byte i of the Philox stream keyed (env, episode, t), as a float32 in 0..255
__host__ __device__ inline float mz_synth_obs_u8(uint64_t seed, uint32_t env, uint32_t episode, uint32_t t, uint32_t i) {
  mz_u4 r = mz_philox(seed, env, episode, t, (MZ_RNG_OBS << 24) | (i >> 4));
  const uint32_t sel = (i >> 2) & 3u;
  const uint32_t word = sel == 0 ? r.x : (sel == 1 ? r.y : (sel == 2 ? r.z : r.w));
  return (float)((word >> (8u * (i & 3u))) & 255u);
}

// reward_t ~ U(-1,1) with 24-bit resolution, exact in float32
__host__ __device__ inline float mz_synth_reward(uint64_t seed, uint32_t env, uint32_t episode, uint32_t t) {
  mz_u4 r = mz_philox(seed, env, episode, t, MZ_RNG_REWARD << 24);
  return (float)(r.x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}
