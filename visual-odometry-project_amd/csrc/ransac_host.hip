// Host-side RANSAC control that must reproduce the reference bit for bit:
//  * the sample stream of  np.random.default_rng(2023).choice(np.arange(n),
//    replace=False, size=s)   (src/vo/algorithms/ransac.py:52, 92-94): NumPy's
//    PCG64 (XSL-RR 128/64, 32-bit halves buffered low-then-high), Lemire bounded
//    integers, Floyd's sampling with NumPy's open-addressed hash set, and the final
//    Fisher-Yates shuffle of the s draws;
//  * the sequential accept / adaptive-bound rule (ransac.py:58-67, 90-121), replayed
//    over the per-hypothesis (valid, inlier count) that the GPU computed in bulk.
// Pure host code (no device work); compiled into libvo_hip.so.
#include <cmath>

#include "vo_internal.h"

namespace {

typedef unsigned __int128 u128;

inline u128 make128(uint64_t hi, uint64_t lo) { return ((u128)hi << 64) | lo; }

const u128 PCG_MULT = ((u128)2549297995355413924ULL << 64) | 4865540595714422341ULL;

inline uint64_t rotr64(uint64_t v, unsigned r) { return (v >> r) | (v << ((-r) & 63)); }

struct pcg {
  u128 state, inc;
  uint32_t has32, u32;
  uint64_t next64() {
    state = state * PCG_MULT + inc;
    return rotr64((uint64_t)(state >> 64) ^ (uint64_t)state, (unsigned)(state >> 122));
  }
  uint32_t next32() {
    if (has32) {
      has32 = 0;
      return u32;
    }
    uint64_t n = next64();
    has32 = 1;
    u32 = (uint32_t)(n >> 32);
    return (uint32_t)(n & 0xffffffffu);
  }
  // numpy random_bounded_uint64(off=0, rng, mask=0, masked=false)
  uint64_t bounded(uint64_t rng) {
    if (rng == 0) return 0;
    if (rng <= 0xFFFFFFFFull) {
      if (rng == 0xFFFFFFFFull) return next32();
      const uint32_t r = (uint32_t)rng, rex = r + 1;
      uint64_t m = (uint64_t)next32() * rex;
      uint32_t left = (uint32_t)m;
      if (left < rex) {
        const uint32_t thr = (0xFFFFFFFFu - r) % rex;
        while (left < thr) {
          m = (uint64_t)next32() * rex;
          left = (uint32_t)m;
        }
      }
      return m >> 32;
    }
    if (rng == 0xFFFFFFFFFFFFFFFFull) return next64();
    const uint64_t rex = rng + 1;
    u128 m = (u128)next64() * rex;
    uint64_t left = (uint64_t)m;
    if (left < rex) {
      const uint64_t thr = (0xFFFFFFFFFFFFFFFFull - rng) % rex;
      while (left < thr) {
        m = (u128)next64() * rex;
        left = (uint64_t)m;
      }
    }
    return (uint64_t)(m >> 64);
  }
};

pcg load(const vo_pcg64* r) {
  pcg p;
  p.state = make128(r->state_hi, r->state_lo);
  p.inc = make128(r->inc_hi, r->inc_lo);
  p.has32 = r->has_uint32;
  p.u32 = r->uinteger;
  return p;
}

void store(const pcg& p, vo_pcg64* r) {
  r->state_hi = (uint64_t)(p.state >> 64);
  r->state_lo = (uint64_t)p.state;
  r->inc_hi = (uint64_t)(p.inc >> 64);
  r->inc_lo = (uint64_t)p.inc;
  r->has_uint32 = p.has32;
  r->uinteger = p.u32;
}

// one Generator.choice(arange(pop), replace=False, size=s) with shuffle=True (Floyd branch)
void choice_floyd(pcg& g, int64_t pop, int s, int64_t* out, uint64_t* hash, uint64_t mask) {
  for (uint64_t i = 0; i <= mask; ++i) hash[i] = ~0ull;
  for (int64_t j = pop - s; j < pop; ++j) {
    const uint64_t val = g.bounded((uint64_t)j);
    uint64_t loc = val & mask;
    while (hash[loc] != ~0ull && hash[loc] != val) loc = (loc + 1) & mask;
    if (hash[loc] == ~0ull) {
      hash[loc] = val;
      out[j - pop + s] = (int64_t)val;
    } else {
      loc = (uint64_t)j & mask;
      while (hash[loc] != ~0ull) loc = (loc + 1) & mask;
      hash[loc] = (uint64_t)j;
      out[j - pop + s] = j;
    }
  }
  for (int64_t i = s - 1; i >= 1; --i) {
    const int64_t j = (int64_t)g.bounded((uint64_t)i);
    const int64_t tmp = out[j];
    out[j] = out[i];
    out[i] = tmp;
  }
}

// The same draws for small s: the hash set only answers "was this value drawn before", which
// for a handful of values is a few compares; the output is identical.
template <int S>
inline void choice_small(pcg& g, uint32_t pop, int32_t* out) {
  int32_t v[S];
#pragma unroll
  for (int k = 0; k < S; ++k) {
    const uint32_t j = pop - S + k;
    const uint32_t val = (uint32_t)g.bounded(j);
    bool seen = false;
#pragma unroll
    for (int q = 0; q < k; ++q) seen |= (v[q] == (int32_t)val);
    v[k] = seen ? (int32_t)j : (int32_t)val;
  }
#pragma unroll
  for (int i = S - 1; i >= 1; --i) {
    const int j = (int)g.bounded((uint64_t)i);
    const int32_t tmp = v[j];
    v[j] = v[i];
    v[i] = tmp;
  }
#pragma unroll
  for (int k = 0; k < S; ++k) out[k] = v[k];
}

int64_t n_iterations_for(double confidence, double outlier_ratio, int s) {
  // ransac.py:64-67: int(np.ceil(np.log(1 - conf) / np.log(1 - (1 - outlier_ratio) ** s)))
  const double k = std::ceil(std::log(1.0 - confidence) / std::log(1.0 - std::pow(1.0 - outlier_ratio, (double)s)));
  return (int64_t)k;
}

}  // namespace

void vo_rng_raw32(vo_pcg64* rng, int count, uint32_t* out) {
  pcg g = load(rng);
  for (int i = 0; i < count; ++i) out[i] = g.next32();
  store(g, rng);
}

extern "C" {

int vo_rng_choice(vo_pcg64* rng, int pop, int s, int count, int32_t* out) {
  if (!rng || !out || pop < 1 || s < 1 || s > pop || s > 64 || count < 0) return VO_EINVAL;
  // NumPy switches to a tail shuffle when pop > 10000 and s > pop // 50; not reachable for s <= 64
  if (pop > 10000 && s > pop / 50) return VO_EINVAL;
  uint64_t set_size = (uint64_t)(1.2 * (double)s);
  uint64_t mask = set_size;
  mask |= mask >> 1;
  mask |= mask >> 2;
  mask |= mask >> 4;
  mask |= mask >> 8;
  mask |= mask >> 16;
  mask |= mask >> 32;
  uint64_t hash[128];
  int64_t idx[64];
  pcg g = load(rng);
  if (s == 4 && pop >= 8) {
    for (int c = 0; c < count; ++c) choice_small<4>(g, (uint32_t)pop, out + (size_t)c * 4);
    store(g, rng);
    return VO_OK;
  }
  for (int c = 0; c < count; ++c) {
    choice_floyd(g, pop, s, idx, hash, mask);
    for (int k = 0; k < s; ++k) out[(size_t)c * s + k] = (int32_t)idx[k];
  }
  store(g, rng);
  return VO_OK;
}

int64_t vo_ransac_num_iterations(double confidence, double outlier_ratio, int s) {
  return n_iterations_for(confidence, outlier_ratio, s);
}

// Replays ransac.py:90-121 over pre-computed hypotheses.  `st` carries the
// reference object's persistent fields; `n_done`/`best_*` carry the loop state so a
// second batch can continue the same find_best_model call.  Returns the number of
// hypotheses consumed from this batch in *consumed; *finished = 1 when the
// reference loop would have exited.
int vo_ransac_replay(vo_ransac_state* st, const uint8_t* valid, const int32_t* counts, int B, int N,
                     int64_t* n_done, int32_t* best_count, int32_t* best_idx, int idx_offset, int* consumed,
                     int* finished) {
  if (!st || !valid || !counts || !n_done || !best_count || !best_idx || !consumed || !finished || B < 0 || N < 1)
    return VO_EINVAL;
  int b = 0;
  *finished = 0;
  while (*n_done < st->n_iterations) {
    if (b >= B) {
      *consumed = b;
      return VO_OK;
    }
    const int cur = b++;
    if (!valid[cur]) continue;   // model is None: `continue` without counting the iteration
    const int32_t c = counts[cur];
    if (c > *best_count) {
      *best_count = c;
      *best_idx = idx_offset + cur;
      if (st->adaptive) {
        double orat = 1.0 - (double)c / (double)N;
        orat = std::fmin(std::fmax(orat, 0.01), 0.99);
        st->outlier_ratio = orat;
        const int64_t k = n_iterations_for(st->confidence, orat, st->s);
        st->n_iterations = (st->max_iterations >= 0 && st->max_iterations < k) ? st->max_iterations : k;
      }
    }
    *n_done += 1;
  }
  *consumed = b;
  *finished = 1;
  return VO_OK;
}

}  // extern "C"
