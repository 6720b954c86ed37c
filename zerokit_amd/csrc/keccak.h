// Keccak-256 (original 0x01 padding == tiny-keccak `Keccak::v256`) and the reference's hash_to_field
// (/root/reference/rln/src/hashers.rs:73-93).  Host helper: O(1) per call, not on the proving path.
#pragma once
#include <stdint.h>
#include <string.h>

#include "field.h"

namespace rlnamd {

inline void keccak_f1600(uint64_t st[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
      0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
      0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static const int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  auto rol = [](uint64_t x, int n) { return (x << n) | (x >> (64 - n)); };
  for (int round = 0; round < 24; round++) {
    uint64_t bc[5];
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      uint64_t t = bc[(i + 4) % 5] ^ rol(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    uint64_t t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = PILN[i];
      uint64_t b = st[j];
      st[j] = rol(t, ROTC[i]);
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= RC[round];
  }
}

inline void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  const size_t rate = 136;
  uint64_t st[25];
  memset(st, 0, sizeof(st));
  uint8_t block[136];
  while (len >= rate) {
    for (size_t i = 0; i < rate / 8; i++) {
      uint64_t w;
      memcpy(&w, data + 8 * i, 8);
      st[i] ^= w;
    }
    keccak_f1600(st);
    data += rate;
    len -= rate;
  }
  memset(block, 0, rate);
  memcpy(block, data, len);
  block[len] ^= 0x01;
  block[rate - 1] ^= 0x80;
  for (size_t i = 0; i < rate / 8; i++) {
    uint64_t w;
    memcpy(&w, block + 8 * i, 8);
    st[i] ^= w;
  }
  keccak_f1600(st);
  memcpy(out, st, 32);
}

// little-endian 256-bit integer mod r -> canonical LE bytes (2^256 < 6r: at most 5 subtractions)
inline void reduce_mod_r_le(const uint8_t in[32], uint8_t out[32]) {
  uint32_t v[8];
  memcpy(v, in, 32);
  while (limbs_geq(v, FrParams::MOD)) {
    uint32_t borrow = 0;
    for (int i = 0; i < 8; i++) {
      uint64_t s = (uint64_t)v[i] - FrParams::MOD[i] - borrow;
      v[i] = (uint32_t)s;
      borrow = (uint32_t)(s >> 63);
    }
  }
  memcpy(out, v, 32);
}

inline void hash_to_field_le(const uint8_t* data, size_t len, uint8_t out[32]) {
  uint8_t h[32];
  keccak256(data, len, h);
  reduce_mod_r_le(h, out);
}
// hashers.rs:84-93: digest reversed and read big-endian == the little-endian reading of the digest
inline void hash_to_field_be(const uint8_t* data, size_t len, uint8_t out[32]) { hash_to_field_le(data, len, out); }

}  // namespace rlnamd
