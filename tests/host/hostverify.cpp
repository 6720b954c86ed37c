// CPU build of the product's host-side verifier (zerokit_amd/csrc/{zkey.cpp,pairing.h}: arkzkey parser, point
// decompression, subgroup check, Groth16 pairing check) behind a tiny C interface, so that tests/test_host_math.py can
// run it against the golden proofs without a GPU and time it.  Test infrastructure only: nothing in the product links
// this.  Built with g++ against the HIP headers (host declarations only; no HIP call is reached).
#include <string.h>

#include <chrono>
#include <vector>

#include "../../zerokit_amd/csrc/zkey.cpp"
#include "pairing.h"
using namespace rlnamd;

static Zkey g_zk;
static bool g_have = false;

extern "C" {
int hv_load_zkey(const uint8_t* data, size_t len) {
  try {
    g_zk = parse_arkzkey(data, len);
    (void)prepared(g_zk);
    g_have = true;
    return 0;
  } catch (const std::exception&) {
    return 1;
  }
}
// proof: 128-byte ark-serialize compressed (A, B, C); pub: n canonical 32-byte LE values.  1 valid, 0 invalid, -1 error
int hv_verify(const uint8_t* proof, const uint8_t* pub, size_t n) {
  if (!g_have) return -1;
  try {
    G1Affine A, C;
    G2Affine B;
    if (!g1_decompress(proof, &A) || !g2_decompress(proof + 32, &B) || !g1_decompress(proof + 96, &C) ||
        !g2_in_subgroup(B))
      return 0;
    std::vector<Fr> x(n);
    for (size_t i = 0; i < n; i++) {
      uint32_t c[8];
      memcpy(c, pub + 32 * i, 32);
      if (limbs_geq(c, FrParams::MOD)) return 0;
      x[i] = Fr::from_canonical(c);
    }
    return groth16_verify(g_zk, A, B, C, x) ? 1 : 0;
  } catch (const std::exception&) {
    return -1;
  }
}
// G2 point as x.c0 | x.c1 | y.c0 | y.c1 canonical LE; bit 0: on the twist, bit 1: psi(P) == [6 u^2] P, bit 2: [r] P == O
int hv_g2_checks(const uint8_t* xy) {
  auto ld = [&](int k) {
    uint32_t c[8];
    memcpy(c, xy + 32 * k, 32);
    return Fq::from_canonical(c);
  };
  G2Affine P{{ld(0), ld(1)}, {ld(2), ld(3)}};
  return (g2_on_curve(P) ? 1 : 0) | (g2_in_subgroup(P) ? 2 : 0) | (g2_in_subgroup_by_order(P) ? 4 : 0);
}
// mean microseconds of `reps` verifications of the same proof
double hv_verify_us(const uint8_t* proof, const uint8_t* pub, size_t n, int reps) {
  auto t0 = std::chrono::steady_clock::now();
  int ok = 0;
  for (int r = 0; r < reps; r++) ok += hv_verify(proof, pub, n);
  auto t1 = std::chrono::steady_clock::now();
  return ok == reps ? std::chrono::duration<double, std::micro>(t1 - t0).count() / reps : -1.0;
}
}
