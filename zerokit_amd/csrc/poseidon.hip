#include "poseidon.h"

#include <string.h>

namespace rlnamd {

// ---------------------------------------------------------------- Grain LFSR (poseidon_constants.rs:15-205)
namespace {
struct Grain {
  bool st[80];
  int head = 0;
  int nbits;
  Grain(int prime_bits, int t, int rf, int rp) : nbits(prime_bits) {
    memset(st, 0, sizeof(st));
    st[1] = true;  // field; s-box bits stay 0 (x^alpha)
    auto put = [&](int lo, int hi, uint64_t v) {
      for (int i = hi; i >= lo; i--) {
        st[i] = v & 1;
        v >>= 1;
      }
    };
    put(6, 17, prime_bits);
    put(18, 29, t);
    put(30, 39, rf);
    put(40, 49, rp);
    for (int i = 50; i < 80; i++) st[i] = true;
    for (int i = 0; i < 160; i++) update();
  }
  bool update() {
    bool b = st[(head + 62) % 80] ^ st[(head + 51) % 80] ^ st[(head + 38) % 80] ^ st[(head + 23) % 80] ^
             st[(head + 13) % 80] ^ st[head];
    st[head] = b;
    head = (head + 1) % 80;
    return b;
  }
  // n bits, first generated bit is the most significant (get_bits + reverse + LE packing in the reference)
  void value(uint32_t* limbs) {
    for (int i = 0; i < 8; i++) limbs[i] = 0;
    for (int k = nbits - 1; k >= 0; k--) {
      bool b = update();
      while (!b) {
        update();
        b = update();
      }
      if (update()) limbs[k >> 5] |= 1u << (k & 31);
    }
  }
  Fr rejection() {
    uint32_t v[8];
    for (;;) {
      value(v);
      if (!limbs_geq(v, FrParams::MOD)) return Fr::from_canonical(v);
    }
  }
  Fr mod_p() {
    uint32_t v[8];
    value(v);  // < 2^254 < 2r: one conditional subtraction == from_le_bytes_mod_order
    if (limbs_geq(v, FrParams::MOD)) Fr::reduce_once(v);
    return Fr::from_canonical(v);
  }
};
// rln/src/hashers.rs:14-23
const int kRoundParams[][4] = {{2, 8, 56, 0}, {3, 8, 57, 0}, {4, 8, 56, 0}, {5, 8, 60, 0},
                               {6, 8, 60, 0}, {7, 8, 63, 0}, {8, 8, 64, 0}, {9, 8, 63, 0}};
}  // namespace

PoseidonParams poseidon_derive_params(int t) {
  const int* rp = nullptr;
  for (auto& p : kRoundParams)
    if (p[0] == t) rp = p;
  if (!rp) throw Error("no Poseidon parameters for input length " + std::to_string(t - 1));
  PoseidonParams P;
  P.t = t;
  P.rf = rp[1];
  P.rp = rp[2];
  Grain g(254, t, P.rf, P.rp);
  for (int i = 0; i < (P.rf + P.rp) * t; i++) P.ark.push_back(g.rejection());
  for (int k = 0; k < rp[3]; k++)
    for (int i = 0; i < 2 * t; i++) g.mod_p();
  std::vector<Fr> xs, ys;
  for (int i = 0; i < t; i++) xs.push_back(g.mod_p());
  for (int i = 0; i < t; i++) ys.push_back(g.mod_p());
  P.mds.resize(t * t);
  for (int i = 0; i < t; i++)
    for (int j = 0; j < t; j++) P.mds[i * t + j] = (xs[i] + ys[j]).inv();
  return P;
}

__global__ void k_fr_to29(const Fr* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fr29 v = Fr29::from_fq(src[t]);
  v.normalize();
#pragma unroll
  for (int k = 0; k < 9; k++) dst[t * 9 + k] = v.v[k];
}

void PoseidonDev::init() {
  if (ready) return;
  require_gpu();
  for (int t = 2; t <= POSEIDON_MAX_T; t++) {
    PoseidonParams P = poseidon_derive_params(t);
    rf[t] = P.rf;
    rp[t] = P.rp;
    ark[t].alloc(P.ark.size());
    mds[t].alloc(P.mds.size());
    RLN_HIP(hipMemcpy(ark[t].p, P.ark.data(), P.ark.size() * sizeof(Fr), hipMemcpyHostToDevice));
    RLN_HIP(hipMemcpy(mds[t].p, P.mds.data(), P.mds.size() * sizeof(Fr), hipMemcpyHostToDevice));
    ark29[t].alloc(P.ark.size() * 9);
    mds29[t].alloc(P.mds.size() * 9);
    hipLaunchKernelGGL(k_fr_to29, dim3(div_up(P.ark.size(), 64)), dim3(64), 0, 0, ark[t].p, ark29[t].p, (uint32_t)P.ark.size());
    hipLaunchKernelGGL(k_fr_to29, dim3(div_up(P.mds.size(), 64)), dim3(64), 0, 0, mds[t].p, mds29[t].p, (uint32_t)P.mds.size());
    RLN_HIP(hipDeviceSynchronize());
  }
  ready = true;
}

PoseidonDev& poseidon_dev() {
  static PoseidonDev d;
  d.init();
  return d;
}

PoseidonView poseidon_view(int t) {
  if (t < 2 || t > POSEIDON_MAX_T) throw Error("unsupported Poseidon width t=" + std::to_string(t));
  PoseidonDev& d = poseidon_dev();
  return {d.ark[t].p, d.mds[t].p, d.rf[t], d.rp[t], d.ark29[t].p, d.mds29[t].p};
}

template <int T>
__global__ void __launch_bounds__(256) k_poseidon_batch(const uint32_t* __restrict__ in, size_t n,
                                                        uint32_t* __restrict__ out, PoseidonView pv) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr x[T - 1];
#pragma unroll
  for (int j = 0; j < T - 1; j++) x[j] = Fr::from_canonical(in + (i * (T - 1) + j) * 8);
  Fr h = poseidon_hash_dev<T>(x, pv);
  h.to_canonical(out + i * 8);
}

void poseidon_hash_batch_device(const uint8_t* d_in, size_t n, int arity, uint8_t* d_out, hipStream_t s) {
  if (n == 0) return;
  PoseidonView pv = poseidon_view(arity + 1);
  dim3 grid(div_up(n, 256)), block(256);
  const uint32_t* in = (const uint32_t*)d_in;
  uint32_t* out = (uint32_t*)d_out;
  switch (arity) {
    case 1: hipLaunchKernelGGL(k_poseidon_batch<2>, grid, block, 0, s, in, n, out, pv); break;
    case 2: hipLaunchKernelGGL(k_poseidon_batch<3>, grid, block, 0, s, in, n, out, pv); break;
    case 3: hipLaunchKernelGGL(k_poseidon_batch<4>, grid, block, 0, s, in, n, out, pv); break;
    case 4: hipLaunchKernelGGL(k_poseidon_batch<5>, grid, block, 0, s, in, n, out, pv); break;
    case 5: hipLaunchKernelGGL(k_poseidon_batch<6>, grid, block, 0, s, in, n, out, pv); break;
    case 6: hipLaunchKernelGGL(k_poseidon_batch<7>, grid, block, 0, s, in, n, out, pv); break;
    case 7: hipLaunchKernelGGL(k_poseidon_batch<8>, grid, block, 0, s, in, n, out, pv); break;
    case 8: hipLaunchKernelGGL(k_poseidon_batch<9>, grid, block, 0, s, in, n, out, pv); break;
    default: throw Error("unsupported Poseidon arity " + std::to_string(arity));
  }
  RLN_HIP(hipGetLastError());
}

}  // namespace rlnamd
